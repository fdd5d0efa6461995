"""CPU tests of the host-side logic (recipe grammar, Python-2 number formatting, lna
renaming, feature files) and property tests of the sufficient-statistics algebra
the kernels rely on."""
import io
import math
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from conftest import pkg

recipe = pkg('recipe')
feaio = pkg('feaio')
synth = pkg('synth')


def test_py2_float_str_matches_python2_semantics():
    f = recipe.py2_float_str
    assert f(1.0) == '1.0'
    assert f(13.492) == '13.492'
    assert f(1561.5 / 125 + 1.0) == '13.492'          # 12 significant digits hide the last ulps
    assert f(0.1 + 0.2) == '0.3'
    assert f(123456789012.5) in ('123456789012.0', '123456789013.0')   # round-half-even of the 12th digit
    assert f(1e16) == '1e+16'
    assert f(1e-5) == '1e-05'
    assert f(float('inf')) == 'inf' and f(float('-inf')) == '-inf' and f(float('nan')) == 'nan'
    assert f(100000000000.0) == '100000000000.0'
    assert f(1000000000000.0) == '1e+12'


@given(st.floats(min_value=0.0, max_value=1e6, allow_nan=False))
@settings(max_examples=300, deadline=None)
def test_roundtrip_is_idempotent(x):
    once = float(recipe.py2_float_str(x))
    assert float(recipe.py2_float_str(once)) == once
    assert abs(once - x) <= 5e-12 * max(1.0, abs(x))


def test_parse_recipe_skips_and_echoes_bad_lines():
    text = ['audio=a.wav lna=a_1 start-time=1.0 end-time=2.5\n',
            'audio=a.wav lna=a_2 start-time=3 end-time=4.0\n',        # integer-only time: no match
            'lna=a_3 start-time=5.0 end-time=6.0\n',                   # no audio
            'audio=b.wav lna=b_1 start-time=1e+3 end-time=2000.0 speaker=x\n',   # exponent form: no match
            'audio=c.wav alignment=x lna=c_9 start-time=10.25 end-time=11.5 speaker=spk_turn\n']
    echoed = []
    r = recipe.parse_recipe(text, echo=echoed.append)
    assert [tuple(x) for x in r] == [('a.wav', 'a_1', 1.0, 2.5), ('c.wav', 'c_9', 10.25, 11.5)]
    assert echoed.count('Recipe line without recognizable data:') == 3


def test_lna_renaming_state_machine():
    out = io.StringIO()
    w = recipe.RecipeWriter(out, 125.0)
    rl = lambda lna: recipe.RecipeLine('x.wav', lna, 0.0, 1.0)
    for lna in ('a_1', 'a_1', 'a_7', 'b_2', 'b_2', 'a_3', 'noscore'):
        w.write(rl(lna), 0, 125, 0, 'spk_turn')
    names = [ln.split()[1] for ln in out.getvalue().splitlines()]
    # counter continues inside a prefix, restarts on a new one; without '_' the prefix is
    # everything but the last character and the name becomes the counter alone (A-10)
    assert names == ['lna=a_1', 'lna=a_2', 'lna=a_3', 'lna=b_1', 'lna=b_2', 'lna=a_1', 'lna=1']
    out2 = io.StringIO()
    w2 = recipe.RecipeWriter(out2, 125.0, rename_lna=False)
    w2.write(rl('zz_9'), 250, 375, 1.5, 'speaker_3')
    assert out2.getvalue() == 'audio=x.wav lna=zz_9 start-time=3.5 end-time=4.5 speaker=speaker_3\n'


def test_fea_roundtrip_and_errors(tmp_path):
    feats = np.arange(5 * 39, dtype=np.float32).reshape(5, 39)
    p = os.path.join(str(tmp_path), 'a.fea')
    synth.write_fea(p, feats)
    dim, back = feaio.load_features(p)
    assert dim == 39 and np.array_equal(back, feats)
    with open(p, 'ab') as f:
        f.write(b'\x00\x00\x00\x00')                    # a dangling partial frame
    with pytest.raises(ValueError):
        feaio.load_features(p)
    assert feaio.fea_path('/x/y/meeting.wav', 'fea', '.fea') == os.path.join('fea', 'meeting.fea')
    assert feaio.fea_path('/x/y/meeting.wav', 'fea/', '.fea', join=False) == 'fea/meeting.fea'


def test_synthetic_generator_is_reproducible():
    a, va, ta = synth.make_session(11, 40, 3)
    b, vb, tb = synth.make_session(11, 40, 3)
    assert synth.fea_sha256(a) == synth.fea_sha256(b) and va == vb and ta == tb
    c, _, _ = synth.make_session(12, 40, 3)
    assert synth.fea_sha256(a) != synth.fea_sha256(c)
    assert a.dtype == np.float32 and a.shape == (5000, 39)
    covered = sum(e - s for s, e in va)
    assert 0 < covered < a.shape[0]


# ---------------------------------------------------------------- statistics algebra
def _rec(x):
    xa = np.concatenate([x.astype(np.float64), np.ones((x.shape[0], 1))], axis=1)
    m = xa.T @ xa
    return np.concatenate([m[r, r:] for r in range(40)])


def _cov_from_rec(rec):
    m = np.zeros((40, 40))
    k = 0
    for r in range(40):
        m[r, r:] = rec[k:k + 40 - r]
        k += 40 - r
    m = m + np.triu(m, 1).T
    n = m[39, 39]
    s = m[:39, 39]
    return (m[:39, :39] - np.outer(s, s) / n) / (n - 1)


@given(st.integers(min_value=0, max_value=2 ** 31), st.integers(min_value=45, max_value=400),
       st.integers(min_value=45, max_value=400))
@settings(max_examples=25, deadline=None)
def test_merged_statistics_give_numpy_cov_of_the_concatenation(seed, n1, n2):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((n1, 39)) * rng.uniform(0.5, 2.0, 39) + rng.standard_normal(39)).astype(np.float32)
    y = (rng.standard_normal((n2, 39)) * rng.uniform(0.5, 2.0, 39) + rng.standard_normal(39)).astype(np.float32)
    got = _cov_from_rec(_rec(x) + _rec(y))
    want = np.cov(np.concatenate((x, y)), rowvar=0)
    assert np.max(np.abs(got - want)) <= 1e-11 * np.max(np.abs(want))
    # permutation invariance of the record
    perm = rng.permutation(n1)
    assert np.allclose(_rec(x[perm]), _rec(x), rtol=1e-13, atol=1e-9)


def test_c_oracle_statistics_match_the_algebra():
    from oracle.c_engine import COracleEngine
    feats, _, _ = synth.make_session(3, 30, 2)
    e = COracleEngine()
    e.set_features(feats)
    got = e.stats([[(10, 500)], [(0, 100), (200, 260)]])
    assert np.allclose(got[0], _rec(feats[10:500]), rtol=1e-13, atol=1e-9)
    assert np.allclose(got[1], _rec(np.concatenate((feats[0:100], feats[200:260]))), rtol=1e-13, atol=1e-9)


def test_library_roundtrip_helper_equals_python_formatting():
    """spkd_py2_roundtrip (exact 128-bit fast path + printf fallback) against the
    pure-Python '%.12g' round trip, bit for bit."""
    hipabi = pkg('hipabi')
    rng = np.random.default_rng(2026)
    groups = [rng.random(50000) * 10.0 ** rng.integers(-3, 11, 50000),
              rng.integers(0, 4500000, 50000) / 125.0 + rng.integers(0, 3000, 50000) / 1000.0,
              np.arange(20000) * 0.0005 + 0.00025,                      # decimal ties after scaling
              np.array([0.0, 1e-3, 99999999999.5, 9.9999999999995, 1.0000000000005, 13.492,
                        1561.5 / 125 + 1.0, 1e11, 5e-4, 1e15, 2.5e-3, 123456789.0125])]
    for v in groups:
        got = hipabi.py2_roundtrip(v)
        want = np.array([float(recipe.py2_float_str(x)) for x in v])
        assert np.array_equal(got, want)


def test_labels_from_merges_helper():
    hipabi = pkg('hipabi')
    import random
    random.seed(5)
    n = 40
    clusters = [[i] for i in range(n)]
    a_list, b_list = [], []
    for _ in range(31):
        m = len(clusters)
        a = random.randrange(m - 1)
        b = random.randrange(a + 1, m)
        a_list.append(a); b_list.append(b)
        clusters[a].extend(clusters[b])
        clusters.pop(b)
    want = [0] * n
    for k, c in enumerate(clusters):
        for i in c:
            want[i] = k + 1
    assert hipabi.labels_from_merges(n, a_list, b_list).tolist() == want


def test_labels_from_merges_batch_matches_the_replay():
    """The batch form on ragged problems (spkd_ahc's output layout), against a literal
    replay of speakers[a].extend(speakers[b]); speakers.pop(b) (spk-clustering.py:216-217)."""
    hipabi = pkg('hipabi')
    import random
    random.seed(11)
    sizes = [1, 2, 57, 0, 300, 5] * 8
    seg_off = np.zeros(len(sizes) + 1, dtype=np.int64)
    seg_off[1:] = np.cumsum(sizes)
    a_all = np.zeros(int(seg_off[-1]), dtype=np.int32)
    b_all = np.zeros(int(seg_off[-1]), dtype=np.int32)
    n_merges = np.zeros(len(sizes), dtype=np.int32)
    want = np.zeros(int(seg_off[-1]), dtype=np.int32)
    for p, n in enumerate(sizes):
        clusters = [[i] for i in range(n)]
        nm = random.randrange(0, n) if n > 1 else 0
        for k in range(nm):
            m = len(clusters)
            a = random.randrange(m - 1)
            b = random.randrange(a + 1, m)
            a_all[seg_off[p] + k] = a
            b_all[seg_off[p] + k] = b
            clusters[a].extend(clusters[b])
            clusters.pop(b)
        n_merges[p] = nm
        for k, c in enumerate(clusters):
            for i in c:
                want[seg_off[p] + i] = k + 1
    got = hipabi.labels_from_merges_batch(seg_off, n_merges, a_all, b_all)
    assert np.array_equal(got, want)
    for p, n in enumerate(sizes):
        o, nm = int(seg_off[p]), int(n_merges[p])
        one = hipabi.labels_from_merges(n, a_all[o:o + nm], b_all[o:o + nm])
        assert np.array_equal(one, want[o:o + n])


def test_gw_lines_helper_matches_the_vectorised_restatement():
    """spkd_count_flags / spkd_gw_lines (host-side helpers of the batch pipeline) against the
    numpy expressions they replaced: same doubles, same frame ranges, same order."""
    hipabi = pkg('hipabi')
    rng = np.random.default_rng(77)
    nt, rate = 300, 125.0
    cap = rng.integers(3, 40, nt)
    off = np.zeros(nt + 1, dtype=np.int64)
    off[1:] = np.cumsum(cap)
    n_win = np.array([rng.integers(0, c + 1) for c in cap], dtype=np.int32)
    flags = rng.integers(-3, 4, int(off[-1])).astype(np.int32)          # garbage behind the windows included
    for t in range(nt):
        flags[off[t]:off[t] + n_win[t]] = rng.integers(0, 2, n_win[t])
    nd = hipabi.count_flags(flags, off[:-1], n_win)
    assert nd.tolist() == [int(flags[off[t]:off[t] + n_win[t]].sum()) for t in range(nt)]
    det_start = rng.uniform(0, 5000, int(off[-1]))
    det_maxi = rng.uniform(50, 900, int(off[-1])) + rng.integers(0, 2, int(off[-1])) * 0.5
    final_start = rng.uniform(0, 6000, nt)
    ls = np.round(rng.uniform(0, 3000, nt), 3)
    le = ls + np.round(rng.uniform(1, 60, nt), 3)
    tb = rng.integers(0, 10 ** 8, nt).astype(np.int64)
    te = tb + rng.integers(100, 8000, nt)
    for text in (True, False):
        got = hipabi.gw_lines(off[:-1], nd, det_start, det_maxi, final_start, ls, le, tb, te, rate,
                              text_contract=text, want_frames=True)
        times, turn, fb, fe, ix = [], [], [], [], []
        for t in range(nt):
            for j in range(int(nd[t])):
                s = det_start[off[t] + j]
                e = s + det_maxi[off[t] + j]
                times.append((s / rate + ls[t], e / rate + ls[t]))
                fb.append(tb[t] + int(s)); fe.append(tb[t] + int(e)); ix.append(off[t] + j); turn.append(t)
            times.append((final_start[t] / rate + ls[t], ((le[t] - ls[t]) * rate) / rate + ls[t]))
            fb.append(tb[t] + int(final_start[t])); fe.append(te[t]); ix.append(off[t] + int(nd[t])); turn.append(t)
        want = np.array(times)
        if text:
            want = hipabi.py2_roundtrip(want.ravel()).reshape(-1, 2)
        assert np.array_equal(got['times'], want)
        assert got['turn'].tolist() == turn and got['frame_b'].tolist() == fb and got['frame_e'].tolist() == fe
        assert got['index'].tolist() == ix


def test_in_flight_keeps_order_and_surfaces_errors():
    """pipeline.in_flight: one job per context at a time, results in job order, a job's
    exception re-raised to the consumer (contexts faked: no GPU involved)."""
    import threading
    import time
    pl = pkg('pipeline')
    users = {}

    def job(ctx, k):
        users.setdefault(ctx, set()).add(threading.get_ident())
        time.sleep(0.005 * (3 - k % 3))
        return k * k

    assert list(pl.in_flight(['A', 'B'], 9, job)) == [k * k for k in range(9)]
    assert set(users) == {'A', 'B'} and all(len(v) == 1 for v in users.values())    # a context stays with its thread
    assert list(pl.in_flight(['A'], 3, lambda c, k: (c, k))) == [('A', 0), ('A', 1), ('A', 2)]

    def bad(ctx, k):
        if k == 2:
            raise ValueError('boom')
        return k

    with pytest.raises(ValueError):
        list(pl.in_flight(['A', 'B'], 6, bad))


def _merge_run(tmp, feats, spec, monkeypatch, tag):
    """-m m over a hand-made recipe on the C oracle engine; spec=False switches the look-ahead off."""
    from oracle.c_engine import COracleEngine
    cli = pkg('cli')
    cd = pkg('change_detection')
    os.makedirs(os.path.join(tmp, 'fea'), exist_ok=True)
    synth.write_fea(os.path.join(tmp, 'fea', 'f.fea'), feats)
    lines = []
    t = 0.0
    for k in range(24):
        lines.append('audio=f.wav lna=a_%d start-time=%s end-time=%s speaker=spk_turn\n' % (k + 1, t, t + 3.0))
        t += 3.5
    rec = os.path.join(tmp, 'in.recipe')
    with open(rec, 'w') as fh:
        fh.writelines(lines)
    if not spec:
        monkeypatch.setattr(cd.ChangeDetectionRun, '_merge_speculate', lambda self, recipe, l, n: {})
    out = os.path.join(tmp, tag + '.recipe')
    if os.path.exists(out):
        os.remove(out)
    said = io.StringIO()
    err = None
    try:
        cli.main_change_detection([rec, os.path.join(tmp, 'fea'), '-o', out, '-m', 'm', '-d', 'BIC', '-l', '1.3', '-tt'],
                                  engine=COracleEngine(), stdout=said)
    except ValueError as e:
        err = str(e)
    monkeypatch.undo()
    return (open(out).read() if os.path.exists(out) else None), said.getvalue(), err


def test_merge_mode_look_ahead_changes_nothing(tmp_path, monkeypatch):
    """Merge mode takes the terms of every adjacent pair of recipe lines in one engine call and
    goes back for single pairs only behind a merge (change_detection._merge_speculate): same
    lines, same printed distances as one call per decision -- on clean frames, and on frames
    with a NaN in a late line, where the reference raises AT the offending pair after the
    earlier lines were written (the look-ahead then stands down)."""
    feats, _, _ = synth.make_session(909, 90, 3)
    ref = _merge_run(str(tmp_path / 'a'), feats, False, monkeypatch, 'out')
    got = _merge_run(str(tmp_path / 'a'), feats, True, monkeypatch, 'out')
    assert got == ref and ref[2] is None
    assert ref[0].count('\n') < 24                  # some lines were merged: the single calls ran too
    bad = feats.copy()
    bad[int(60.0 * 125) + 10, 3] = np.nan           # inside line 18
    ref = _merge_run(str(tmp_path / 'c'), bad, False, monkeypatch, 'out')
    got = _merge_run(str(tmp_path / 'c'), bad, True, monkeypatch, 'out')
    assert ref[2] is not None and 'infs or NaNs' in ref[2]
    assert got == ref
