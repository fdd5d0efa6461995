"""GPU tests at BASELINE.json's full sizes (1 h, 4 and 8 speakers): the HIP path
against the C oracle (oracle/spkd_oracle.c, itself pinned on the reference
goldens), plus size-independent properties (additivity of statistics, batch ==
single file, idempotence of the in-memory pipeline vs the file-based scripts)."""
import importlib
import io
import os
import re

import numpy as np
import pytest

from helpers import ROOT
from conftest import pkg

pytestmark = pytest.mark.gpu

CD = ['-m', 'gw', '-d', 'BIC', '-w', '1.0', '-st', '3.0', '-dws', '0.1', '-l', '1.0']
CL = ['-m', 'hi', '-l', '1.3']


@pytest.fixture(scope='module')
def eng():
    e = pkg('engine').HipEngine(0)
    yield e
    e.close()


def _run_scripts(tmp, engine, tag, variant=1):
    cli = pkg('cli')
    spkc = os.path.join(tmp, tag + '.spkc.recipe')
    final = os.path.join(tmp, tag + '.out.recipe')
    out1, out2 = io.StringIO(), io.StringIO()
    cli.main_change_detection([os.path.join(tmp, 'vad.recipe'), os.path.join(tmp, 'fea') + '/', '-o', spkc] + CD,
                              engine=engine, stdout=out1)
    cli.main_clustering([spkc, os.path.join(tmp, 'fea') + '/', '-o', final] + CL, variant=variant,
                        engine=engine, stdout=out2)
    return open(spkc).read(), open(final).read(), out1.getvalue(), out2.getvalue()


def _merge_term_scales(feats, spkc_text, merges, lambdac=1.3):
    """Size of the terms each BIC merge distance is a difference of: the merge log (compacted
    indices, spk-clustering.py:204-217) replayed on the segments' statistics records, the
    three log dets of every merged pair from the oracle's pair function."""
    import ctypes as C
    from oracle.c_engine import COracleEngine
    orc = COracleEngine(1)
    orc.set_features(feats)
    segs = [(int(float(a) * 125.0), int(float(b) * 125.0)) for a, b in
            re.findall(r'start-time=(\S+) end-time=(\S+)', spkc_text)]
    recs = list(orc.stats([[s_] for s_ in segs]))
    out8 = np.zeros(8, dtype=np.float64)
    scales = []
    for a, b in merges:                         # (printed 1-based, spk-clustering.py:214)
        a, b = a - 1, b - 1
        ra, rb = np.ascontiguousarray(recs[a]), np.ascontiguousarray(recs[b])
        rc = orc.lib.orc_pair_terms(C.c_void_p(ra.ctypes.data), C.c_void_p(rb.ctypes.data), 0, 0, None, None,
                                    C.c_void_p(out8.ctypes.data))
        assert rc == 0
        n1, n2, l1, l2, lx = out8[:5]
        n = n1 + n2
        scales.append((0.5 * n * abs(lx) + 0.5 * n1 * abs(l1) + 0.5 * n2 * abs(l2) + lambdac * 0.5 * 819.0 * np.log(n),
                       min(n1, n2)))
        recs[a] = ra + rb
        del recs[b]
    return scales


def _merge_lines(text):
    return [(int(a), int(b), float(d)) for a, b, d in
            re.findall(r'Merging: (\d+) and (\d+) distance: (\S+)', text)]


@pytest.mark.parametrize('nspk,seed', [(4, 5150), (8, 8088)])
def test_one_hour_matches_c_oracle(tmp_path, eng, nspk, seed):
    from oracle.c_engine import COracleEngine
    synth = pkg('synth')
    feats, vad, _ = synth.make_session(seed, 3600, nspk)
    tmp = str(tmp_path)
    os.makedirs(os.path.join(tmp, 'fea'))
    synth.write_fea(os.path.join(tmp, 'fea', 'hour.fea'), feats)
    with open(os.path.join(tmp, 'vad.recipe'), 'w') as f:
        f.write(synth.vad_recipe_text('hour.wav', vad))
    h = _run_scripts(tmp, eng, 'hip')
    o = _run_scripts(tmp, COracleEngine(), 'orc')
    assert h[0] == o[0]                    # change-detection recipe: integer boundaries bit exact
    assert h[1] == o[1]                    # speaker recipe: labels identical (numbering included)
    mh, mo = _merge_lines(h[3]), _merge_lines(o[3])
    assert [(a, b) for a, b, _ in mh] == [(a, b) for a, b, _ in mo]
    worst = max(abs(x[2] - y[2]) / max(1.0, abs(y[2])) for x, y in zip(mh, mo))
    assert worst < 1e-9, worst
    print('%d speakers: %d turns, %d merges, worst merge-distance rel err %.2g' % (
        nspk, h[0].count('\n'), len(mh), worst))


def test_twenty_minutes_glr_matches_c_oracle(tmp_path, eng):
    """The GLR branch of both stages at length (the goldens hold it at 150 s): growing-window
    GLR change detection (three matrices per candidate), then v2 clustering on GLR."""
    from oracle.c_engine import COracleEngine
    synth = pkg('synth')
    cli = pkg('cli')
    feats, vad, _ = synth.make_session(60606, 1200, 5)
    tmp = str(tmp_path)
    os.makedirs(os.path.join(tmp, 'fea'))
    synth.write_fea(os.path.join(tmp, 'fea', 'glr.fea'), feats)
    with open(os.path.join(tmp, 'vad.recipe'), 'w') as f:
        f.write(synth.vad_recipe_text('glr.wav', vad))
    res = {}
    for tag, engine in (('hip', eng), ('orc', COracleEngine())):
        spkc = os.path.join(tmp, tag + '.spkc.recipe')
        final = os.path.join(tmp, tag + '.out.recipe')
        o1, o2 = io.StringIO(), io.StringIO()
        cli.main_change_detection([os.path.join(tmp, 'vad.recipe'), os.path.join(tmp, 'fea') + '/', '-o', spkc,
                                   '-m', 'gw', '-d', 'GLR', '-w', '1.0', '-st', '3.0', '-dws', '0.1', '-t', '900'],
                                  engine=engine, stdout=o1)
        cli.main_clustering([spkc, os.path.join(tmp, 'fea') + '/', '-o', final, '-m', 'hi', '-d', 'GLR',
                             '-t', '1500'], variant=2, engine=engine, stdout=o2)
        res[tag] = (open(spkc).read(), open(final).read(), o2.getvalue())
    assert res['hip'][0] == res['orc'][0]
    assert res['hip'][1] == res['orc'][1]
    assert res['hip'][0].count('\n') > 100
    mh, mo = _merge_lines(res['hip'][2]), _merge_lines(res['orc'][2])
    assert len(mh) > 50 and [(a, b) for a, b, _ in mh] == [(a, b) for a, b, _ in mo]
    assert max(abs(x[2] - y[2]) / max(1.0, abs(y[2])) for x, y in zip(mh, mo)) < 1e-9


def test_random_parameters_match_c_oracle(tmp_path, eng):
    """The growing-window state machine over parameter combinations the goldens do not hold
    (window, step, growth, penalty / threshold, every distance), short sessions with ragged
    turns, against the C oracle (itself pinned by the reference's goldens): the change
    recipe byte for byte, then the clustering recipe of either script version."""
    import random
    from oracle.c_engine import COracleEngine
    synth = pkg('synth')
    cli = pkg('cli')
    rnd = random.Random(20261004)
    orc = COracleEngine()
    tmp = str(tmp_path)
    os.makedirs(os.path.join(tmp, 'fea'))
    for trial in range(14):
        secs = rnd.choice([45, 90, 140])
        nspk = rnd.choice([2, 3, 4])
        kw = rnd.choice([{}, dict(min_turn=1.5, max_turn=4.0), dict(min_turn=6.0, max_turn=25.0)])
        feats, vad, _ = synth.make_session(9000 + trial, secs, nspk, **kw)
        name = 'r%d' % trial
        synth.write_fea(os.path.join(tmp, 'fea', name + '.fea'), feats)
        vr = os.path.join(tmp, name + '.vad.recipe')
        with open(vr, 'w') as f:
            f.write(synth.vad_recipe_text(name + '.wav', vad))
        kind = rnd.choice(['BIC', 'BIC', 'GLR', 'KL2'])
        w = rnd.choice(['0.5', '1.0', '1.5', '2.0'])
        st = rnd.choice(['0.3', '1.0', '2.5', '3.0'])
        dws = rnd.choice(['0.05', '0.1', '0.3'])
        cd = ['-m', 'gw', '-d', kind, '-w', w, '-st', st, '-dws', dws]
        cd += {'BIC': ['-l', rnd.choice(['0.8', '1.0', '1.4'])], 'GLR': ['-t', rnd.choice(['700', '900', '1200'])],
               'KL2': ['-t', rnd.choice(['40', '60'])]}[kind]
        ckind = rnd.choice(['BIC', 'GLR'])
        variant = rnd.choice([1, 2])
        cl = ['-m', 'hi', '-d', ckind] + (['-l', rnd.choice(['1.0', '1.3'])] if ckind == 'BIC' else ['-t', '1500'])
        out = {}
        for tag, engine in (('hip', eng), ('orc', orc)):
            spkc = os.path.join(tmp, '%s.%s.spkc' % (name, tag))
            final = os.path.join(tmp, '%s.%s.out' % (name, tag))
            cli.main_change_detection([vr, os.path.join(tmp, 'fea') + '/', '-o', spkc] + cd, engine=engine,
                                      stdout=io.StringIO())
            cli.main_clustering([spkc, os.path.join(tmp, 'fea') + '/', '-o', final] + cl, variant=variant,
                                engine=engine, stdout=io.StringIO())
            out[tag] = (open(spkc).read(), open(final).read())
        assert out['hip'][0] == out['orc'][0], (trial, cd)
        assert out['hip'][1] == out['orc'][1], (trial, cd, cl, variant)


def test_statistics_are_additive(eng):
    synth = pkg('synth')
    feats, _, _ = synth.make_session(99, 300, 3)
    eng.set_features(feats)
    n = feats.shape[0]
    cuts = [0, 17, 1024, 1025, 5000, 20000, n]
    parts = eng.stats([[(a, b)] for a, b in zip(cuts[:-1], cuts[1:])])
    whole = eng.stats([[(0, n)]])[0]
    one_set = eng.stats([[(a, b) for a, b in zip(cuts[:-1], cuts[1:])]])[0]
    s = parts.sum(axis=0)
    scale = np.maximum(1.0, np.abs(whole))
    assert np.max(np.abs(s - whole) / scale) < 1e-13
    assert np.max(np.abs(one_set - whole) / scale) < 1e-13
    assert whole[819] == n


def test_distance_matrix_is_symmetric_and_matches_pair_terms(eng):
    synth = pkg('synth')
    hipabi = pkg('hipabi')
    cd = pkg('change_detection')
    feats, _, truth = synth.make_session(123, 240, 4)
    eng.set_features(feats)
    segs = [(a, b) for a, b, _ in truth][:24]
    d = eng._stats_of_sets([[s] for s in segs])
    n = len(segs)
    dm = eng.ctx.dev_alloc(n * n * 8)
    try:
        eng.ctx.distance_matrix('BIC', 1.3, d, n, dm)
        m = np.empty((n, n))
        eng.ctx.d2h(m, dm)
    finally:
        eng.ctx.dev_free(dm)
        eng.ctx.dev_free(d)
    assert np.array_equal(m, m.T)
    assert np.all(np.diag(m) == 2.0 ** 63)
    jobs = [([segs[i]], [segs[j]]) for i in range(n) for j in range(i + 1, n)]
    terms = eng.pair_terms(jobs)
    k = 0
    for i in range(n):
        for j in range(i + 1, n):
            want = cd.bic_from_terms(terms[k], 1.3)
            assert abs(m[i, j] - want) <= 1e-10 * max(1.0, abs(want))
            k += 1


@pytest.mark.parametrize('variant', [1, 2])
def test_ahc_ragged_problems_wide_and_mono_agree(eng, variant):
    """Several problems of different sizes in one spkd_ahc call (the largest ~1 700
    segments: the statistics of a 4 h recording, BASELINE.json config 5 in small):
    the chained-launch form (grids sized for the largest problem, smaller ones
    finishing early) and the one-workgroup form give identical merge logs."""
    synth = pkg('synth')
    hipabi = pkg('hipabi')
    feats, _, truth = synth.make_session(31337, 14400, 8)
    eng.set_features(feats)
    segs = [(a, b) for a, b, _ in truth]
    sizes = [len(segs), 1, 37, 2, 300]
    sets, seg_off = [], [0]
    for k, n in enumerate(sizes):
        sets += [[s] for s in segs[k:k + n]]
        seg_off.append(seg_off[-1] + n)
    d = eng._stats_of_sets(sets)
    try:
        res = {}
        for path in (hipabi.AHC_MONO, hipabi.AHC_WIDE):
            p = hipabi.AhcParams(variant, hipabi.KINDS['BIC'], 0, path, 1.3, 0.0)
            res[path] = eng.ctx.ahc(d, seg_off, p)
    finally:
        eng.ctx.dev_free(d)
    a, b = res[hipabi.AHC_MONO], res[hipabi.AHC_WIDE]
    assert a['status'] == b['status'] == 0
    assert np.array_equal(a['n_merges'], b['n_merges'])
    assert a['n_merges'][0] > len(segs) // 2 and a['n_merges'][1] == 0
    for k in range(len(sizes)):
        lo, hi = seg_off[k], seg_off[k] + int(a['n_merges'][k])
        assert np.array_equal(a['a'][lo:hi], b['a'][lo:hi])
        assert np.array_equal(a['b'][lo:hi], b['b'][lo:hi])
        assert np.array_equal(a['d'][lo:hi], b['d'][lo:hi])
    assert np.array_equal(a['stat_max'], b['stat_max'], equal_nan=True)
    assert np.array_equal(a['stat_min'], b['stat_min'], equal_nan=True)


def test_batch_pipeline_equals_file_based_scripts(tmp_path, eng):
    """The in-memory batch pipeline (what bench.py times) must give, per file, the
    recipe rows the two drop-in scripts produce through files."""
    synth = pkg('synth')
    pipeline = pkg('pipeline')
    recipe = pkg('recipe')
    sessions = [synth.make_session(31 + i, 200 + 40 * i, 3 + (i % 2)) for i in range(3)]
    frames = np.concatenate([s[0] for s in sessions])
    eng.set_features(frames)
    files, off = [], 0
    for feats, vad, _ in sessions:
        v = [(float(recipe.py2_float_str(a / 125.0)), float(recipe.py2_float_str(b / 125.0))) for a, b in vad]
        files.append(pipeline.BatchFile(off, feats.shape[0], v))
        off += feats.shape[0]
    got = pipeline.diarize_batch(eng.ctx, eng.d_frames, frames.shape[0], files)
    # the opt-in fused mode (no 12-digit text between the stages): same segmentation and
    # labels here, times equal to 12 digits
    fused = pipeline.diarize_batch(eng.ctx, eng.d_frames, frames.shape[0], files, text_contract=False)
    for g, f in zip(got, fused):
        assert g.shape == f.shape and np.array_equal(g[:, 2], f[:, 2])
        assert np.allclose(g[:, :2], f[:, :2], rtol=1e-11, atol=0.0)
    e2 = pkg('engine').HipEngine(0)
    try:
        for k, (feats, vad, _) in enumerate(sessions):
            tmp = os.path.join(str(tmp_path), 'f%d' % k)
            os.makedirs(os.path.join(tmp, 'fea'))
            synth.write_fea(os.path.join(tmp, 'fea', 'x.fea'), feats)
            with open(os.path.join(tmp, 'vad.recipe'), 'w') as f:
                f.write(synth.vad_recipe_text('x.wav', vad))
            _, final, _, _ = _run_scripts(tmp, e2, 'hip')
            rows = re.findall(r'start-time=(\S+) end-time=(\S+) speaker=speaker_(\d+)', final)
            want = [(float(a), float(b), int(c)) for a, b, c in rows]
            assert [(a, b, int(c)) for a, b, c in got[k].tolist()] == want, k
    finally:
        e2.close()


def test_batch_with_empty_and_tiny_files(eng):
    """Ragged batches: a file without any VAD turn, a file whose only turn is shorter than
    two windows (one segment, one speaker) and a normal file, in one launch each stage."""
    synth = pkg('synth')
    pipeline = pkg('pipeline')
    recipe = pkg('recipe')
    feats, vad, _ = synth.make_session(77, 120, 3)
    T = feats.shape[0]
    frames = np.concatenate([feats, feats[:3000], feats])
    eng.set_features(frames)
    v = [(float(recipe.py2_float_str(a / 125.0)), float(recipe.py2_float_str(b / 125.0))) for a, b in vad]
    files = [pipeline.BatchFile(0, T, []),                        # no speech at all
             pipeline.BatchFile(T, 3000, [(2.0, 3.5)]),           # 187 frames: no scan fits
             pipeline.BatchFile(T + 3000, T, v)]
    got = pipeline.diarize_batch(eng.ctx, eng.d_frames, frames.shape[0], files)
    assert got[0].shape == (0, 3)
    assert got[1].tolist() == [[2.0, 3.5, 1.0]]
    alone = pipeline.diarize_batch(eng.ctx, eng.d_frames, frames.shape[0], [files[2]])
    assert np.array_equal(got[2], alone[0]) and len(got[2]) >= len(v)
    assert pipeline.diarize_batch(eng.ctx, eng.d_frames, frames.shape[0], [files[0]])[0].shape == (0, 3)


def test_chain_from_exp_files(tmp_path, eng):
    """The three stages as the reference chains them (voice-detection2.py -> change
    detection -> clustering): a `.exp` speech / non-speech token stream instead of a
    ready-made VAD recipe, the HIP path against the C oracle on the same files."""
    from oracle.c_engine import COracleEngine
    synth = pkg('synth')
    vd = pkg('voice_detection')
    feats, vad, _ = synth.make_session(4711, 600, 4)
    tmp = str(tmp_path)
    os.makedirs(os.path.join(tmp, 'fea'))
    os.makedirs(os.path.join(tmp, 'exp'))
    synth.write_fea(os.path.join(tmp, 'fea', 'talk.fea'), feats)
    toks = ['0 <w>'] + ['%d p %d <w>' % (a, b) for a, b in vad]
    with open(os.path.join(tmp, 'exp', 'talk.exp'), 'w') as f:
        f.write(' '.join(toks) + '\n')
    with open(os.path.join(tmp, 'exp', 'talk.last_frame'), 'w') as f:
        f.write(str(feats.shape[0]))
    with open(os.path.join(tmp, 'wav.recipe'), 'w') as f:
        f.write('audio=/data/talk.wav\n')
    vd.main([os.path.join(tmp, 'wav.recipe'), os.path.join(tmp, 'exp'), '-o', os.path.join(tmp, 'vad.recipe')],
            stdout=io.StringIO())
    turns = open(os.path.join(tmp, 'vad.recipe')).read().splitlines()
    assert len(turns) == len(vad) and turns[0].startswith('audio=/data/talk.wav lna=a_1 start-time=')
    h = _run_scripts(tmp, eng, 'hip')
    o = _run_scripts(tmp, COracleEngine(), 'orc')
    assert h[0] == o[0] and h[1] == o[1]
    assert h[1].count('speaker=') == h[0].count('\n')


def test_der_against_the_synthetic_ground_truth(tmp_path, eng):
    """Accuracy sanity of the whole path with the reference's own scorer
    (clus-performance.py, ported in exporters.py): the synthetic sessions come with the
    true speaker of every frame.  The C oracle gives 0.87 % on this session."""
    synth = pkg('synth')
    ex = pkg('exporters')
    feats, vad, truth = synth.make_session(4711, 600, 4)
    tmp = str(tmp_path)
    os.makedirs(os.path.join(tmp, 'fea'))
    synth.write_fea(os.path.join(tmp, 'fea', 'x.fea'), feats)
    with open(os.path.join(tmp, 'vad.recipe'), 'w') as f:
        f.write(synth.vad_recipe_text('x.wav', vad))
    _, final, _, _ = _run_scripts(tmp, eng, 'hip')
    rows = re.findall(r'start-time=(\S+) end-time=(\S+) speaker=(\S+)', final)
    proposed = sorted((float(a), float(b), c) for a, b, c in rows)
    baseline = [(a / 125.0, b / 125.0, 'true_%d' % k) for a, b, k in truth]
    correct, incorrect = ex.der(baseline, proposed)
    rate = incorrect / float(correct + incorrect)
    print('DER vs ground truth: %.4f' % rate)
    assert rate < 0.02


def test_empty_and_short_turns(eng):
    """Edge cases of dist_gw: a turn shorter than two windows gives only the tail
    line; an empty turn list is a no-op."""
    synth = pkg('synth')
    feats, _, _ = synth.make_session(5, 30, 2)
    eng.set_features(feats)
    assert eng.gw([], 'BIC', 1.0, 0.0, 125.0, 375.0, 12.0, 125.0) == []
    res = eng.gw([(100, 400), (0, 249), (10, 10)], 'BIC', 1.0, 0.0, 125.0, 375.0, 12.0, 125.0)
    assert res[1].events == [] and res[1].final_start == 0.0
    assert res[2].events == [] and res[2].final_start == 0.0
    assert [e[0] for e in res[0].events].count('win') >= 1


def test_dropin_scripts_as_subprocesses(tmp_path):
    """The executables spk-diarization2.py would call (./spk-change-detection.py,
    ./spk-clustering.py; spk-diarization2.py:122-128), run as child processes with
    its exact argv from the repository root, against the reference goldens."""
    import subprocess
    import sys
    from helpers import load_cases, session
    synth = pkg('synth')
    cases = {c['name']: c for c in load_cases()}
    cd, cl = cases['B_cd_gw_bic'], cases['B_cl1_hi_bic']
    feats, _, _ = session(cd['session'])
    tmp = str(tmp_path)
    os.makedirs(os.path.join(tmp, 'fea'))
    synth.write_fea(os.path.join(tmp, 'fea', 'meeting.fea'), feats)
    with open(os.path.join(tmp, 'vad.recipe'), 'w') as f:
        f.write(cd['input_recipe'])
    spkc = os.path.join(tmp, 'spkc.recipe')
    out = os.path.join(tmp, 'out.recipe')
    r1 = subprocess.run([sys.executable, './spk-change-detection.py', os.path.join(tmp, 'vad.recipe'),
                         os.path.join(tmp, 'fea'), '-o', spkc, '-m', 'gw', '-d', 'BIC', '-w', '1.0',
                         '-st', '3.0', '-dws', '0.1', '-l', '1.0'], cwd=ROOT, capture_output=True, text=True)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, './spk-clustering.py', spkc, os.path.join(tmp, 'fea'), '-o', out,
                         '-m', 'hi', '-l', '1.3'], cwd=ROOT, capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert open(spkc).read() == cd['output_recipe']
    assert open(out).read() == cl['output_recipe']
    assert 'Merging:' in r2.stdout and 'Total detected speakers:' in r2.stdout


def _device_batch(n_files, seconds, nspk, seed0):
    """n_files distinct synthetic sessions generated on the GPU (synth_device.py: the same
    bytes as synth.make_session), concatenated in HBM.  Returns (frames tensor, files,
    per-file frame counts, vad lists)."""
    torch = pytest.importorskip('torch')
    sd = pkg('synth_device')
    pipeline = pkg('pipeline')
    recipe = pkg('recipe')
    parts, files, vads, off = [], [], [], 0
    for i in range(n_files):
        feats, vad, _ = sd.make_session_device(seed0 + i, seconds, nspk, device='cuda')
        v = [(float(recipe.py2_float_str(a / 125.0)), float(recipe.py2_float_str(b / 125.0))) for a, b in vad]
        files.append(pipeline.BatchFile(off, feats.shape[0], v))
        vads.append(vad)
        parts.append(feats)
        off += feats.shape[0]
    return torch.cat(parts), files, vads


def test_device_generator_is_bit_identical_to_the_host_one():
    pytest.importorskip('torch')
    synth = pkg('synth')
    sd = pkg('synth_device')
    for seed, secs, k, kw in ((4242, 150, 3, {}), (31337, 240, 6, dict(min_turn=1.5, max_turn=4.0))):
        f, v, t = synth.make_session(seed, secs, k, **kw)
        fd, vd, td = sd.make_session_device(seed, secs, k, device='cuda', **kw)
        assert synth.fea_sha256(f) == synth.fea_sha256(fd.cpu().numpy())
        assert v == vd and t == td


def test_batch_of_64_one_hour_files(tmp_path):
    """BASELINE.json config 4 at its size on one GPU: 64 distinct 1 h / 4-speaker files in
    one launch per stage.  Every file's rows equal those of the same file processed alone
    (batch == single file), the fused single-read mode gives the same rows, and two of the
    files are checked against the C oracle through the file-based scripts."""
    torch = pytest.importorskip('torch')
    from oracle.c_engine import COracleEngine
    hipabi = pkg('hipabi')
    pipeline = pkg('pipeline')
    synth = pkg('synth')
    frames, files, vads = _device_batch(64, 3600, 4, 640000)
    total = int(frames.shape[0])
    torch.cuda.synchronize()
    ctx = hipabi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        tm = {}
        got = pipeline.diarize_batch(ctx, frames.data_ptr(), total, files, timings=tm)
        fused = pipeline.diarize_batch(ctx, frames.data_ptr(), total, files, fused=True, timings=tm)
        # the single read is the rule: only segments whose edge the 12-digit text round trip
        # moved across a frame boundary (about 1 %) are summed from the frames again
        assert tm['stats_recomputed'] <= tm['stats_sets'] // 20
        for i, f in enumerate(files):
            assert got[i].shape[0] > 200 and np.array_equal(got[i], fused[i]), i
            alone = pipeline.diarize_batch(ctx, frames.data_ptr(), total, [f])[0]
            assert np.array_equal(got[i], alone), i
    finally:
        ctx.close()
    for i in (5, 41):
        f = files[i]
        feats = frames[f.frame_off:f.frame_off + f.n_frames].cpu().numpy()
        tmp = os.path.join(str(tmp_path), 'f%d' % i)
        os.makedirs(os.path.join(tmp, 'fea'))
        synth.write_fea(os.path.join(tmp, 'fea', 'x.fea'), feats)
        with open(os.path.join(tmp, 'vad.recipe'), 'w') as fh:
            fh.write(synth.vad_recipe_text('x.wav', vads[i]))
        _, final, _, _ = _run_scripts(tmp, COracleEngine(), 'orc')
        rows = re.findall(r'start-time=(\S+) end-time=(\S+) speaker=speaker_(\d+)', final)
        want = [(float(a), float(b), int(c)) for a, b, c in rows]
        assert [(a, b, int(c)) for a, b, c in got[i].tolist()] == want, i


def test_ten_hour_file_matches_c_oracle(tmp_path):
    """BASELINE.json config 5's input on one GPU: a single 10 h / 8-speaker recording
    (~4 000 segments, ~8 M initial pairs).  Change detection and clustering (launch shape
    picked by SPKD_AHC_AUTO: the chip-wide form) against the C oracle: recipes and merge
    sequence identical, merge distances within 1e-7 (of the distance; 1e-13 of its terms)."""
    torch = pytest.importorskip('torch')
    from oracle.c_engine import COracleEngine
    synth = pkg('synth')
    sd = pkg('synth_device')
    feats_d, vad, _ = sd.make_session_device(101010, 36000, 8, device='cuda')
    feats = feats_d.cpu().numpy()
    del feats_d
    tmp = str(tmp_path)
    os.makedirs(os.path.join(tmp, 'fea'))
    synth.write_fea(os.path.join(tmp, 'fea', 'day.fea'), feats)
    with open(os.path.join(tmp, 'vad.recipe'), 'w') as f:
        f.write(synth.vad_recipe_text('day.wav', vad))
    e = pkg('engine').HipEngine(0)
    try:
        h = _run_scripts(tmp, e, 'hip')
    finally:
        e.close()
    o = _run_scripts(tmp, COracleEngine(), 'orc')
    assert h[0].count('\n') > 3500
    assert h[0] == o[0]
    assert h[1] == o[1]
    mh, mo = _merge_lines(h[3]), _merge_lines(o[3])
    assert len(mh) > 3000 and [(a, b) for a, b, _ in mh] == [(a, b) for a, b, _ in mo]
    # clusters reach millions of frames here: a distance of a few thousand is the difference
    # of terms of 1e8, so 1e-8 relative on the distance is 1e-13 on the terms (bar: 1e-5)
    worst = max(abs(x[2] - y[2]) / max(1.0, abs(y[2])) for x, y in zip(mh, mo))
    assert worst < 1e-7, worst
    # ... and pinned against what it is a difference OF (the bound above was 1e-8 until a run
    # measured 1.26e-8): every merge distance against the size of its own terms,
    #   0.5 N |log det S| + 0.5 N1 |log det S1| + 0.5 N2 |log det S2| + penalty,
    # the terms taken from the oracle's pair function on the clusters the merge log replays.
    # Measured: 1.0e-11 of the terms (bound here 1e-10; the parity bar is 1e-5 of the distance).
    # It is not 1e-12: clusters of 10^6 frames are sums of 10^6 products accumulated in fp64 in
    # two different orders (per-segment sweeps merged pairwise here, per-segment sums added in
    # merge order in the oracle), ~ sqrt(N) roundings apart, times the condition of a 39x39
    # covariance in the log det -- neither side is the exact value.
    scales = _merge_term_scales(feats, h[0], [(a, b) for a, b, _ in mo])
    worst_t = max(abs(x[2] - y[2]) / t for x, y, (t, _) in zip(mh, mo, scales))
    worst_big = max([abs(x[2] - y[2]) / t for x, y, (t, nmin) in zip(mh, mo, scales) if nmin >= 400] or [0.0])
    assert worst_t < 1e-10, worst_t
    print('10 h: %d turns, %d merges, worst merge-distance rel err %.2g (%.2g of its terms; %.2g where both '
          'sides hold >= 400 frames)' % (h[0].count('\n'), len(mh), worst, worst_t, worst_big))
