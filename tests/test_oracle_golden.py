"""Pins the CPU oracle (oracle/numpy_engine.py) and the host drivers against the
golden vectors produced by the reference scripts themselves."""
import json
import math
import os

import numpy as np
import pytest

from helpers import ROOT, load_cases, run_case, session, assert_stdout_close
from oracle import numpy_engine as ne

CASES = load_cases()


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_c_oracle_cli_case_matches_reference(case, tmp_path):
    """oracle/spkd_oracle.c (fp64 sufficient statistics) against the same goldens:
    recipes byte for byte, printed scores to 1e-9 (KL2: 1e-5, it mixes float32
    means into fp64)."""
    from oracle.c_engine import COracleEngine
    status, stdout, recipe, seg = run_case(case, tmp_path, COracleEngine())
    assert status == case['status']
    assert recipe == case['output_recipe']
    assert seg == case['seg_recipe']
    if case['status'] == 'ok':
        kl2 = 'KL2' in case['argv_tail']
        assert_stdout_close(stdout, case['stdout'], rel=1e-5 if kl2 else 1e-9)


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_cli_case_matches_reference(case, tmp_path):
    status, stdout, recipe, seg = run_case(case, tmp_path, ne.NumpyEngine())
    assert status == case['status']
    assert recipe == case['output_recipe']          # byte for byte
    assert seg == case['seg_recipe']
    if case['status'] == 'ok':
        assert_stdout_close(stdout, case['stdout'])


def test_function_level_scores():
    with open(os.path.join(ROOT, 'tests', 'golden', 'functions.json')) as f:
        g = json.load(f)
    feats, _, _ = session(g['session'])
    lam = g['lambda']
    for p in g['pairs']:
        x, y = feats[p['a'][0]:p['a'][1]], feats[p['b'][0]:p['b'][1]]
        want = {k: float.fromhex(v) for k, v in p.items() if k not in ('a', 'b')}
        got_b, got_g, got_k = ne.bic_frames(x, y, lam), ne.glr_frames(x, y), ne.kl2_frames(x, y)
        for key, got in (('cd_bic_l1.3', got_b), ('cl_bic_l1.3', got_b), ('cl2_bic_l1.3', got_b),
                         ('glr', got_g), ('cd_glr', got_g), ('kl2', got_k), ('cd_kl2', got_k)):
            w = want[key]
            if math.isnan(w):
                assert math.isnan(got)
            else:
                assert got == w, (key, p['a'], p['b'], got, w)    # same library calls: exact


def test_degenerate_inputs():
    with open(os.path.join(ROOT, 'tests', 'golden', 'functions.json')) as f:
        g = json.load(f)
    feats, _, truth = session(g['session'])
    zero = np.zeros((200, 39), dtype=np.float32)
    const = np.ones((150, 39), dtype=np.float32)
    speech = feats[truth[0][0]:truth[0][0] + 300]
    arrs = {'speech_zero': (speech, zero), 'zero_zero': (zero, zero), 'const_zero': (const, zero)}
    for rec in g['degenerate']:
        x, y = arrs[rec['name']]
        for key, fn in (('bic', lambda: ne.bic_frames(x, y, 1.3)), ('glr', lambda: ne.glr_frames(x, y))):
            w = float.fromhex(rec[key]) if rec[key] not in ('ValueError',) else None
            got = fn()
            if w is None:
                continue
            assert (math.isnan(w) and math.isnan(got)) or got == w


def test_two_window_bic_function_goldens():
    """The sliding-window geometry with BIC: the reference command line crashes
    (SURVEY.md A-6), its bic() function on the same windows is the oracle
    (tests/golden/functions_sw.json).  Both CPU oracles must reproduce it."""
    from oracle.c_engine import COracleEngine
    with open(os.path.join(ROOT, 'tests', 'golden', 'functions_sw.json')) as f:
        g = json.load(f)
    feats, _, _ = session(g['session'])
    engines = [ne.NumpyEngine(), COracleEngine()]
    for e in engines:
        e.set_features(feats)
    for w in g['windows']:
        want = np.array([float.fromhex(v) for v in w['bic']])
        assert len(want) > 3
        for e, tol in zip(engines, (0.0, 1e-9)):
            got = e.sw([tuple(w['turn'])], 'BIC', w['lambda'], w['winsize'], w['winstep'])[0]
            assert len(got) == len(want)
            assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) <= tol
