"""Reference-direct goldens AT LENGTH (SURVEY.md §8(c) "oracle cost" row): the true
reference's `spk-change-detection.py` (DIA2 flags) and `spk-clustering.py -m hi -l 1.3`
executed on the one-hour sessions of tests/test_full_size.py (tests/golden/
make_golden_r03.py: seed 5150 / 4 speakers, seed 8088 / 8 speakers; CD ~1 min, CL1
24 - 60 min of CPU each).  tests/golden/hour_<seed>.json hold the outputs only.

  * not gpu: the C oracle reproduces them (recipes byte for byte, every printed number --
    the 37x merge distances included -- to 1e-9): clusters of 10^5 frames and 380-merge
    sequences are pinned by the reference itself, not only transitively;
  * gpu: the HIP path through the same command lines, same bar.
"""
import glob
import importlib
import json
import os

import pytest

from helpers import ROOT, run_case, assert_stdout_close


def _hour_cases():
    out = []
    for path in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'hour_*.json'))):
        with open(path) as f:
            d = json.load(f)
        if 'cd' in d:
            out.append(dict(d['cd'], session=d['session']))
        if 'cl1' in d:
            out.append(dict(d['cl1'], session=d['session'], input_recipe=d['cd']['output_recipe']))
    return out


CASES = _hour_cases()


def _check(case, tmp_path, engine):
    status, stdout, recipe, seg = run_case(case, tmp_path, engine)
    assert status == case['status'] == 'ok'
    assert recipe == case['output_recipe']              # byte for byte: boundaries, labels, numbering
    assert_stdout_close(stdout, case['stdout'], rel=1e-9)
    return recipe


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_c_oracle_matches_the_reference_at_one_hour(case, tmp_path):
    from oracle.c_engine import COracleEngine
    _check(case, tmp_path, COracleEngine())


@pytest.fixture(scope='module')
def eng():
    engine = importlib.import_module('speaker-diarization_amd.engine')
    e = engine.HipEngine(0)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_hip_matches_the_reference_at_one_hour(case, tmp_path, eng):
    recipe = _check(case, tmp_path, eng)
    assert recipe.count('\n') > 300
