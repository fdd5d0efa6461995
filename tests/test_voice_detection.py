"""VAD-recipe producer against outputs of the reference's own
voice-detection2.py (tests/golden/vad_cases.json, made by tests/golden/make_golden_vad.py)."""
import io
import json
import os

import pytest

from conftest import pkg
from helpers import ROOT

with open(os.path.join(ROOT, 'tests', 'golden', 'vad_cases.json')) as _f:
    GOLD = json.load(_f)


@pytest.mark.parametrize('case', GOLD['cases'], ids=[c['name'] for c in GOLD['cases']])
def test_cli_case_matches_reference(case, tmp_path, monkeypatch):
    vd = pkg('voice_detection')
    tmp = str(tmp_path)
    os.makedirs(os.path.join(tmp, 'exp'))
    for base, (text, last) in case['exps'].items():
        with open(os.path.join(tmp, 'exp', base + '.exp'), 'w') as f:
            f.write(text)
        with open(os.path.join(tmp, 'exp', base + '.last_frame'), 'w') as f:
            f.write(last)
    with open(os.path.join(tmp, 'in.recipe'), 'w') as f:
        f.write(case['recipe'])
    monkeypatch.chdir(tmp)
    argv = ['in.recipe', 'exp'] + ([] if case['to_stdout'] else ['-o', 'out.recipe']) + case['argv_tail']
    out = io.StringIO()
    vd.main(argv, stdout=out)
    assert out.getvalue() == case['stdout']                       # byte for byte, recipe lines included
    if case['output_recipe'] is not None:
        with open(os.path.join(tmp, 'out.recipe')) as f:
            assert f.read() == case['output_recipe']


def test_lna_base_names():
    vd = pkg('voice_detection')
    for name, nxt in GOLD['inc_lna'].items():
        assert vd.inc_lna(name) == nxt
    cur = 'a'
    for want in GOLD['inc_lna_chain']:
        assert cur == want
        cur = vd.inc_lna(cur)


def test_output_feeds_the_change_detection_reader(tmp_path):
    """The producer's lines parse with the hot path's recipe reader: same grammar."""
    vd = pkg('voice_detection')
    recipe = pkg('recipe')
    case = GOLD['cases'][0]
    lines = case['output_recipe'].splitlines(True)
    parsed = recipe.parse_recipe(lines, echo=lambda s: None)
    assert len(parsed) == len(lines)
    assert parsed[0][1].startswith('a_1')
    opt = vd.VadOptions()
    toks = [('10', '<w>'), ('500', 'p'), ('520', '<w>'), ('900', 'p')]
    assert vd.turns_from_tokens(toks, 'q', opt, last_frame=lambda: '905') == []       # 5 frames < -ms
    assert vd.turns_from_tokens(toks, 'q', opt, last_frame=lambda: '1000') == [('q_1', 900 / 125.0, 8.0)]
