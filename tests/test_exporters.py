"""aku2ann / clus-performance drop-ins against outputs of the
reference's own scripts (tests/golden/export_cases.json, made by make_golden_export.py)."""
import io
import json
import os

import pytest

from conftest import pkg
from helpers import ROOT

with open(os.path.join(ROOT, 'tests', 'golden', 'export_cases.json')) as _f:
    GOLD = json.load(_f)


@pytest.mark.parametrize('case', GOLD['cases'], ids=[c['name'] for c in GOLD['cases']])
def test_cli_case_matches_reference(case, tmp_path, monkeypatch):
    ex = pkg('exporters')
    tmp = str(tmp_path)
    for name, text in case['files'].items():
        with open(os.path.join(tmp, name), 'w') as f:
            f.write(text)
    monkeypatch.chdir(tmp)
    out = io.StringIO()
    main = ex.main_aku2ann if case['script'] == 'aku2ann.py' else ex.main_clus_performance
    main(case['argv'], stdout=out)
    assert out.getvalue() == case['stdout']
    if case['output_file'] is not None:
        with open(os.path.join(tmp, 'out.txt')) as f:
            assert f.read() == case['output_file']


def test_der_of_a_perfect_and_a_relabelled_proposal():
    ex = pkg('exporters')
    base = [(0.0, 2.0, 'a'), (2.5, 4.0, 'b'), (4.0, 9.0, 'a')]
    assert ex.der(base, base)[1] == 0
    swapped = [(s, e, {'a': 'x', 'b': 'y'}[l]) for s, e, l in base]
    assert ex.der(base, swapped)[1] == 0                          # labels are matched, not compared
    merged = [(s, e, 'one') for s, e, _ in base]
    # every BASELINE label picks its most frequent partner, so a proposal that merges two
    # speakers scores no error, while a proposal that splits one does (the scorer's quirk)
    assert ex.der(base, merged)[1] == 0
    assert ex.der(merged, base)[1] == int((4.0 - 2.5) / 0.001)
