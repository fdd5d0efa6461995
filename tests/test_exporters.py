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


def test_aku2elan_document_structure(tmp_path):
    """aku2elan.py through the standard library (the reference writes through lxml, which
    is not installed here: PARITY UNPINNED, see exporters.py).  Checked against the
    reference's source semantics: two millisecond time slots per recipe line (truncated),
    one alignable annotation per line referring to them, the speaker as its value."""
    import io
    import xml.etree.ElementTree as ET
    ex = pkg('exporters')
    case = next(c for c in GOLD['cases'] if c['script'] == 'aku2ann.py')
    recipe_text = max(case['files'].values(), key=len)          # the recipe the aku2ann golden converts
    rin = os.path.join(str(tmp_path), 'in.recipe')
    with open(rin, 'w') as f:
        f.write(recipe_text)
    out = os.path.join(str(tmp_path), 'out.eaf')
    say = io.StringIO()
    ex.main_aku2elan([rin, '-o', out], stdout=say, date='2026-10-04T12:00:00+00:00')
    assert say.getvalue().splitlines()[0] == 'Reading recipe from: ' + rin
    doc = ET.parse(out).getroot()
    lines = ex.parse_recipe_ann(recipe_text.splitlines(), lambda *a: None)
    assert doc.tag == 'ANNOTATION_DOCUMENT' and doc.get('FORMAT') == '2.7' and doc.get('DATE').startswith('2026-10-04')
    assert doc.find('HEADER/MEDIA_DESCRIPTOR').get('MEDIA_URL') == 'file://' + lines[0][0]
    assert doc.find('HEADER/PROPERTY').text == str(len(lines))
    slots = doc.findall('TIME_ORDER/TIME_SLOT')
    assert [s.get('TIME_SLOT_ID') for s in slots] == ['ts%d' % (k + 1) for k in range(2 * len(lines))]
    want = []
    for l in lines:
        want += [str(int(l[2] * 1000)), str(int(l[3] * 1000))]
    assert [s.get('TIME_VALUE') for s in slots] == want
    anns = doc.findall('TIER/ANNOTATION/ALIGNABLE_ANNOTATION')
    assert len(anns) == len(lines) and doc.find('TIER').get('TIER_ID') == 'Speakers'
    for n, (a, l) in enumerate(zip(anns, lines), 1):
        assert (a.get('ANNOTATION_ID'), a.get('TIME_SLOT_REF1'), a.get('TIME_SLOT_REF2')) == (
            'a%d' % n, 'ts%d' % (2 * n - 1), 'ts%d' % (2 * n))
        v = a.find('ANNOTATION_VALUE')
        assert (v.text if v is not None else '') == l[4]
