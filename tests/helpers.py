"""Shared helpers: materialise a golden CLI case in a temp dir and run one of the
product command lines on it with a chosen engine."""
import importlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

cli = importlib.import_module('speaker-diarization_amd.cli')
synth = importlib.import_module('speaker-diarization_amd.synth')

_SESSIONS = {}


def session(meta):
    key = json.dumps(meta, sort_keys=True)
    if key not in _SESSIONS:
        feats, vad, truth = synth.make_session(meta['seed'], meta['seconds'],
                                               meta['n_speakers'], **meta['kwargs'])
        assert synth.fea_sha256(feats) == meta['sha256'], 'synthetic generator is not reproducible here'
        _SESSIONS[key] = (feats, vad, truth)
    return _SESSIONS[key]


def load_cases():
    """Reference command lines: the DIA2-flag set plus the non-default growing-window
    parameters and the sub-half-second window steps (make_golden.py /
    make_golden_params.py / make_golden_r02.py)."""
    cases = []
    for name in ('cli_cases.json', 'cli_cases_params.json', 'cli_cases_r02.json'):
        with open(os.path.join(ROOT, 'tests', 'golden', name)) as f:
            cases += json.load(f)['cases']
    return cases


def run_case(case, tmp, engine):
    """Returns (status, stdout_text, recipe_text, seg_text) with tmp -> <TMP>."""
    tmp = str(tmp)
    feats, _, _ = session(case['session'])
    feadir = os.path.join(tmp, 'fea')
    os.makedirs(feadir, exist_ok=True)
    fea = os.path.join(feadir, os.path.splitext(case['audio'])[0] + '.fea')
    if not os.path.exists(fea):
        synth.write_fea(fea, feats)
    name = case['name']
    rin = os.path.join(tmp, name + '.in.recipe')
    rout = os.path.join(tmp, name + '.out.recipe')
    with open(rin, 'w') as f:
        f.write(case['input_recipe'])
    argv = [rin, feadir + '/', '-o', rout] + [a.replace('<TMP>', tmp) for a in case['argv_tail']]
    buf = io.StringIO()
    status = 'ok'
    try:
        if case['script'] == 'spk-change-detection.py':
            cli.main_change_detection(argv, engine=engine, stdout=buf)
        else:
            variant = 2 if case['script'].endswith('2.py') else 1
            cli.main_clustering(argv, variant=variant, engine=engine, stdout=buf)
    except Exception as ex:
        status = type(ex).__name__ + ': ' + str(ex)
    out = open(rout).read() if os.path.exists(rout) else ''
    segp = os.path.join(tmp, name + '.out-seg.recipe')
    seg = open(segp).read().replace(tmp, '<TMP>') if os.path.exists(segp) else None
    return status, buf.getvalue().replace(tmp, '<TMP>'), out, seg


def split_numbers(text):
    """Tokenise stdout so numeric fields can be compared with a tolerance."""
    toks = []
    for line in text.splitlines():
        row = []
        for w in line.split(' '):
            try:
                row.append(float(w))
            except ValueError:
                row.append(w)
        toks.append(row)
    return toks


def assert_stdout_close(got, want, rel=1e-9):
    g, w = split_numbers(got), split_numbers(want)
    assert len(g) == len(w), 'stdout line count differs:\n%s\n---\n%s' % (got[-2000:], want[-2000:])
    for lg, lw in zip(g, w):
        assert len(lg) == len(lw), (lg, lw)
        for a, b in zip(lg, lw):
            if isinstance(a, float) and isinstance(b, float):
                if a != a or b != b:
                    assert a != a and b != b, (lg, lw)
                elif a in (float('inf'), float('-inf')) or b in (float('inf'), float('-inf')):
                    assert a == b, (lg, lw)
                else:
                    assert abs(a - b) <= rel * max(1.0, abs(a), abs(b)), (lg, lw)
            else:
                assert a == b, (lg, lw)
