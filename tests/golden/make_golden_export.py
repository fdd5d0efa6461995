#!/usr/bin/env python3
"""Golden vectors for the recipe consumers aku2ann.py and clus-performance.py (SURVEY.md
§8(f) row 3), made by executing the REFERENCE scripts through the in-memory py2 loader of
make_golden.py on small synthetic recipes (stored in the fixture as text).

    python tests/golden/make_golden_export.py     # writes tests/golden/export_cases.json
"""
import contextlib
import io
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg            # noqa: E402

fmt = mg.recipe_mod.py2_float_str


def labelled_recipe(seed, n, audio='/d/meeting.wav', nspk=4, gaps=True, tags=True, t0=0.0, jitter=0.0):
    rnd = random.Random(seed)
    t = t0
    lines = []
    for k in range(n):
        if gaps and rnd.random() < 0.4:
            t += rnd.choice([0.25, 0.5, 1.125, 2.0])
        d = rnd.choice([0.8, 1.5, 2.25, 4.0, 7.5, 12.0]) + (rnd.random() * jitter)
        spk = ' speaker=speaker_%d' % rnd.randrange(1, nspk + 1) if tags else ''
        lines.append('audio=%s lna=a_%d start-time=%s end-time=%s%s\n' % (audio, k + 1, fmt(t), fmt(t + d), spk))
        t += d
    return ''.join(lines)


def run(script, files, argv):
    with tempfile.TemporaryDirectory() as tmp:
        for name, text in files.items():
            with open(os.path.join(tmp, name), 'w') as f:
                f.write(text)
        buf = io.StringIO()
        old = os.getcwd()
        os.chdir(tmp)
        try:
            with contextlib.redirect_stdout(buf):
                mg.load_reference(script, '__main__', argv)
        finally:
            os.chdir(old)
        out = None
        if os.path.exists(os.path.join(tmp, 'out.txt')):
            with open(os.path.join(tmp, 'out.txt')) as f:
                out = f.read()
    return {'script': script, 'files': files, 'argv': argv, 'stdout': buf.getvalue(), 'output_file': out}


def main():
    cases = []
    r1 = labelled_recipe(1, 25)
    cases.append(dict(run('aku2ann.py', {'in.recipe': r1}, ['in.recipe', '-o', 'out.txt']), name='ann_basic'))
    two = labelled_recipe(2, 6, audio='a.wav') + 'garbage line\n' + labelled_recipe(3, 5, audio='b.wav', tags=False)
    cases.append(dict(run('aku2ann.py', {'in.recipe': two}, ['in.recipe', '-o', 'out.txt']), name='ann_two_files_bad_line'))
    cases.append(dict(run('aku2ann.py', {'in.recipe': r1}, ['in.recipe']), name='ann_to_stdout'))
    base = labelled_recipe(10, 40)
    cases.append(dict(run('clus-performance.py', {'b.recipe': base, 'p.recipe': base}, ['b.recipe', 'p.recipe']),
                      name='der_identical'))
    renamed = base.replace('speaker_1', 'X').replace('speaker_2', 'speaker_1').replace('X', 'speaker_2')
    cases.append(dict(run('clus-performance.py', {'b.recipe': base, 'p.recipe': renamed}, ['b.recipe', 'p.recipe']),
                      name='der_renamed_labels'))
    other = labelled_recipe(11, 45, nspk=3, jitter=0.37)
    cases.append(dict(run('clus-performance.py', {'b.recipe': base, 'p.recipe': other},
                          ['b.recipe', 'p.recipe', '-o', 'out.txt', '-t', '0.5']), name='der_different_segmentation'))
    shifted = labelled_recipe(10, 40, t0=0.3)
    cases.append(dict(run('clus-performance.py', {'b.recipe': base, 'p.recipe': shifted}, ['b.recipe', 'p.recipe']),
                      name='der_shifted'))
    untagged = labelled_recipe(12, 30, tags=False)
    cases.append(dict(run('clus-performance.py', {'b.recipe': base, 'p.recipe': untagged}, ['b.recipe', 'p.recipe']),
                      name='der_untagged_proposal'))
    multi = labelled_recipe(13, 12, audio='x.wav') + labelled_recipe(14, 9, audio='y.wav')
    cases.append(dict(run('clus-performance.py', {'b.recipe': multi, 'p.recipe': labelled_recipe(15, 14, audio='x.wav')},
                          ['b.recipe', 'p.recipe']), name='der_two_audio_files_quirk'))
    with open(os.path.join(HERE, 'export_cases.json'), 'w') as f:
        json.dump({'generator': 'tests/golden/make_golden_export.py', 'cases': cases}, f, indent=1, sort_keys=True)
    print('wrote %d cases' % len(cases))
    for c in cases:
        print(c['name'], repr(c['stdout'][-90:]))


if __name__ == '__main__':
    main()
