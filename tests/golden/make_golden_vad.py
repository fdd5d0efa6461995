#!/usr/bin/env python3
"""Golden vectors for the VAD-recipe producer (voice-detection2.py),
made by executing the REFERENCE script through the in-memory py2 loader of
make_golden.py.  Inputs are synthetic `.exp` token streams from a seeded generator
(stored in the fixture as text, they are tiny); outputs are the recipe and the stdout.

    python tests/golden/make_golden_vad.py        # writes tests/golden/vad_cases.json
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
import make_golden as mg            # noqa: E402  (the loader; reads /root/reference at run time)


def token_stream(seed, n_tokens, rate=125, open_end=False, lines=1):
    """A plausible generate_exp.py output: alternating speech (p) / silence (<w>) marks at
    increasing frame numbers, with some gaps below the usual thresholds and some repeats."""
    rnd = random.Random(seed)
    frame = rnd.randrange(0, 40)
    tok = 'p' if rnd.random() < 0.5 else '<w>'
    items = []
    for _ in range(n_tokens):
        items.append('%d %s' % (frame, tok))
        r = rnd.random()
        if r < 0.25:
            gap = rnd.randrange(1, int(0.3 * rate))          # shorter than the thresholds
        elif r < 0.9:
            gap = rnd.randrange(int(0.3 * rate), 6 * rate)
        else:
            gap = rnd.randrange(6 * rate, 30 * rate)
        frame += gap
        if rnd.random() > 0.12:                               # mostly alternate, sometimes repeat
            tok = '<w>' if tok == 'p' else 'p'
    if open_end and tok == 'p':
        pass
    elif open_end:
        items.append('%d p' % frame)
        frame += rnd.randrange(1, 3 * rate)
    last_frame = frame + rnd.randrange(0, 4 * rate)
    per = max(1, len(items) // lines)
    text = '\n'.join(' '.join(items[i:i + per]) for i in range(0, len(items), per)) + '\n'
    return text, str(last_frame)


def run_case(name, wavs, exps, argv_tail, recipe_text=None, to_stdout=False):
    """wavs: audio names in recipe order; exps: {basename: (exp_text, last_frame_text)}."""
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, 'exp'))
        for base, (text, last) in exps.items():
            with open(os.path.join(tmp, 'exp', base + '.exp'), 'w') as f:
                f.write(text)
            with open(os.path.join(tmp, 'exp', base + '.last_frame'), 'w') as f:
                f.write(last)
        if recipe_text is None:
            recipe_text = ''.join('audio=%s\n' % w for w in wavs)
        with open(os.path.join(tmp, 'in.recipe'), 'w') as f:
            f.write(recipe_text)
        argv = ['in.recipe', 'exp'] + ([] if to_stdout else ['-o', 'out.recipe']) + argv_tail
        status = 'ok'
        buf = io.StringIO()
        old = os.getcwd()
        os.chdir(tmp)
        try:
            with contextlib.redirect_stdout(buf):
                mg.load_reference('voice-detection2.py', '__main__', argv)
        except SystemExit:
            status = 'exit'                                   # the reference called exit()
        finally:
            os.chdir(old)
        stdout = buf.getvalue()
        out = None
        if os.path.exists(os.path.join(tmp, 'out.recipe')):
            with open(os.path.join(tmp, 'out.recipe')) as f:
                out = f.read()
    return {'name': name, 'recipe': recipe_text, 'exps': {k: list(v) for k, v in exps.items()},
            'argv_tail': argv_tail, 'to_stdout': to_stdout, 'status': status, 'stdout': stdout,
            'output_recipe': out}


def main():
    cases = []
    e1 = {'one': token_stream(1, 60), 'two': token_stream(2, 45, lines=3)}
    cases.append(run_case('defaults_two_files', ['/data/one.wav', 'two.wav'], e1, []))
    cases.append(run_case('options', ['/data/one.wav', 'two.wav'], e1,
                          ['-ms', '0.5', '-mns', '0.6', '-sbe', '0.1', '-see', '0.15']))
    e3 = {'r100': token_stream(3, 80, rate=100)}
    cases.append(run_case('rate_100', ['r100.wav'], e3, ['-r', '100']))
    many = ['f%02d.wav' % i for i in range(30)]
    e4 = {'f%02d' % i: token_stream(100 + i, 12) for i in range(30)}
    cases.append(run_case('thirty_files_lna_names', many, e4, []))
    e5 = {'open_ok': token_stream(7, 30, open_end=True), 'open_short': ('10 <w> 500 p 520 <w> 900 p\n', '905')}
    cases.append(run_case('open_last_turn', ['open_ok.wav', 'open_short.wav'], e5, []))
    cases.append(run_case('bad_recipe_line', ['one.wav'], {'one': token_stream(1, 60)}, [],
                          recipe_text='audio=one.wav lna=x start-time=0.0\nno audio here\naudio=one.wav\n'))
    cases.append(run_case('to_stdout', ['two.wav'], {'two': token_stream(2, 45)}, [], to_stdout=True))
    cases.append(run_case('expansion_negative_start', ['early.wav'],
                          {'early': ('0 <w> 3 p 200 <w> 400 p 900 <w> 2000 p\n', '2100')}, ['-sbe', '0.2']))
    cases.append(run_case('missing_exp_file', ['one.wav', 'nothere.wav', 'two.wav'],
                          {'one': token_stream(1, 60), 'two': token_stream(2, 45)}, []))
    ref = mg.load_reference('voice-detection2.py')
    names = ['a', 'b', 'y', 'z', 'az', 'zz', 'azz', 'zzz', 'abz', 'q']
    inc = {n: ref['inc_lna'](n) for n in names}
    chain, cur = [], 'a'
    for _ in range(60):
        chain.append(cur)
        cur = ref['inc_lna'](cur)
    with open(os.path.join(HERE, 'vad_cases.json'), 'w') as f:
        json.dump({'generator': 'tests/golden/make_golden_vad.py', 'reference': 'voice-detection2.py',
                   'cases': cases, 'inc_lna': inc, 'inc_lna_chain': chain}, f, indent=1, sort_keys=True)
    print('wrote %d cases' % len(cases))


if __name__ == '__main__':
    main()
