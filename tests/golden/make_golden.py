#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by executing the REFERENCE
scripts themselves (read from /root/reference at generation time, never copied).

Runs only in the build container: the GPU box has no /root/reference, and the
tests consume the committed JSON, not this script.

How the Python-2 sources are executed under Python 3, in memory only
(SURVEY.md §8c):
  1. lib2to3 ``refactor_string`` (print statements, xrange, sys.maxint);
  2. an ``ast.NodeTransformer`` that routes every ``/`` through a py2-division
     helper (floor for int/int, true division otherwise) and every slice bound
     through an int()-truncating helper (old numpy accepted float bounds);
  3. a module-global ``str``/``print`` pair that formats floats as py2 did
     (12 significant digits);
  4. ``exec`` with ``__name__ == '__main__'`` and a patched ``sys.argv`` to run
     the real command line, or under another name to get at the functions.
No reference logic is edited by hand.  Inputs come from the repo's own
bit-reproducible generator, so a fixture stores (seed, sha256, argv, outputs).
"""
import ast
import contextlib
import importlib
import io
import json
import os
import struct
import sys
import tempfile
import time
import warnings

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('SPKD_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
pkg = importlib.import_module('speaker-diarization_amd')
synth = importlib.import_module('speaker-diarization_amd.synth')
recipe_mod = importlib.import_module('speaker-diarization_amd.recipe')

warnings.filterwarnings('ignore')


# --------------------------------------------------------------------------- loader
def _py2_div(a, b):
    ints = (int, np.integer)
    if isinstance(a, ints) and isinstance(b, ints) and not isinstance(a, bool):
        return a // b
    return a / b


def _py2_bound(x):
    if x is None:
        return None
    if isinstance(x, (float, np.floating)):
        return int(x)
    return x


class _Py2Semantics(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            return ast.copy_location(ast.Call(
                func=ast.Name(id='__py2_div', ctx=ast.Load()),
                args=[node.left, node.right], keywords=[]), node)
        return node

    def visit_Slice(self, node):
        self.generic_visit(node)
        for fld in ('lower', 'upper'):
            v = getattr(node, fld)
            if v is not None:
                setattr(node, fld, ast.copy_location(ast.Call(
                    func=ast.Name(id='__py2_bound', ctx=ast.Load()),
                    args=[v], keywords=[]), v))
        return node


def _py2_print(*args, **kw):
    out = kw.get('file', sys.stdout)
    out.write(' '.join(_py2_str(a) for a in args) + kw.get('end', '\n'))


def _py2_str(x=''):
    if isinstance(x, tuple):
        return '(' + ', '.join(repr(v) for v in x) + (',)' if len(x) == 1 else ')')
    return recipe_mod.py2_str(x)


def load_reference(script, name='ref_module', argv=None):
    """Returns the namespace dict after executing the transformed script."""
    from lib2to3 import refactor
    path = os.path.join(REF, script)
    with open(path) as f:
        src = f.read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    src3 = str(tool.refactor_string(src, script))
    tree = _Py2Semantics().visit(ast.parse(src3, filename=script))
    ast.fix_missing_locations(tree)
    code = compile(tree, script, 'exec')
    ns = {'__name__': name, '__py2_div': _py2_div, '__py2_bound': _py2_bound,
          'str': _py2_str, 'print': _py2_print, '__file__': path}
    old_argv = sys.argv
    try:
        if argv is not None:
            sys.argv = [script] + list(argv)
        exec(code, ns)
    finally:
        sys.argv = old_argv
    return ns


def run_cli(script, argv, cwd):
    """Run the reference command line; returns captured stdout text."""
    buf = io.StringIO()
    old = os.getcwd()
    os.chdir(cwd)
    try:
        with contextlib.redirect_stdout(buf):
            load_reference(script, '__main__', argv)
    finally:
        os.chdir(old)
    return buf.getvalue()


# --------------------------------------------------------------------------- helpers
def hexf(x):
    return float(x).hex()


def session_meta(seed, seconds, nspk, **kw):
    feats, vad, truth = synth.make_session(seed, seconds, nspk, **kw)
    meta = {'seed': seed, 'seconds': seconds, 'n_speakers': nspk, 'kwargs': kw,
            'frames': int(feats.shape[0]), 'sha256': synth.fea_sha256(feats)}
    return feats, vad, truth, meta


class Args(object):
    pass


def function_level():
    """bic/glr/kl2 of both scripts on fixed array pairs (scores as hex floats)."""
    feats, vad, truth, meta = session_meta(7001, 400, 4)
    cd = load_reference('spk-change-detection.py')
    cl = load_reference('spk-clustering.py')
    cl2 = load_reference('spk-clustering2.py')
    a = Args(); a.lambdac = 1.3; a.tt = False; a.dlr = False
    for ns in (cd, cl, cl2):
        ns['args'] = a
    segs = [(s, e) for (s, e, k) in truth]
    pairs = []
    # (rangeA, rangeB): adjacent same/different speakers, short, long, tiny
    t = truth
    pairs.append((segs[0], segs[1]))
    pairs.append((segs[1], segs[2]))
    pairs.append(((t[0][0], t[0][0] + 50), (t[0][0] + 50, t[0][0] + 113)))
    pairs.append(((t[2][0], t[2][0] + 125), (t[2][0] + 125, t[2][0] + 250)))
    same = [x for x in t if x[2] == t[0][2]]
    pairs.append(((same[0][0], same[0][1]), (same[1][0], same[1][1])))
    pairs.append(((t[3][0], t[3][0] + 400), (t[5][0], t[5][0] + 900)))
    pairs.append(((t[4][0], t[4][0] + 62), (t[4][0] + 62, t[4][0] + 1000)))
    pairs.append(((0, 20000), (20000, 50000)))
    pairs.append(((t[6][0], t[6][0] + 45), (t[7][0], t[7][0] + 300)))   # n close to d
    pairs.append(((t[6][0], t[6][0] + 30), (t[7][0], t[7][0] + 300)))   # n < d: rank deficient
    out = []
    for (a0, a1), (b0, b1) in pairs:
        x, y = feats[a0:a1], feats[b0:b1]
        xy = np.concatenate((x, y))
        rec = {'a': [int(a0), int(a1)], 'b': [int(b0), int(b1)]}
        with np.errstate(all='ignore'):
            rec['cd_bic_l1.3'] = hexf(cd['bic'](x, y, xy, 0, {}))
            rec['cl_bic_l1.3'] = hexf(cl['bic'](x, y))
            rec['cl2_bic_l1.3'] = hexf(cl2['bic'](x, y))
            rec['glr'] = hexf(cl['glr'](x, y))
            rec['cd_glr'] = hexf(cd['glr'](x, y))
            rec['kl2'] = hexf(cl['kl2'](x, y))
            rec['cd_kl2'] = hexf(cd['kl2'](x, y))
        out.append(rec)
    # degenerate inputs
    zero = np.zeros((200, 39), dtype=np.float32)
    const = np.ones((150, 39), dtype=np.float32)
    speech = feats[t[0][0]:t[0][0] + 300]
    deg = []
    for nm, x, y in (('speech_zero', speech, zero), ('zero_zero', zero, zero),
                     ('const_zero', const, zero)):
        with np.errstate(all='ignore'):
            try:
                v = hexf(cl['bic'](x, y))
            except ValueError as ex:
                v = 'ValueError'
            try:
                g = hexf(cl['glr'](x, y))
            except ValueError as ex:
                g = 'ValueError'
        deg.append({'name': nm, 'bic': v, 'glr': g})
    # frozen-c1 behaviour of the 5-argument CD bic called without i/saved (A-8)
    x1, y1 = feats[segs[0][0]:segs[0][1]], feats[segs[1][0]:segs[1][1]]
    x2, y2 = feats[segs[2][0]:segs[2][1]], feats[segs[3][0]:segs[3][1]]
    cd_fresh = load_reference('spk-change-detection.py')
    cd_fresh['args'] = a
    f1 = cd_fresh['bic'](x1, y1, np.concatenate((x1, y1)))
    f2 = cd_fresh['bic'](x2, y2, np.concatenate((x2, y2)))
    frozen = {'first': [list(map(int, segs[0])), list(map(int, segs[1]))],
              'second': [list(map(int, segs[2])), list(map(int, segs[3]))],
              'first_score': hexf(f1), 'second_score_frozen': hexf(f2)}
    return {'session': meta, 'lambda': 1.3, 'pairs': out, 'degenerate': deg,
            'frozen_c1': frozen}


def cli_case(name, feats, meta, recipe_text, script, argv_tail, tmp, audio='synth.wav'):
    """Run one reference CLI; returns a dict with argv, stdout, output recipe."""
    feadir = os.path.join(tmp, 'fea')
    os.makedirs(feadir, exist_ok=True)
    fea = os.path.join(feadir, os.path.splitext(audio)[0] + '.fea')
    if not os.path.exists(fea):
        synth.write_fea(fea, feats)
    rin = os.path.join(tmp, name + '.in.recipe')
    rout = os.path.join(tmp, name + '.out.recipe')
    with open(rin, 'w') as f:
        f.write(recipe_text)
    argv_tail = [a.replace('<TMP>', tmp) for a in argv_tail]
    argv = [rin, feadir + '/', '-o', rout] + list(argv_tail)
    t0 = time.time()
    status = 'ok'
    try:
        stdout = run_cli(script, argv, tmp)
    except Exception as ex:           # the reference itself crashes on some modes (A-6)
        stdout = ''
        status = type(ex).__name__ + ': ' + str(ex)
    dt = time.time() - t0
    out = ''
    if os.path.exists(rout):
        with open(rout) as f:
            out = f.read()
    stdout = stdout.replace(tmp, '<TMP>')
    seg_out = None
    for cand in (os.path.join(tmp, name + '.out-seg.recipe'),):
        if os.path.exists(cand):
            with open(cand) as f:
                seg_out = f.read().replace(tmp, '<TMP>')
    argv_tail = [a.replace(tmp, '<TMP>') for a in argv_tail]
    print('  %-28s %-24s %6.1fs  %d lines  %s' % (name, script, dt, out.count('\n'), status))
    return {'name': name, 'script': script, 'argv_tail': list(argv_tail), 'audio': audio,
            'session': meta, 'input_recipe': recipe_text, 'output_recipe': out,
            'seg_recipe': seg_out, 'stdout': stdout, 'status': status, 'ref_seconds': round(dt, 2)}


DIA2_CD = ['-m', 'gw', '-d', 'BIC', '-w', '1.0', '-st', '3.0', '-dws', '0.1', '-l', '1.0']
DIA2_CL = ['-m', 'hi', '-l', '1.3']


def pipeline_cases(tmp):
    cases = []
    # --- session A: 150 s, 3 speakers
    fa, va, ta, ma = session_meta(4242, 150, 3)
    vad_a = synth.vad_recipe_text('/data/audio/synth.wav', va)
    c = cli_case('A_cd_gw_bic', fa, ma, vad_a, 'spk-change-detection.py', DIA2_CD, tmp)
    cases.append(c)
    spkc_a = c['output_recipe']
    cases.append(cli_case('A_cd_gw_bic_tt', fa, ma, vad_a, 'spk-change-detection.py', DIA2_CD + ['-tt'], tmp))
    cases.append(cli_case('A_cd_gw_glr', fa, ma, vad_a, 'spk-change-detection.py',
                          ['-m', 'gw', '-d', 'GLR', '-w', '1.0', '-st', '3.0', '-dws', '0.1', '-t', '900'], tmp))
    cases.append(cli_case('A_cd_gw_kl2', fa, ma, vad_a, 'spk-change-detection.py',
                          ['-m', 'gw', '-d', 'KL2', '-w', '1.0', '-st', '3.0', '-dws', '0.1', '-t', '60'], tmp))
    cases.append(cli_case('A_cd_sw_glr', fa, ma, vad_a, 'spk-change-detection.py', ['-t', '3000'], tmp))
    cases.append(cli_case('A_cd_sw_glr_tt', fa, ma, vad_a, 'spk-change-detection.py', ['-t', '3000', '-tt'], tmp))
    cases.append(cli_case('A_cd_sw_kl2', fa, ma, vad_a, 'spk-change-detection.py',
                          ['-m', 'sw', '-d', 'KL2', '-w', '2.0', '-st', '0.25', '-t', '25'], tmp))
    cases.append(cli_case('A_cd_sw_bic_crash', fa, ma, vad_a, 'spk-change-detection.py',
                          ['-m', 'sw', '-d', 'BIC'], tmp))
    cases.append(cli_case('A_cd_gw_bic_dlr_seg', fa, ma, vad_a, 'spk-change-detection.py',
                          DIA2_CD + ['-dlr', '-seg', '<TMP>/'], tmp))
    cases.append(cli_case('A_cd_m_bic', fa, ma, spkc_a, 'spk-change-detection.py', ['-m', 'm', '-d', 'BIC', '-l', '1.3'], tmp))
    cases.append(cli_case('A_cd_m_glr', fa, ma, spkc_a, 'spk-change-detection.py', ['-m', 'm', '-d', 'GLR', '-t', '1500'], tmp))
    cases.append(cli_case('A_cd_m_kl2', fa, ma, spkc_a, 'spk-change-detection.py', ['-m', 'm', '-d', 'KL2', '-t', '30'], tmp))
    cases.append(cli_case('A_cl1_hi_bic', fa, ma, spkc_a, 'spk-clustering.py', DIA2_CL, tmp))
    cases.append(cli_case('A_cl2_hi_bic', fa, ma, spkc_a, 'spk-clustering2.py', DIA2_CL, tmp))
    cases.append(cli_case('A_cl1_hi_glr', fa, ma, spkc_a, 'spk-clustering.py', ['-m', 'hi', '-d', 'GLR', '-t', '1500'], tmp))
    cases.append(cli_case('A_cl1_hi_kl2', fa, ma, spkc_a, 'spk-clustering.py', ['-m', 'hi', '-d', 'KL2', '-t', '30'], tmp))
    cases.append(cli_case('A_cl1_in_bic', fa, ma, spkc_a, 'spk-clustering.py', ['-m', 'in', '-l', '1.3'], tmp))
    cases.append(cli_case('A_cl1_in_bic_tt', fa, ma, spkc_a, 'spk-clustering.py', ['-m', 'in', '-l', '1.3', '-tt'], tmp))
    cases.append(cli_case('A_cl2_in_bic', fa, ma, spkc_a, 'spk-clustering2.py', ['-m', 'in', '-l', '1.3'], tmp))
    cases.append(cli_case('A_cl1_hi_bic_ms2', fa, ma, spkc_a, 'spk-clustering.py', DIA2_CL + ['-ms', '2'], tmp))
    cases.append(cli_case('A_cl2_hi_bic_ms2', fa, ma, spkc_a, 'spk-clustering2.py', DIA2_CL + ['-ms', '2'], tmp))
    cases.append(cli_case('A_cl1_hi_bic_l3_dlr', fa, ma, spkc_a, 'spk-clustering.py', ['-m', 'hi', '-l', '3.0', '-dlr'], tmp))
    cases.append(cli_case('A_cl1_hi_bic_seg', fa, ma, spkc_a, 'spk-clustering.py', DIA2_CL + ['-seg', '<TMP>'], tmp))
    cases.append(cli_case('A_cl2_hi_bic_seg', fa, ma, spkc_a, 'spk-clustering2.py', DIA2_CL + ['-seg', '<TMP>/'], tmp))
    # --- session B: 420 s, 4 speakers (more merges, longer clusters)
    tmpb = os.path.join(tmp, 'B'); os.makedirs(tmpb)
    fb, vb, tb, mb = session_meta(9090, 420, 4)
    vad_b = synth.vad_recipe_text('meeting.wav', vb)
    c = cli_case('B_cd_gw_bic', fb, mb, vad_b, 'spk-change-detection.py', DIA2_CD, tmpb, audio='meeting.wav')
    cases.append(c)
    spkc_b = c['output_recipe']
    cases.append(cli_case('B_cl1_hi_bic', fb, mb, spkc_b, 'spk-clustering.py', DIA2_CL, tmpb, audio='meeting.wav'))
    cases.append(cli_case('B_cl2_hi_bic', fb, mb, spkc_b, 'spk-clustering2.py', DIA2_CL, tmpb, audio='meeting.wav'))
    # --- session C: 6 similar speakers, short turns: many near-threshold merges
    tmpc = os.path.join(tmp, 'C'); os.makedirs(tmpc)
    fc, vc, tc, mc = session_meta(31337, 240, 6, min_turn=1.5, max_turn=4.0)
    vad_c = synth.vad_recipe_text('c.wav', vc)
    c = cli_case('C_cd_gw_bic', fc, mc, vad_c, 'spk-change-detection.py', DIA2_CD, tmpc, audio='c.wav')
    cases.append(c)
    spkc_c = c['output_recipe']
    cases.append(cli_case('C_cl1_hi_bic', fc, mc, spkc_c, 'spk-clustering.py', DIA2_CL, tmpc, audio='c.wav'))
    cases.append(cli_case('C_cl2_hi_bic', fc, mc, spkc_c, 'spk-clustering2.py', DIA2_CL, tmpc, audio='c.wav'))
    return cases


def main():
    which = sys.argv[1:] or ['functions', 'cli']
    env = {'python': sys.version.split()[0], 'numpy': np.__version__, 'scipy': scipy.__version__,
           'note': 'generated by executing the reference scripts via tests/golden/make_golden.py'}
    if 'functions' in which:
        print('function-level goldens ...')
        data = function_level()
        data['env'] = env
        with open(os.path.join(HERE, 'functions.json'), 'w') as f:
            json.dump(data, f, indent=1)
    if 'cli' in which:
        print('CLI goldens ...')
        with tempfile.TemporaryDirectory() as tmp:
            cases = pipeline_cases(tmp)
        with open(os.path.join(HERE, 'cli_cases.json'), 'w') as f:
            json.dump({'env': env, 'cases': cases}, f, indent=1)


if __name__ == '__main__':
    main()
