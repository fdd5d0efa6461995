"""The py3 counterpart of spk-diarization2.py (BASELINE.json config 1, "plumbing only").
The two AaltoASR-backed producers it calls -- ./generate_exp.py and feacat -- are not
available anywhere offline, so the tests put stand-ins of those names into a scratch
working directory: they write the `.exp` token stream / the feacat `.fea` bytes of a
synthetic session whose VAD recipe, change-detection recipe and speaker recipe the
reference's own scripts have produced (golden session B, `meeting.wav`).  Everything
downstream of them is the repository's drop-in executables, run as child processes
with the reference's argv lists and cwd-relative names."""
import io
import json
import os
import stat
import subprocess
import sys

import pytest

from conftest import pkg
from helpers import ROOT, load_cases

SCRIPTS = ['voice-detection2.py', 'spk-change-detection.py', 'spk-clustering.py', 'aku2ann.py', 'aku2elan.py',
           'spk-diarization2.py']

STUB_GENERATE_EXP = '''#!/usr/bin/env python3
# stand-in for the reference's generate_exp.py (AaltoASR decoder): same argv
# (recipe -e exppath -l lnapath), writes <base>.exp and <base>.last_frame
import json, os, re, sys
sys.path.insert(0, %(root)r)
import importlib
synth = importlib.import_module('speaker-diarization_amd.synth')
recipe, exppath = sys.argv[1], sys.argv[sys.argv.index('-e') + 1]
meta = json.load(open(%(meta)r))
for line in open(recipe):
    wav = re.search(r'audio=(\\S+)', line).group(1)
    base = os.path.splitext(os.path.basename(wav))[0]
    total, pieces, vad, truth = synth.plan_session(meta['seed'], meta['seconds'], meta['n_speakers'], **meta['kwargs'])
    toks = ['0 <w>'] + ['%%d p %%d <w>' %% (a, b) for a, b in vad]
    open(os.path.join(exppath, base + '.exp'), 'w').write(' '.join(toks) + '\\n')
    open(os.path.join(exppath, base + '.last_frame'), 'w').write(str(total))
'''

STUB_FEACAT = '''#!/usr/bin/env python3
# stand-in for feacat -c cfg -H --raw-output wav: the .fea bytes on stdout
import json, sys
sys.path.insert(0, %(root)r)
import importlib
import numpy as np
synth = importlib.import_module('speaker-diarization_amd.synth')
meta = json.load(open(%(meta)r))
feats, _, _ = synth.make_session(meta['seed'], meta['seconds'], meta['n_speakers'], **meta['kwargs'])
out = sys.stdout.buffer
out.write(np.array([feats.shape[1]], dtype='<i4').tobytes())
out.write(np.ascontiguousarray(feats, dtype='<f4').tobytes())
'''


def _scratch_cwd(tmp, meta):
    for d in ('lna', 'exp', 'fea', 'tmp', 'bin'):
        os.makedirs(os.path.join(tmp, d))
    for name in SCRIPTS:
        os.symlink(os.path.join(ROOT, name), os.path.join(tmp, name))
    mpath = os.path.join(tmp, 'session.json')
    with open(mpath, 'w') as f:
        json.dump(meta, f)
    for name, text in (('generate_exp.py', STUB_GENERATE_EXP), (os.path.join('bin', 'feacat'), STUB_FEACAT)):
        p = os.path.join(tmp, name)
        with open(p, 'w') as f:
            f.write(text % {'root': ROOT, 'meta': mpath})
        os.chmod(p, os.stat(p).st_mode | stat.S_IXUSR)
    with open(os.path.join(tmp, 'meeting.wav'), 'wb') as f:
        f.write(b'RIFF')                      # only its name and existence matter to the plumbing


def test_argument_checks_like_the_reference(tmp_path):
    orch = pkg('orchestrator')
    out = []
    orch.main([os.path.join(str(tmp_path), 'nothing.wav')], say=lambda *a: out.append(' '.join(str(x) for x in a)))
    assert out == ['%s does not exist, exiting' % os.path.join(str(tmp_path), 'nothing.wav')]
    wav = os.path.join(str(tmp_path), 'a.wav')
    open(wav, 'w').close()
    out = []
    orch.main([wav, '-fc', str(tmp_path)], say=lambda *a: out.append(' '.join(str(x) for x in a)))
    assert out[0] == 'Reading file: ' + wav and out[1] == 'Writing output to: stdout'
    assert out[2] == '%s does not exist, exiting' % os.path.join(str(tmp_path), 'feacat')


@pytest.mark.gpu
def test_spk_diarization2_chain_matches_reference_goldens(tmp_path):
    cases = {c['name']: c for c in load_cases()}
    cd, cl = cases['B_cd_gw_bic'], cases['B_cl1_hi_bic']
    tmp = str(tmp_path)
    _scratch_cwd(tmp, cd['session'])
    out = os.path.join(tmp, 'result.recipe')
    r = subprocess.run([sys.executable, './spk-diarization2.py', 'meeting.wav', '-o', out, '-fc', os.path.join(tmp, 'bin'),
                        '-tmp', os.path.join(tmp, 'tmp')], cwd=tmp, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    said = r.stdout
    for step in ('Performing exp generation and feacat concurrently', 'Calling voice-detection2.py',
                 'Waiting for feacat to end.', 'Calling spk-change-detection.py', 'Calling spk-clustering.py',
                 'Calling aku2ann.py', 'Calling aku2elan.py'):
        assert step in said, said[-2000:]
    # the intermediate recipes are the mkstemp files of the run
    inter = sorted(os.listdir(os.path.join(tmp, 'tmp')))
    vad = [n for n in inter if n.startswith('vad')][0]
    spkc = [n for n in inter if n.startswith('spkc')][0]
    assert open(os.path.join(tmp, 'tmp', vad)).read() == cd['input_recipe']
    assert open(os.path.join(tmp, 'tmp', spkc)).read() == cd['output_recipe']
    assert open(out).read() == cl['output_recipe']
    ann = open(os.path.join(tmp, 'result.ann')).read()
    assert ann.startswith('# meeting.wav\n') and ann.count('\n') == 1 + cl['output_recipe'].count('\n')
    assert '<ANNOTATION_DOCUMENT' in open(os.path.join(tmp, 'result.eaf')).read()
