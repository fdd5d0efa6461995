"""Python-3 counterpart of the reference's orchestrator, spk-diarization2.py: one media
file in, one speaker recipe (plus .ann / .eaf exports) out.  It is plumbing only -- the
contract of the reference script restated (paths relative to the reference tree):

  * argv surface, path checks and the progress lines ........ spk-diarization2.py:12-74
  * non-wav media goes through ffmpeg first ................ spk-diarization2.py:76-86
  * the initial recipe `audio=<wav>` in a mkstemp file ..... spk-diarization2.py:88-92
  * generate_exp.py and feacat run side by side ............ spk-diarization2.py:94-103
  * voice-detection2.py -ms 0.5 -mns 1.5 ................... spk-diarization2.py:109-112
  * spk-change-detection.py gw / BIC with the DIA2 flags ... spk-diarization2.py:119-124
  * spk-clustering.py hi, lambda 1.3 ....................... spk-diarization2.py:126-128
  * aku2ann.py / aku2elan.py when the output is a file ..... spk-diarization2.py:130-138

Every stage is addressed as ./<script> relative to the working directory and its return
code is ignored, exactly like the reference: the stages ARE the interface.  The two the
reference takes from AaltoASR (./generate_exp.py's decoder, feacat) stay external
programs of those names; the other five are this repository's drop-in executables.
"""
import argparse
import os.path as op
import sys
from mimetypes import guess_type
from os import getcwd
from subprocess import Popen, call
from tempfile import gettempdir, mkstemp

# ---- the contract as data -------------------------------------------------------------
# command-line options: (flag, dest, default (a callable of the working directory), help)
OPTIONS = [
    ('-o', 'outfile', lambda cwd: 'stdout', 'Specifies an output recipe file, default stdout.'),
    ('-fc', 'fcpath', lambda cwd: cwd, 'Specifies the path to feacat, defaults to ./'),
    ('-fcfg', 'fcfg', lambda cwd: cwd + '/fconfig.cfg',
     'Specifies the feacat acoustic model config, defaults ./fconfig.cfg'),
    ('-lna', 'lnapath', lambda cwd: cwd + '/lna', 'Specifies the path to the lna files, defaults to ./lna'),
    ('-exp', 'exppath', lambda cwd: cwd + '/exp', 'Specifies the path to the exp files, defaults to ./exp'),
    ('-fp', 'feapath', lambda cwd: cwd + '/fea', 'Specifies the path to the feature files, defaults to ./fea'),
    ('-tmp', 'tmppath', lambda cwd: '', 'Specifies where to write the temporal files, defaults to system temporary folder.'),
]

# directory checks in the order the reference makes them: (directory tested, directory named in
# the refusal, progress line, directory named in the progress line).  The third check tests the
# exp directory again where it means the feature directory (spk-diarization2.py:70-72): kept.
DIR_CHECKS = [
    ('lnapath', 'lnapath', 'Writing lna files in:', 'lnapath'),
    ('exppath', 'exppath', 'Writing exp files in:', 'exppath'),
    ('exppath', 'feapath', 'Writing features in:', 'feapath'),
]

# the serial tail of the pipeline: (progress line, argv template); {name} fields come from the
# run's paths.  Every stage is ./<script> relative to the working directory.
SERIAL_STAGES = [
    ('Calling spk-change-detection.py',
     ['./spk-change-detection.py', '{vad}', '{fea}', '-o', '{spkc}', '-m', 'gw', '-d', 'BIC', '-w', '1.0',
      '-st', '3.0', '-dws', '0.1', '-l', '1.0']),
    ('Calling spk-clustering.py', ['./spk-clustering.py', '{spkc}', '{fea}', '-o', '{out}', '-m', 'hi', '-l', '1.3']),
]
EXPORT_STAGES = [                                  # only when the output is a file
    ('Calling aku2ann.py', ['./aku2ann.py', '{out}', '-o', '{stem}.ann']),
    ('Calling aku2elan.py', ['./aku2elan.py', '{out}', '-o', '{stem}.eaf']),
]
EXP_STAGE = ['./generate_exp.py', '{init}', '-e', '{exp}', '-l', '{lna}']
FEACAT_STAGE = ['{feacat}', '-c', '{fcfg}', '-H', '--raw-output', '{wav}']
VAD_STAGE = ['./voice-detection2.py', '{init}', '{exp}', '-o', '{vad}', '-ms', '0.5', '-mns', '1.5']
FFMPEG_STAGE = ['ffmpeg', '-i', '{media}', '-ar', '16000', '-ac', '1', '-ab', '32k', '{wav}']


def _fill(template, paths):
    return [t.format(**paths) for t in template]


def _scratch_recipe(prefix, where):
    return mkstemp(suffix='.recipe', prefix=prefix, dir=where)[1]


def main(argv=None, say=None):
    if say is None:
        def say(*a):
            print(*a)
            sys.stdout.flush()
    cwd = getcwd()
    parser = argparse.ArgumentParser(description='Process a media file to perform segmentation and '
                                                 'speaker clustering on it.')
    parser.add_argument('infile', type=str, help='Specifies the media file')
    for flag, dest, default, text in OPTIONS:
        parser.add_argument(flag, dest=dest, type=str, default=default(cwd), help=text)
    o = parser.parse_args(argv)

    # ---- what must exist, said in the reference's words and order; a miss ends the run quietly
    if not op.isfile(o.infile):
        say('%s does not exist, exiting' % o.infile)
        return 0
    say('Reading file:', o.infile)
    say('Writing output to:', o.outfile)
    feacat = op.join(o.fcpath, 'feacat')
    if not op.isfile(feacat):
        say('%s does not exist, exiting' % feacat)
        return 0
    say('Using feacat from:', feacat)
    tmp = o.tmppath if op.isdir(o.tmppath) else gettempdir()
    say('Writing temporal files in:', tmp)
    for tested, named, line, shown in DIR_CHECKS:
        if not op.isdir(getattr(o, tested)):
            say('Path %s does not exist, exiting' % getattr(o, named))
            return 0
        say(line, getattr(o, shown))

    paths = dict(media=o.infile, wav=o.infile, feacat=feacat, fcfg=o.fcfg, exp=o.exppath, lna=o.lnapath,
                 fea=o.feapath, out=o.outfile)
    if guess_type(o.infile)[0] != 'audio/x-wav':
        say('Media is not a .wav audio file, attempting to extract a .wav file')
        say('Calling ffmpeg')
        paths['wav'] = op.splitext(o.infile)[0] + '.wav'
        call(_fill(FFMPEG_STAGE, paths))
    paths['init'] = _scratch_recipe('init', tmp)
    with open(paths['init'], 'w') as f:
        f.write('audio=' + paths['wav'] + '\n')

    # ---- the two producers side by side; the VAD recipe as soon as the .exp files are there
    say('Performing exp generation and feacat concurrently')
    decoder = Popen(_fill(EXP_STAGE, paths))
    fea_file = op.join(o.feapath, op.splitext(op.basename(paths['wav']))[0] + '.fea')
    with open(fea_file, 'w') as sink:
        extractor = Popen(_fill(FEACAT_STAGE, paths), stdout=sink)
        decoder.wait()
        say('Calling voice-detection2.py')
        paths['vad'] = _scratch_recipe('vad', tmp)
        call(_fill(VAD_STAGE, paths))
        say('Waiting for feacat to end.')
        extractor.wait()

    # ---- the hot path, then the exports; return codes are not looked at (like the reference)
    paths['spkc'] = _scratch_recipe('spkc', tmp)
    stages = list(SERIAL_STAGES)
    if o.outfile != 'stdout':
        paths['stem'] = op.join(op.dirname(o.outfile), op.splitext(op.basename(o.outfile))[0])
        stages += EXPORT_STAGES
    for line, template in stages:
        say(line)
        call(_fill(template, paths))
    return 0
