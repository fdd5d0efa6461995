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


def main(argv=None, say=None):
    if say is None:
        def say(*a):
            print(*a)
            sys.stdout.flush()
    parser = argparse.ArgumentParser(description='Process a media file to perform segmentation and '
                                                 'speaker clustering on it.')
    parser.add_argument('infile', type=str, help='Specifies the media file')
    parser.add_argument('-o', dest='outfile', type=str, default='stdout',
                        help='Specifies an output recipe file, default stdout.')
    parser.add_argument('-fc', dest='fcpath', type=str, default=getcwd(),
                        help='Specifies the path to feacat, defaults to ./')
    parser.add_argument('-fcfg', dest='fcfg', type=str, default=getcwd() + '/fconfig.cfg',
                        help='Specifies the feacat acoustic model config, defaults ./fconfig.cfg')
    parser.add_argument('-lna', dest='lnapath', type=str, default=getcwd() + '/lna',
                        help='Specifies the path to the lna files, defaults to ./lna')
    parser.add_argument('-exp', dest='exppath', type=str, default=getcwd() + '/exp',
                        help='Specifies the path to the exp files, defaults to ./exp')
    parser.add_argument('-fp', dest='feapath', type=str, default=getcwd() + '/fea',
                        help='Specifies the path to the feature files, defaults to ./fea')
    parser.add_argument('-tmp', dest='tmppath', type=str, default='',
                        help='Specifies where to write the temporal files, defaults to system temporary folder.')
    args = parser.parse_args(argv)

    if not op.isfile(args.infile):
        say('%s does not exist, exiting' % args.infile)
        return 0
    say('Reading file:', args.infile)
    outfile = args.outfile
    say('Writing output to:', outfile)
    args.fcpath = op.join(args.fcpath, 'feacat')
    if not op.isfile(args.fcpath):
        say('%s does not exist, exiting' % args.fcpath)
        return 0
    say('Using feacat from:', args.fcpath)
    if not op.isdir(args.tmppath):
        args.tmppath = gettempdir()
    say('Writing temporal files in:', args.tmppath)
    if not op.isdir(args.lnapath):
        say('Path %s does not exist, exiting' % args.lnapath)
        return 0
    say('Writing lna files in:', args.lnapath)
    if not op.isdir(args.exppath):
        say('Path %s does not exist, exiting' % args.exppath)
        return 0
    say('Writing exp files in:', args.exppath)
    # (the reference tests exppath a second time here where it means feapath, :70-72)
    if not op.isdir(args.exppath):
        say('Path %s does not exist, exiting' % args.feapath)
        return 0
    say('Writing features in:', args.feapath)

    if guess_type(args.infile)[0] != 'audio/x-wav':
        say('Media is not a .wav audio file, attempting to extract a .wav file')
        say('Calling ffmpeg')
        infile = op.splitext(args.infile)[0] + '.wav'
        call(['ffmpeg', '-i', args.infile, '-ar', '16000', '-ac', '1', '-ab', '32k', infile])
    else:
        infile = args.infile

    init_recipe = mkstemp(suffix='.recipe', prefix='init', dir=args.tmppath)[1]
    with open(init_recipe, 'w') as f:
        f.write('audio=' + infile + '\n')

    say('Performing exp generation and feacat concurrently')
    child1 = Popen(['./generate_exp.py', init_recipe, '-e', args.exppath, '-l', args.lnapath])
    with open(op.join(args.feapath, op.splitext(op.basename(infile))[0] + '.fea'), 'w') as feafile:
        child2 = Popen([args.fcpath, '-c', args.fcfg, '-H', '--raw-output', infile], stdout=feafile)
        child1.wait()                                  # the exp files are needed first
        say('Calling voice-detection2.py')
        vad_recipe = mkstemp(suffix='.recipe', prefix='vad', dir=args.tmppath)[1]
        call(['./voice-detection2.py', init_recipe, args.exppath, '-o', vad_recipe, '-ms', '0.5', '-mns', '1.5'])
        say('Waiting for feacat to end.')
        child2.wait()

    spkchange_recipe = mkstemp(suffix='.recipe', prefix='spkc', dir=args.tmppath)[1]
    say('Calling spk-change-detection.py')
    call(['./spk-change-detection.py', vad_recipe, args.feapath, '-o', spkchange_recipe, '-m', 'gw', '-d', 'BIC',
          '-w', '1.0', '-st', '3.0', '-dws', '0.1', '-l', '1.0'])
    say('Calling spk-clustering.py')
    call(['./spk-clustering.py', spkchange_recipe, args.feapath, '-o', outfile, '-m', 'hi', '-l', '1.3'])

    if outfile != 'stdout':
        outf = op.splitext(op.basename(outfile))[0]
        outfpath = op.dirname(outfile)
        say('Calling aku2ann.py')
        call(['./aku2ann.py', outfile, '-o', op.join(outfpath, outf + '.ann')])
        say('Calling aku2elan.py')
        call(['./aku2elan.py', outfile, '-o', op.join(outfpath, outf + '.eaf')])
    return 0
