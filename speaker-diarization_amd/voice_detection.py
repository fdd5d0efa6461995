"""Speech / non-speech turn recipe from VAD `.exp` files: the step immediately upstream
of change detection (SURVEY.md §8(f) "next" row 1), host-side only — a token-stream state
machine, no numerics.  Mirrors the command line of the reference's voice-detection2.py;
its output is the `vad.recipe` the hot path starts from and defines the `lna=a_1 ...`
names the later stages rewrite.

Reference behaviour restated here (paths relative to the reference tree):
  * recipe reader (only `audio=` is looked at) ......... voice-detection2.py:8-19
  * wav name -> `.exp` path, exit when it is missing ... voice-detection2.py:22-30
  * lna base names a, b, ..., z, aa, ab, ... ........... voice-detection2.py:33-41
  * the turn state machine over `<frame> p|<w>` tokens  voice-detection2.py:44-117
  * an unfinished last turn ends at `.last_frame` ..... voice-detection2.py:108-116
  * writer, py2 str(float) = 12 significant digits ..... voice-detection2.py:120-125
  * argv surface and the configuration echo ........... voice-detection2.py:136-198
"""
import argparse
import os.path as op
import re
import sys

from .recipe import py2_str

_AUDIO = re.compile(r'audio=(\S+)')
_TOKEN = re.compile(r'(\d+) (p|<w>)')


class MissingExp(SystemExit):
    """The reference calls exit() after printing the error line."""


def parse_recipe(lines, say):
    """Audio file of every recipe line that names one (others are echoed and skipped)."""
    out = []
    for line in lines:
        m = _AUDIO.search(line)
        if m is None:
            say('Recipe line without recognizable audio files:')
            say(line)
            continue
        out.append(m.group(1))
    return out


def wav_to_exp(wavfile, exppath, say):
    exp = op.join(exppath, op.splitext(op.basename(wavfile))[0] + '.exp')
    if not op.isfile(exp):
        say('Error,', exp, 'does not exist')
        raise MissingExp()
    return exp


def inc_lna(lna):
    """Next base name: the last letter that is not 'z' is advanced and everything
    behind it restarts at 'a'; all-'z' names grow by one letter."""
    for c in range(len(lna) - 1, -1, -1):
        if lna[c] != 'z':
            return lna[:c] + chr(ord(lna[c]) + 1) + 'a' * (len(lna) - c - 1)
    return 'a' * (len(lna) + 1)


class VadOptions(object):
    def __init__(self, rate=125, minspeech=0.2, minnonspeech=0.3, seg_before_exp=0.0, seg_end_exp=0.0):
        self.rate = rate
        self.minspeech = minspeech
        self.minnonspeech = minnonspeech
        self.seg_before_exp = seg_before_exp
        self.seg_end_exp = seg_end_exp


def turns_from_tokens(tokens, lna, opt, last_frame=None):
    """tokens: iterable of (frame_text, 'p' | '<w>').  Returns [(lna_n, start_s, end_s)].
    `last_frame` is a callable giving the file's last frame number; it is only consulted
    when a turn is still open at the end (the reference opens `.last_frame` only then).

    The clock advances by (frame - previous frame) / rate per token, accumulated in the
    order the reference does (so the same doubles come out).  A turn opens at a 'p' that
    is followed by at least `minspeech` of speech, and closes at a '<w>' that is followed
    by at least `minnonspeech` of silence."""
    rate = float(opt.rate)
    ms, mns, sbe, see = opt.minspeech, opt.minnonspeech, opt.seg_before_exp, opt.seg_end_exp
    turns = []
    count = 1
    start = end = total = previous = 0.0
    in_speech = False
    for frame_text, token in tokens:
        frame = float(frame_text)
        advanced = (frame - previous) / rate
        previous = frame
        total += advanced
        if in_speech:
            if token == '<w>':
                end = total                          # possible end
            elif end:                                # 'p' after a possible end
                if advanced < mns:
                    end = 0.0                        # too short a pause: the turn goes on
                else:
                    in_speech = False
                    turns.append((lna + '_' + str(count), start - sbe, end + see))
                    count += 1
                    start = total                    # possible start of the next turn
        else:
            if token == '<w>':
                if start:                            # '<w>' after a possible start
                    if advanced < ms:
                        start = 0.0                  # too short to be a turn
                    else:
                        end = total                  # this silence is a possible end already
                        in_speech = True
            else:
                start = total                        # possible start
    if start:
        last_time = float(last_frame()) / rate
        if last_time - start >= ms:
            turns.append((lna + '_' + str(count), start - sbe, last_time))
    return turns


def parse_exp_file(expfile, lna, opt):
    def tokens():
        with open(expfile, 'r') as f:
            for line in f:
                for m in _TOKEN.finditer(line):
                    yield m.group(1), m.group(2)

    def last_frame():
        with open(op.splitext(expfile)[0] + '.last_frame', 'r') as f:
            return f.read()

    return turns_from_tokens(tokens(), lna, opt, last_frame)


def recipe_line(wav, lna, start, end):
    return 'audio=' + wav + ' lna=' + lna + ' start-time=' + py2_str(start) + ' end-time=' + py2_str(end) + '\n'


def write_recipe(wavs, exppath, outf, opt, say):
    lna = 'a'
    for wav in wavs:
        for (name, start, end) in parse_exp_file(wav_to_exp(wav, exppath, say), lna, opt):
            outf.write(recipe_line(wav, name, start, end))
        lna = inc_lna(lna)


def build_parser():
    p = argparse.ArgumentParser(description='Creates a recipe from the Speech Activity Detection '
                                'generate_exp.py output, (.exp files) that is, speech/non-speech turn detection')
    p.add_argument('recfile', type=str, help='Specifies the input recipe file')
    p.add_argument('exppath', type=str, help='Specifies the input .exp files path')
    p.add_argument('-o', dest='outfile', type=str, default='stdout', help='Specifies an output file, default stdout.')
    p.add_argument('-r', dest='rate', type=int, default=125, help='Specifies the sample rate, default 125.')
    p.add_argument('-ms', dest='minspeech', type=float, default=0.2,
                   help='Specifies the minimum speech turn duration, default 0.2 seconds (roughly one word).')
    p.add_argument('-mns', dest='minnonspeech', type=float, default=0.3,
                   help='Specifies the minimum nonspeech between-turns duration, default 0.3 seconds (NIST standard).')
    p.add_argument('-sbe', dest='seg_before_exp', type=float, default=0.0,
                   help='Specifies a segment expansion time removed before each detected segment. Default 0.0.')
    p.add_argument('-see', dest='seg_end_exp', type=float, default=0.0,
                   help='Specifies a segment end expansion time added after each detected segment. Default 0.0.')
    return p


def main(argv=None, stdout=None):
    out = stdout or sys.stdout

    def say(*items):
        out.write(' '.join(py2_str(x) for x in items) + '\n')

    args = build_parser().parse_args(argv)
    say('Reading recipe from:', args.recfile)
    with open(args.recfile, 'r') as f:
        wavs = parse_recipe(f, say)
    say('Reading .exp files from:', args.exppath)
    if not op.isdir(args.exppath):
        say('Error,', args.exppath, 'is not a valid directory')
        return None
    if args.outfile != 'stdout':
        say('Writing output to:', args.outfile)
    else:
        say('Writing output to: stdout')
    say('Sample rate set to:', args.rate)
    say('Minimum speech turn duration:', args.minspeech, 'seconds')
    say('Minimum nonspeech between-turns duration:', args.minnonspeech, 'seconds')
    say('Segment before expansion set to:', args.seg_before_exp, 'seconds')
    say('Segment end expansion set to:', args.seg_end_exp, 'seconds')
    opt = VadOptions(args.rate, args.minspeech, args.minnonspeech, args.seg_before_exp, args.seg_end_exp)
    try:
        if args.outfile != 'stdout':
            with open(args.outfile, 'w') as outf:
                write_recipe(wavs, args.exppath, outf, opt, say)
        else:
            write_recipe(wavs, args.exppath, out, opt, say)
    except MissingExp:
        return None
    return None
