"""Command lines of the three hot-path scripts, argument for argument
(spk-change-detection.py:399-466, spk-clustering.py:296-348,
spk-clustering2.py:265-317), and the configuration echo they print before and
after the work (CD:468-539,563-579; CL1:350-413,436-442; CL2:318-406).

``spk-diarization2.py`` calls ``./spk-change-detection.py`` and
``./spk-clustering.py`` by relative path with exactly these flags
(spk-diarization2.py:122-128); the executable shims of those names at the repo
root forward to ``main_change_detection`` / ``main_clustering``.
"""
import argparse
import os.path as op
import sys

from .change_detection import CDOptions, ChangeDetectionRun
from .clustering import CLOptions, ClusteringRun
from .recipe import RecipeWriter, parse_recipe, py2_str


def _say_to(stream):
    def say(*args):
        stream.write(' '.join(py2_str(a) for a in args) + '\n')
    return say


def _common(p, seg_default, out_default):
    p.add_argument('recfile', type=str, help='input recipe')
    p.add_argument('feapath', type=str, help='directory holding the .fea files')
    p.add_argument('-seg', dest='segpath', type=str, default=seg_default,
                   help='also write a recipe carrying alignment=<segpath><lna>.seg')
    p.add_argument('-o', dest='outfile', type=str, default=out_default,
                   help='output recipe (default stdout); with -seg a second file '
                        '<name>-seg<ext> is written')
    p.add_argument('-fe', dest='feaext', type=str, default='.fea', help='feature file extension')
    p.add_argument('-se', dest='segext', type=str, default='.seg', help='alignment file extension')
    p.add_argument('-f', dest='frame_rate', type=int, default=125, help='frames per second')


def build_cd_parser():
    p = argparse.ArgumentParser(description='Speaker-turn segmentation with a '
                                'BIC / GLR / KL2 distance (HIP, MI355X).')
    _common(p, None, 'stdout')
    p.add_argument('-m', dest='method', type=str, choices=['sw', 'gw', 'm'], default='sw',
                   help='sliding window, growing window, or merge of neighbouring turns')
    p.add_argument('-d', dest='distance', type=str, choices=['GLR', 'BIC', 'KL2'], default='GLR')
    p.add_argument('-w', dest='winsize', type=float, default=5.0,
                   help='window (sw) / minimum window (gw) in seconds')
    p.add_argument('-st', dest='winstep', type=float, default=0.5,
                   help='window step (sw) / largest growth step (gw) in seconds')
    p.add_argument('-dws', dest='deltaws', type=float, default=0.05,
                   help='smallest growth step of the growing window, seconds')
    p.add_argument('-t', dest='threshold', type=float, default=0.0, help='decision threshold')
    p.add_argument('-l', dest='lambdac', type=float, default=1.3, help='BIC penalty weight')
    p.add_argument('-tt', action='store_true', help='print every evaluated distance')
    p.add_argument('-dlr', action='store_true', help='keep the lna names of the input')
    return p


def build_cl_parser(variant):
    p = argparse.ArgumentParser(description='Speaker clustering with a BIC / GLR / KL2 '
                                'distance (HIP, MI355X).')
    if variant == 1:
        _common(p, '', sys.stdout)
    else:
        _common(p, None, 'stdout')
    p.add_argument('-m', dest='method', type=str, choices=['in', 'hi'], default='hi',
                   help='in-order or hierarchical agglomerative clustering')
    p.add_argument('-d', dest='distance', type=str, choices=['GLR', 'BIC', 'KL2'], default='BIC')
    p.add_argument('-t', dest='threshold', type=float, default=0.0, help='decision threshold')
    p.add_argument('-ms', dest='max_spk', type=int, default=0,
                   help='keep merging while more than this many speakers remain (0 = off)')
    p.add_argument('-l', dest='lambdac', type=float, default=1.3, help='BIC penalty weight')
    p.add_argument('-tt', action='store_true', help='print every evaluated distance')
    p.add_argument('-dlr', action='store_true', help='keep the lna names of the input')
    return p


def _default_engine():
    from .engine import HipEngine
    return HipEngine()


def _run_with_outputs(outfile, segfile, body):
    if outfile is sys.stdout:
        body(sys.stdout, None)
        return
    with open(outfile, 'w') as outf:
        if segfile:
            with open(segfile, 'w') as segf:
                body(outf, segf)
        else:
            body(outf, None)


def main_change_detection(argv=None, engine=None, stdout=None):
    out = stdout or sys.stdout
    say = _say_to(out)
    args = build_cd_parser().parse_args(argv)
    say('Reading recipe from:', args.recfile)
    with open(args.recfile, 'r') as f:
        recipe = parse_recipe(f, echo=lambda s: say(s))
    say('Reading feature files from:', args.feapath)
    if args.segpath:
        say('Setting alignment segmentation files path to:', args.segpath)
        say('Segmentation files extension:', args.segext)
    say('Feature files extension:', args.feaext)
    segfile = False
    if args.outfile != 'stdout':
        outfile = args.outfile
        say('Writing output to:', args.outfile)
        if args.segpath:
            segfile = op.splitext(op.basename(outfile))[0] + '-seg' + op.splitext(outfile)[1]
            segfile = op.join(args.segpath, segfile)
            say('Writing seg output to:', segfile)
    else:
        outfile = sys.stdout if stdout is None else stdout
        say('Writing output to: stdout')
    opts = CDOptions(rate=args.frame_rate, method=args.method, distance=args.distance,
                     winsize_s=args.winsize, winstep_s=args.winstep, deltaws_s=args.deltaws,
                     threshold=args.threshold, lambdac=args.lambdac, tt=args.tt, dlr=args.dlr)
    say('Conversion rate set to frame rate:', opts.rate)
    if args.method == 'sw':
        say('Using a fixed-size sliding window')
    elif args.method == 'gw':
        say('Using a growing window')
        say('Deltaws set to:', opts.deltaws / opts.rate, 'seconds')
    else:
        say('Performing similar-segment merge')
    if args.distance == 'GLR':
        say('Using GLR as distance measure')
    elif args.distance == 'BIC':
        say('Using BIC as distance measure, lambda =', args.lambdac)
    else:
        say('Using KL2 as distance measure')
    if args.method != 'm':
        say('Window size set to:', opts.winsize / opts.rate, 'seconds')
        say('Window step set to:', opts.winstep / opts.rate, 'seconds')
    say('Threshold distance:', args.threshold)
    if args.dlr:
        say('Disabling LNA renaming')

    if engine is None:
        engine = _default_engine()
    run = ChangeDetectionRun(engine, opts, args.feapath, args.feaext, say=say)

    def body(outf, segf):
        writer = RecipeWriter(outf, opts.rate, segf=segf, segpath=args.segpath,
                              rename_lna=not args.dlr)
        run.detect_changes(recipe, writer)

    if outfile is sys.stdout or outfile is stdout:
        body(outfile, None)
    else:
        _run_with_outputs(outfile, segfile, body)
    run.print_summary(len(recipe))
    return 0


def main_clustering(argv=None, variant=1, engine=None, stdout=None):
    out = stdout or sys.stdout
    say = _say_to(out)
    args = build_cl_parser(variant).parse_args(argv)
    say('Reading recipe from:', args.recfile)
    with open(args.recfile, 'r') as f:
        recipe = parse_recipe(f, echo=lambda s: say(s))
    say('Reading feature files from:', args.feapath)
    feapath = args.feapath
    segpath = args.segpath
    if variant == 1:
        if feapath[-1] != '/':
            feapath += '/'
        if segpath != '':
            say('Setting alignment segmentation files path to:', segpath)
            if segpath[-1] != '/':
                segpath += '/'
            say('Segmentation files extension:', args.segext)
    elif segpath:
        say('Setting alignment segmentation files path to:', segpath)
        say('Segmentation files extension:', args.segext)
    say('Feature files extension:', args.feaext)
    segfile = False
    to_stdout = (args.outfile is sys.stdout) if variant == 1 else (args.outfile == 'stdout')
    if not to_stdout:
        outfile = args.outfile
        say('Writing output to:', args.outfile)
        if segpath:
            if variant == 1:
                segfile = op.splitext(outfile)[0] + '-seg' + op.splitext(outfile)[1]
            else:
                segfile = op.splitext(op.basename(outfile))[0] + '-seg' + op.splitext(outfile)[1]
                segfile = op.join(segpath, segfile)
            say('Writing seg output to:', segfile)
    else:
        outfile = sys.stdout if stdout is None else stdout
        say('Writing output to: stdout')
    opts = CLOptions(variant=variant, rate=args.frame_rate, method=args.method,
                     distance=args.distance, threshold=args.threshold,
                     max_spk=args.max_spk, lambdac=args.lambdac, tt=args.tt, dlr=args.dlr)
    say('Conversion rate set to frame rate:', opts.rate)
    if args.method == 'hi':
        say('Using hierarchical clustering')
    else:
        say('Using in-order consecutive clustering')
    if args.distance == 'GLR':
        say('Using GLR as distance measure')
    elif args.distance == 'BIC':
        say('Using BIC as distance measure, lambda =', args.lambdac)
    else:
        say('Using KL2 as distance measure')
    say('Threshold distance:', args.threshold)
    say('Maximum speakers:', args.max_spk)
    if args.dlr:
        say('Disabling LNA renaming')

    if engine is None:
        engine = _default_engine()
    run = ClusteringRun(engine, opts, feapath, args.feaext, say=say)

    def body(outf, segf):
        writer = RecipeWriter(outf, opts.rate, segf=segf, segpath=segpath or None,
                              rename_lna=not args.dlr)
        run.process_recipe(recipe, writer)

    if to_stdout:
        body(outfile, None)
    else:
        _run_with_outputs(outfile, segfile, body)
    run.print_summary(len(recipe))
    return 0
