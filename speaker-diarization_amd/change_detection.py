"""Host-side driver for speaker-turn segmentation: the py3 mirror of the
reference's ``detect_changes`` / ``dist_gw`` / ``dist_sw`` / ``merge_rec`` operator
interface (spk-change-detection.py:136-395), with the numerics delegated to an
engine object (``engine.HipEngine`` in the product; tests may plug the CPU oracle).

Split of labour
  * growing window (``gw``): the whole decision chain of CD:180-288 runs on the
    device, one workgroup per VAD turn; the host replays the returned window /
    detection lists only to write lines and to accumulate the summary counters
    in the reference's order.
  * sliding window (``sw``): every window distance is independent -> one device
    launch per file returns d[w]; the positive-run post-pass of CD:316-354 is a
    cheap sequential scan done here.
  * merge (``m``): one distance per recipe line, decision dependent -> host loop
    over ``engine.pair_terms``; reproduces the frozen ``c1`` of the 5-argument
    ``bic`` called without ``i``/``saved`` (CD:72,84-90,149).
    A decision that does NOT merge leaves the next pair of lines as they stand in
    the recipe, so the terms of every adjacent pair of a file are taken in ONE
    batched call up front and only the decisions behind a merge (whose left side
    is a range no recipe line names) go back to the device one by one.
"""
import math
import sys

from . import feaio
from .recipe import py2_str

MAXINT = 9223372036854775807     # sys.maxint on the reference's 64-bit py2
P_DIM = 39


def _isinf(x):
    return x == math.inf or x == -math.inf


class CDOptions(object):
    """Parsed command line in frame units (CD:499-532)."""

    def __init__(self, rate=125, method='sw', distance='GLR', winsize_s=5.0,
                 winstep_s=0.5, deltaws_s=0.05, threshold=0.0, lambdac=1.3,
                 tt=False, dlr=False):
        self.rate = float(rate)
        self.method = method
        self.distance = distance
        self.winsize = float(math.floor(winsize_s * self.rate))
        self.winstep = float(math.floor(winstep_s * self.rate))
        self.deltaws = float(math.floor(self.rate * deltaws_s))
        self.threshold = threshold
        self.lambdac = lambdac
        self.tt = tt
        self.dlr = dlr


class ChangeDetectionRun(object):
    def __init__(self, engine, opts, feapath, feaext='.fea', say=None):
        self.eng = engine
        self.o = opts
        self.feapath = feapath
        self.feaext = feaext
        self.say = say or (lambda *a: sys.stdout.write(' '.join(py2_str(x) for x in a) + '\n'))
        # summary counters, CD:542-549
        self.total_dist = 0
        self.max_dist = 0
        self.min_dist = MAXINT
        self.total_windows = 0
        self.total_det_dist = 0
        self.max_det_dist = 0
        self.min_det_dist = MAXINT
        self.total_segments = 0
        self._frozen_c1 = None          # A-8
        self._prev = None               # merge_rec.prev
        self._ahead = {}                # merge mode: terms of the adjacent pairs of the current file

    # ----------------------------------------------------------------- counters
    def _window_stat(self, d):
        if not _isinf(d):
            self.total_dist += d
            self.total_windows += 1
            if d > self.max_dist:
                self.max_dist = d
            if d < self.min_dist:
                self.min_dist = d

    def _detection_stat(self, d):
        self.total_det_dist += d
        self.total_segments += 1
        if d > self.max_det_dist:
            self.max_det_dist = d
        if d < self.min_det_dist:
            self.min_det_dist = d

    # ----------------------------------------------------------------- features
    def _load(self, recline):
        dim, feats = feaio.load_features(
            feaio.fea_path(recline[0], self.feapath, self.feaext, join=True))
        self.eng.set_features(feats)
        return feats.shape[0]

    # ----------------------------------------------------------------- driver
    def detect_changes(self, recipe, writer):
        """CD:360-395.  Lines are grouped per feature file so one device launch
        covers every VAD turn of that file; output order is unchanged."""
        if self.o.method == 'm':
            return self._merge_pass(recipe, writer)
        this_wav, this_lna = '', ''
        group = []              # [(recline, f0, f1)] for the file currently loaded
        nframes = 0
        rate = self.o.rate
        for rl in recipe:
            if rl[0] != this_wav:
                self._flush(group, writer)
                group = []
                this_wav = rl[0]
                nframes = self._load(rl)
            if rl[1] != this_lna:       # A-11: consecutive equal lna are skipped
                this_lna = rl[1]
                f0 = min(int(rl[2] * rate), nframes)
                f1 = min(int(rl[3] * rate), nframes)
                group.append((rl, f0, max(f0, f1)))
        self._flush(group, writer)

    def _flush(self, group, writer):
        if not group:
            return
        if self.o.method == 'gw':
            self._gw_group(group, writer)
        else:
            self._sw_group(group, writer)

    # ----------------------------------------------------------------- gw
    def _gw_group(self, group, writer):
        o = self.o
        turns = [(f0, f1) for (_, f0, f1) in group]
        results = self.eng.gw(turns, kind=o.distance, lambdac=o.lambdac,
                              threshold=o.threshold, winsize=o.winsize,
                              winstep=o.winstep, deltaws=o.deltaws, rate=o.rate,
                              trace=o.tt)
        for (rl, f0, f1), res in zip(group, results):
            lna_start, lna_end = rl[2], rl[3]
            for ev in res.events:
                # ('cand', start, i, n1, n2, d, coarse) | ('win', maxd) | ('det', start, maxi, d)
                if ev[0] == 'cand':
                    _, start, i, n1, n2, d, coarse = ev
                    if o.tt and coarse:
                        self.say('Time:', start / o.rate + i / o.rate + lna_start,
                                 '- Distance:', d)
                    if _isinf(d):
                        self.say('Inf:', (n1, P_DIM), (n2, P_DIM), d)
                elif ev[0] == 'win':
                    maxd = ev[1]
                    if maxd is None:                  # no candidate was accepted
                        maxd = -MAXINT - 1
                    self._window_stat(maxd)
                else:
                    _, start, maxi, dfinal = ev
                    writer.write(rl, start, start + maxi, lna_start, 'spk_turn')
                    self._detection_stat(dfinal)
            writer.write(rl, res.final_start, (lna_end - lna_start) * o.rate,
                         lna_start, 'spk_turn')

    # ----------------------------------------------------------------- sw
    def _sw_group(self, group, writer):
        o = self.o
        turns = [(f0, f1) for (_, f0, f1) in group]
        if o.distance == 'BIC':
            # The reference passes features[start:end] with end == 0 as the pooled
            # array, i.e. an empty slice: NaN covariance -> scipy's finite check
            # raises on the first window of the run (CD:308-310).
            for (rl, f0, f1) in group:
                if 2 * o.winsize <= (f1 - f0):
                    raise ValueError('array must not contain infs or NaNs')
                lna_start, lna_end = rl[2], rl[3]
                writer.write(rl, 0, (lna_end - lna_start) * o.rate, lna_start, 'spk_turn')
            return
        dists = self.eng.sw(turns, kind=o.distance, lambdac=o.lambdac,
                            winsize=o.winsize, winstep=o.winstep)
        for (rl, f0, f1), d_arr in zip(group, dists):
            self._sw_postpass(rl, f1 - f0, d_arr, writer)

    def _sw_postpass(self, rl, nframes, d_arr, writer):
        """CD:299-357 with the distance calls replaced by the precomputed d[w]."""
        o = self.o
        lna_start, lna_end = rl[2], rl[3]
        winsize, winstep, thr = o.winsize, o.winstep, o.threshold
        start = 0
        end = 0
        bestd = -1
        best_position = -1
        last_positive = -1
        w = 0
        while start + 2 * winsize <= nframes:
            d = float(d_arr[w]); w += 1
            if o.tt:
                self.say('Time:', (start + winsize) / o.rate + lna_start, '- Distance:', d)
            self._window_stat(d)
            if d < thr or _isinf(d):
                if start - winstep == last_positive:
                    writer.write(rl, end, best_position, lna_start, 'spk_turn')
                    self._detection_stat(bestd)
                    bestd = 0
                    end = best_position
            else:
                if d > bestd:
                    bestd = d
                    best_position = start + winsize
                last_positive = start
            start += winstep
        if start - winstep == last_positive:
            writer.write(rl, end, best_position, lna_start, 'spk_turn')
            self._detection_stat(bestd)
            bestd = 0
            end = best_position
        writer.write(rl, end, (lna_end - lna_start) * o.rate, lna_start, 'spk_turn')

    # ----------------------------------------------------------------- merge
    def _merge_ranges(self, nframes, prev, nxt):
        rate = self.o.rate
        clamp = lambda t: max(0, min(int(t), nframes))
        a = (clamp(prev[2] * rate), clamp(prev[3] * rate))
        b = (clamp(nxt[2] * rate), clamp(nxt[3] * rate))
        return (a[0], max(a)), (b[0], max(b))

    def _merge_speculate(self, recipe, l, nframes):
        """Terms of every adjacent pair of lines of the file that starts at recipe[l], as the
        recipe names them, in one engine call -> {(a, b): terms}.  A pair with a non-finite
        covariance makes the reference raise AT that pair, after the lines before it were
        written: then nothing is taken ahead and every decision makes its own call."""
        o = self.o
        jobs, keys, seen = [], [], set()
        wav = recipe[l][0]
        while l + 1 < len(recipe) and recipe[l + 1][0] == wav:
            a, b = self._merge_ranges(nframes, recipe[l], recipe[l + 1])
            if (a, b) not in seen:
                seen.add((a, b))
                keys.append((a, b))
                jobs.append(([a], [b]))
            l += 1
        if len(jobs) < 2:
            return {}
        try:
            terms = self.eng.pair_terms(jobs, want_glr=(o.distance == 'GLR'), want_kl2=(o.distance == 'KL2'))
        except ValueError:
            return {}
        return dict(zip(keys, terms))

    def _merge_distance(self, nframes, prev, nxt):
        o = self.o
        a, b = self._merge_ranges(nframes, prev, nxt)
        t = self._ahead.get((a, b))
        if t is None:
            t = self.eng.pair_terms([([a], [b])], want_glr=(o.distance == 'GLR'),
                                    want_kl2=(o.distance == 'KL2'))[0]
        if o.distance == 'BIC':
            if self._frozen_c1 is None:
                self._frozen_c1 = 0.5 * t.n1 * t.logdet1
            n = t.n1 + t.n2
            d = 0.5 * n * t.logdet_union - self._frozen_c1 - 0.5 * t.n2 * t.logdet2
            d -= o.lambdac * 0.5 * (P_DIM + 0.5 * P_DIM * (P_DIM + 1)) * math.log(n)
            return d
        if o.distance == 'GLR':
            return glr_from_terms(t)
        return t.kl2

    def _merge_step(self, nframes, nxt, writer):
        """CD:136-177."""
        o = self.o
        prev = self._prev
        d = self._merge_distance(nframes, prev, nxt)
        if o.tt:
            self.say('Time:', prev[3] * o.rate, '- Distance:', d)
        self._window_stat(d)
        if d < o.threshold and not _isinf(d):
            self._prev = (prev[0], prev[1], prev[2], nxt[3])
            self._detection_stat(d)
        else:
            writer.write(prev, prev[2] * o.rate, prev[3] * o.rate, 0, 'spk_turn')
            self._prev = (nxt[0], nxt[1], nxt[2], nxt[3])

    def _merge_pass(self, recipe, writer):
        """Merge-mode branch of CD:366-395."""
        o = self.o
        this_wav = ''
        nframes = 0
        wav_start = True
        l = 0
        while l < len(recipe):
            if recipe[l][0] != this_wav:
                this_wav = recipe[l][0]
                nframes = self._load(recipe[l])
                self._ahead = self._merge_speculate(recipe, l, nframes)
            if l + 1 < len(recipe):
                if recipe[l + 1][0] != this_wav:
                    l += 1
                    wav_start = True
                    continue
                if wav_start:
                    wav_start = False
                    self._prev = tuple(recipe[l])
                self._merge_step(nframes, recipe[l + 1], writer)
            else:
                prev = self._prev
                if prev is None:
                    # single-line recipe: the reference dies on the unset
                    # function attribute (CD:392)
                    raise AttributeError("'function' object has no attribute 'prev'")
                writer.write(prev, prev[2] * o.rate, prev[3] * o.rate, 0, 'spk_turn')
            l += 1

    # ----------------------------------------------------------------- summary
    def print_summary(self, n_recipe_lines):
        """CD:563-579."""
        say = self.say
        say('Useful metrics for determining the right threshold:')
        say('---------------------------------------------------')
        if self.total_windows > 0:
            say('Average between windows distance:', float(self.total_dist) / self.total_windows)
        say('Maximum between windows distance:', self.max_dist)
        if self.min_dist < MAXINT:
            say('Minimum between windows distance:', self.min_dist)
        say('Total windows:', self.total_windows)
        say('Total segments:', self.total_segments + n_recipe_lines)
        if self.total_segments > 0:
            say('Average between detected segments distance:',
                float(self.total_det_dist) / self.total_segments)
        say('Maximum between detected segments distance:', self.max_det_dist)
        if self.min_det_dist < MAXINT:
            say('Minimum between detected segments distance:', self.min_det_dist)
        say('Total detected speaker changes:', self.total_segments)


def _log(x):
    """numpy.log semantics on a Python float: log(0) = -inf, log(<0) = nan."""
    if x != x:
        return x
    if x > 0:
        return math.log(x) if x != math.inf else math.inf
    return -math.inf if x == 0 else math.nan


def bic_from_terms(t, lambdac):
    """CL1:88-100 on (n, log det) terms, same operation order."""
    n = t.n1 + t.n2
    d = 0.5 * n * t.logdet_union - 0.5 * t.n1 * t.logdet1 - 0.5 * t.n2 * t.logdet2
    d -= lambdac * 0.5 * (P_DIM + 0.5 * P_DIM * (P_DIM + 1)) * math.log(n)
    return d


def glr_from_terms(t):
    """CD:103-121 / CL1:103-121 on (n, log det) terms, same operation order."""
    n = float(t.n1 + t.n2)
    return -(n / 2.0) * ((t.n1 / n) * t.logdet1 + (t.n2 / n) * t.logdet2 - t.logdet_glr)
