"""Host-side driver for speaker clustering: the py3 mirror of ``process_recipe`` /
``spk_cluster_hi`` / ``spk_cluster_in`` (spk-clustering.py:136-292 = "v1",
spk-clustering2.py:135-261 = "v2"), with the numerics delegated to an engine.

  * hierarchical (``hi``): segment statistics, the N x N distance matrix and the
    whole merge loop run on the device (``engine.cluster_hi``); the host replays
    the returned merge log to print the ``Merging:`` lines, rebuild the cluster
    lists and emit the recipe in the reference's order.
  * in-order (``in``): one decision per recipe line against the clusters found so
    far -> host loop over ``engine.pair_terms``.

v1/v2 differences kept on purpose: matrix
initialisation and update rule (device side), summary statistics, the int()
casts of ``spk_cluster_in`` and the default of ``-o``.
"""
import math
import sys

from . import feaio
from .change_detection import MAXINT, bic_from_terms, glr_from_terms, _isinf
from .recipe import py2_str


class CLOptions(object):
    def __init__(self, variant=1, rate=125, method='hi', distance='BIC',
                 threshold=0.0, max_spk=0, lambdac=1.3, tt=False, dlr=False):
        self.variant = variant
        self.rate = float(rate)
        self.method = method
        self.distance = distance
        self.threshold = threshold
        self.max_spk = max_spk
        self.lambdac = lambdac
        self.tt = tt
        self.dlr = dlr


class ClusteringRun(object):
    def __init__(self, engine, opts, feapath, feaext='.fea', say=None):
        self.eng = engine
        self.o = opts
        self.feapath = feapath
        self.feaext = feaext
        self.say = say or (lambda *a: sys.stdout.write(' '.join(py2_str(x) for x in a) + '\n'))
        self.max_dist = 0
        self.min_dist = MAXINT
        self.speakers = []
        self.nframes = 0

    def _load(self, recline):
        # v1 concatenates feapath + name (the CLI appends the missing '/'),
        # v2 uses os.path.join: CL1:31-35,357-358 vs CL2:32-37.
        dim, feats = feaio.load_features(
            feaio.fea_path(recline[0], self.feapath, self.feaext, join=(self.o.variant == 2)))
        self.eng.set_features(feats)
        self.nframes = feats.shape[0]
        self._spk_sets = None           # (the clamped frame ranges depend on the file's length)

    def _frames(self, a, b):
        """int() truncation + the clamping a numpy slice applies (CL1:46-52)."""
        n = self.nframes
        a = max(0, min(int(a), n))
        b = max(0, min(int(b), n))
        return (a, max(a, b))

    # ----------------------------------------------------------------- driver
    def process_recipe(self, recipe, writer):
        """CL1:263-292 / CL2:232-261."""
        o = self.o
        rate = o.rate
        this_wav = ''
        first = 0
        if (o.method == 'in' and hasattr(self.eng, 'cluster_in')
                and len(recipe) > 1 and all(rl[0] == recipe[0][0] for rl in recipe)):
            # one file, a distance the device chain knows: the whole decision chain in one call
            this_wav = recipe[0][0]
            first = self._cluster_in_chain(recipe, writer)
        for l, rl in enumerate(recipe):
            if l < first:
                continue
            if rl[0] != this_wav:
                this_wav = rl[0]
                self._load(rl)
            if self.speakers == [] and o.method == 'in':
                self.speakers.append([(rl[2] * rate, rl[3] * rate)])
                writer.write(rl, rl[2] * rate, rl[3] * rate, 0,
                             'speaker_' + str(len(self.speakers)))
            elif o.method == 'hi':
                self.speakers.append([(rl[2] * rate, rl[3] * rate, l)])
            else:
                self._cluster_in(rl, writer)
        if o.method == 'hi':
            self.say('Initial cluster with:', len(self.speakers), 'speakers')
            if not recipe:
                # the reference dereferences the never-assigned feature table
                raise UnboundLocalError("local variable 'feas' referenced before assignment")
            self._cluster_hi(recipe, writer)

    # ----------------------------------------------------------------- distance
    def _distance(self, t):
        o = self.o
        if o.distance == 'BIC':
            return bic_from_terms(t, o.lambdac)
        if o.distance == 'GLR':
            return glr_from_terms(t)
        return t.kl2

    # ----------------------------------------------------------------- in-order
    def _cluster_in_chain(self, recipe, writer):
        """CL1:136-175 / CL2:135-170 for a whole one-file recipe: the decisions are taken on the
        device (engine.cluster_in), the reference's prints, statistics, speaker lists and recipe
        lines are replayed from the returned distances and labels.  Returns the number of lines
        done (fewer than all: a non-finite covariance -- the per-line path takes over there and
        raises where the reference does)."""
        o = self.o
        rate = o.rate
        self._load(recipe[0])
        segs = [self._frames(rl[2] * rate, rl[3] * rate) for rl in recipe]
        labels, dists, done = self.eng.cluster_in(segs, o.distance, o.lambdac, o.threshold)
        for l in range(done):
            rl = recipe[l]
            if l == 0:
                self.speakers.append([(rl[2] * rate, rl[3] * rate)])
                writer.write(rl, rl[2] * rate, rl[3] * rate, 0, 'speaker_' + str(len(self.speakers)))
                continue
            if o.variant == 1:
                start, end = int(rl[2] * rate), int(rl[3] * rate)
            else:
                start, end = rl[2] * rate, rl[3] * rate
            for k, d in enumerate(dists[l]):
                d = float(d)
                if o.tt:
                    self.say('Time:', end, '- Distance:', d, '- Speaker:', k + 1)
                if not _isinf(d):
                    if d > self.max_dist:
                        self.max_dist = d
                    if d < self.min_dist:
                        self.min_dist = d
            best = int(labels[l])
            if best < len(self.speakers):
                self.speakers[best].append((start, end))
                writer.write(rl, start, end, 0, 'speaker_' + str(best + 1))
            else:
                self.speakers.append([(start, end)])
                writer.write(rl, start, end, 0, 'speaker_' + str(len(self.speakers)))
        self._spk_sets = None
        return done

    def _cluster_in(self, rl, writer):
        """CL1:136-175 / CL2:135-170."""
        o = self.o
        if o.variant == 1:
            start = int(rl[2] * o.rate)
            end = int(rl[3] * o.rate)
        else:
            start = rl[2] * o.rate
            end = rl[3] * o.rate
        # every cluster's frame ranges as one tuple, kept across calls and extended when a segment
        # joins: rebuilding them from the float times for every new segment is quadratic in the
        # number of segments, and the engine keys its record cache on exactly these tuples
        seg_r = self._frames(start, end)
        seg = (seg_r,)
        if self._spk_sets is None or len(self._spk_sets) != len(self.speakers):
            self._spk_sets = [tuple(self._frames(s[0], s[1]) for s in spk) for spk in self.speakers]
        jobs = [(spk_set, seg) for spk_set in self._spk_sets]
        dists = [self._distance(t) for t in self.eng.pair_terms(jobs, want_glr=(o.distance == 'GLR'),
                                                                want_kl2=(o.distance == 'KL2'))]
        mind = MAXINT
        best = None
        for k, d in enumerate(dists):
            if o.tt:
                self.say('Time:', end, '- Distance:', d, '- Speaker:', k + 1)
            if not _isinf(d):
                if d > self.max_dist:
                    self.max_dist = d
                if d < self.min_dist:
                    self.min_dist = d
                if d < mind:
                    mind = d
                    best = k
        if mind <= o.threshold:
            self.speakers[best].append((start, end))
            self._spk_sets[best] = self._spk_sets[best] + seg
            writer.write(rl, start, end, 0, 'speaker_' + str(best + 1))
        else:
            self.speakers.append([(start, end)])
            self._spk_sets.append(seg)
            writer.write(rl, start, end, 0, 'speaker_' + str(len(self.speakers)))

    # ----------------------------------------------------------------- hierarchical
    def _cluster_hi(self, recipe, writer):
        """CL1:178-260 / CL2:173-229: device merge loop + host replay."""
        o = self.o
        segs = [self._frames(s[0][0], s[0][1]) for s in self.speakers]
        res = self.eng.cluster_hi(segs, variant=o.variant, kind=o.distance,
                                  lambdac=o.lambdac, threshold=o.threshold,
                                  max_spk=o.max_spk)
        for (a, b, mind) in res.merges:
            self.say('Merging:', a + 1, 'and', b + 1, 'distance:', mind)
            self.speakers[a].extend(self.speakers[b])
            self.speakers.pop(b)
        if o.variant == 1:
            if res.max_dist is not None and res.max_dist > self.max_dist:
                self.max_dist = res.max_dist
            if res.min_dist is not None and res.min_dist < self.min_dist:
                self.min_dist = res.min_dist
        else:
            self.max_dist = res.max_dist
            self.min_dist = res.min_dist
        self.say('Final speakers:', len(self.speakers))
        turns = [(turn, k) for k, spk in enumerate(self.speakers) for turn in spk]
        turns.sort(key=lambda x: x[0])
        for turn, k in turns:
            writer.write(recipe[turn[2]], turn[0], turn[1], 0, 'speaker_' + str(k + 1))

    # ----------------------------------------------------------------- summary
    def print_summary(self, n_recipe_lines):
        """CL1:436-442 / CL2:399-406."""
        say = self.say
        say('Useful metrics for determining the right threshold:')
        say('---------------------------------------------------')
        say('Maximum between segments distance:', self.max_dist)
        if self.min_dist < MAXINT:
            say('Minimum between segments distance:', self.min_dist)
        say('Total segments:', n_recipe_lines)
        say('Total detected speakers:', len(self.speakers))
