"""HipEngine: the product's numerics provider.  Implements the engine interface the
host drivers call (set_features / gw / sw / pair_terms / cluster_hi) on top of
libspkd_hip.so.  There is no CPU path here: if the library is missing or no GPU
is present the constructor raises."""
import math

import numpy as np

from . import hipabi
from .results import GwTurnResult, HiResult, PairTerms


def _f(x):
    return float(x)


class HipEngine(object):
    def __init__(self, device=0, stream=None):
        self.ctx = hipabi.Context(device, stream)
        self.d_frames = None
        self.n_frames = 0
        self._owned = []
        self.last_ms = {}
        self.ahc_path = hipabi.AHC_AUTO     # AHC_MONO / AHC_WIDE force one launch shape
        # pair_terms: statistics records of the frame sets seen so far on this file, on the
        # device, keyed by the set itself.  The host-driven modes ask for the same sets again
        # and again (spk_cluster_in: every cluster against every new segment, one cluster changed
        # since the last call; merge_rec: the next line of the recipe); a record depends on its
        # set alone, so a cached one is the recomputed one bit for bit.
        self._rec_cap = 4096
        self._rec_buf = None
        self._rec_slot = {}

    # ------------------------------------------------------------- memory
    def _forget_records(self):
        self._rec_slot = {}

    def close(self):
        if self._rec_buf is not None:
            self.ctx.dev_free(self._rec_buf)
            self._rec_buf = None
        for p in self._owned:
            self.ctx.dev_free(p)
        self._owned = []
        self.d_frames = None
        self.ctx.close()

    def set_features(self, feats):
        """Uploads one file's frames (float32 [T, 39]) to HBM; stays resident
        for every later call (the reference keeps the array in RAM: CD:367-369)."""
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        if feats.ndim != 2 or feats.shape[1] != hipabi.DIM:
            raise ValueError('expected [T, %d] features, got %r' % (hipabi.DIM, feats.shape))
        if self.d_frames is not None:
            self.ctx.dev_free(self.d_frames)
            self._owned.remove(self.d_frames)
        self.d_frames = self.ctx.dev_alloc(max(feats.nbytes, 16))
        self._owned.append(self.d_frames)
        if feats.nbytes:
            self.ctx.h2d(self.d_frames, feats)
        self.n_frames = feats.shape[0]
        self._forget_records()

    def set_device_features(self, d_ptr, n_frames):
        """Use frames already resident in HBM (e.g. ``tensor.data_ptr()``)."""
        if self.d_frames is not None and self.d_frames in self._owned:
            self.ctx.dev_free(self.d_frames)
            self._owned.remove(self.d_frames)
        self.d_frames = d_ptr
        self.n_frames = n_frames
        self._forget_records()

    @staticmethod
    def _raise_nonfinite(st):
        if st == hipabi.SPKD_ENONFINITE:
            # what scipy.linalg.det does to the reference on a NaN covariance
            raise ValueError('array must not contain infs or NaNs')

    # ------------------------------------------------------------- stats of sets
    def _stats_of_sets(self, sets):
        """sets: list of lists of (begin, end).  Returns device pointer (owned by
        the caller until freed) to len(sets) records."""
        b, e, s = [], [], []
        for k, ranges in enumerate(sets):
            for (x, y) in ranges:
                b.append(x); e.append(y); s.append(k)
        d_stats = self.ctx.dev_alloc(max(len(sets), 1) * hipabi.REC * 8)
        try:
            self.ctx.set_stats(self.d_frames, self.n_frames, b, e, s, len(sets), d_stats)
        except Exception:
            self.ctx.dev_free(d_stats)
            raise
        return d_stats

    def stats(self, sets):
        d = self._stats_of_sets(sets)
        try:
            out = np.empty((len(sets), hipabi.REC), dtype=np.float64)
            if len(sets):
                self.ctx.d2h(out, d)
        finally:
            self.ctx.dev_free(d)
        return out

    # ------------------------------------------------------------- pair terms
    def _record_slots(self, sets):
        """Slots (in the engine's record buffer) of the statistics records of `sets`; the missing
        ones are computed, in one launch, into the next free slots."""
        keys = [r if type(r) is tuple else tuple(tuple(x) for x in r) for r in sets]      # (a tuple: of tuples)
        missing = []
        for k in keys:
            if k not in self._rec_slot and k not in missing:
                missing.append(k)
        if self._rec_buf is None or len(self._rec_slot) + len(missing) > self._rec_cap:
            # full (or first use): start over with room for this call
            self._rec_cap = max(self._rec_cap, 2 * len(keys))
            if self._rec_buf is not None:
                self.ctx.dev_free(self._rec_buf)
            self._rec_buf = self.ctx.dev_alloc(self._rec_cap * hipabi.REC * 8)
            self._rec_slot = {}
            missing = []
            for k in keys:
                if k not in missing:
                    missing.append(k)
        if missing:
            b, e, s_ = [], [], []
            for i, ranges in enumerate(missing):
                for (x, y) in ranges:
                    b.append(x); e.append(y); s_.append(i)
            first = len(self._rec_slot)
            self.ctx.set_stats(self.d_frames, self.n_frames, b, e, s_, len(missing),
                               self._rec_buf + first * hipabi.REC * 8)
            for i, k in enumerate(missing):
                self._rec_slot[k] = first + i
        return [self._rec_slot[k] for k in keys]

    def pair_terms(self, jobs, want_glr=False, want_kl2=False):
        sets = []
        for ra, rb in jobs:
            sets.append(ra); sets.append(rb)
        slots = self._record_slots(sets)
        flags = (hipabi.WANT_GLR if want_glr else 0) | (hipabi.WANT_KL2 if want_kl2 else 0)
        out, st = self.ctx.pair_terms(self._rec_buf, slots[0::2], slots[1::2], flags)
        self._raise_nonfinite(st)
        res = []
        for row in out:
            res.append(PairTerms(int(row[0]), int(row[1]), _f(row[2]), _f(row[3]), _f(row[4]),
                                 _f(row[5]) if want_glr else None,
                                 _f(row[6]) if want_kl2 else None))
        return res

    def cluster_in(self, segs, kind, lambdac, threshold):
        """spk_cluster_in over the segments `segs` (frame ranges, recipe order) as one device
        chain (spkd_cluster_in): (labels, distances per segment, segments done).  done < len(segs):
        a covariance with infs or NaNs at segment `done` -- the caller goes on from there the slow
        way, which raises where the reference does."""
        d = self._stats_of_sets([[s] for s in segs])
        try:
            label, dists, done, _, _ = self.ctx.cluster_in(d, len(segs), kind, lambdac, threshold)
        finally:
            self.ctx.dev_free(d)
        return label, dists, done

    # ------------------------------------------------------------- growing window
    def gw(self, turns, kind, lambdac, threshold, winsize, winstep, deltaws, rate, trace=False):
        p = hipabi.CdParams(hipabi.KINDS[kind], 1 if trace else 0, lambdac, threshold, winsize,
                            winstep, deltaws, rate)
        b = [t[0] for t in turns]
        e = [t[1] for t in turns]
        r = self.ctx.gw(self.d_frames, self.n_frames, b, e, p,
                        log_cap=(1 << 20) if trace else 4096)
        self.last_ms['gw'] = self.ctx.last_ms()
        self._raise_nonfinite(r['status'])
        logs = {}
        for k in range(r['log_count']):
            rec = r['log'][k]
            logs.setdefault(rec.turn, []).append(rec)
        out = []
        for t in range(len(turns)):
            o = int(r['off'][t])
            recs = sorted(logs.get(t, []), key=lambda x: x.seq)
            by_win = {}
            for rec in recs:
                by_win.setdefault(rec.seq >> 32, []).append(rec)
            events = []
            nd = 0
            for w in range(int(r['n_win'][t])):
                cands = by_win.get(w, [])
                for rec in cands:
                    if rec.coarse:
                        events.append(('cand', rec.start, rec.i, int(rec.n1), int(rec.n2), rec.d, True))
                maxd = _f(r['win_maxd'][o + w])
                events.append(('win', None if math.isnan(maxd) else maxd))
                for rec in cands:
                    if not rec.coarse:
                        events.append(('cand', rec.start, rec.i, int(rec.n1), int(rec.n2), rec.d, False))
                if r['win_det'][o + w]:
                    events.append(('det', _f(r['det_start'][o + nd]), _f(r['det_maxi'][o + nd]),
                                   _f(r['det_d'][o + nd])))
                    nd += 1
            out.append(GwTurnResult(events, _f(r['final_start'][t])))
        return out

    # ------------------------------------------------------------- sliding window
    def sw(self, turns, kind, lambdac, winsize, winstep):
        p = hipabi.CdParams(hipabi.KINDS[kind], 0, lambdac, 0.0, winsize, winstep, 0.0, 125.0)
        b = [t[0] for t in turns]
        e = [t[1] for t in turns]
        st, off, d = self.ctx.sw(self.d_frames, self.n_frames, b, e, p)
        self.last_ms['sw'] = self.ctx.last_ms()
        self._raise_nonfinite(st)
        return [d[int(off[t]):int(off[t + 1])] for t in range(len(turns))]

    # ------------------------------------------------------------- hierarchical
    def cluster_hi(self, segs, variant, kind, lambdac, threshold, max_spk):
        d = self._stats_of_sets([[s] for s in segs])
        self.last_ms['stats'] = self.ctx.last_ms()
        try:
            p = hipabi.AhcParams(variant, hipabi.KINDS[kind], max_spk, self.ahc_path, lambdac, threshold)
            r = self.ctx.ahc(d, [0, len(segs)], p)
            self.last_ms['ahc'] = self.ctx.last_ms()
        finally:
            self.ctx.dev_free(d)
        self._raise_nonfinite(r['status'])
        nm = int(r['n_merges'][0])
        merges = [(int(r['a'][m]), int(r['b'][m]), _f(r['d'][m])) for m in range(nm)]
        smax, smin = _f(r['stat_max'][0]), _f(r['stat_min'][0])
        if variant == 1:
            smax = None if math.isnan(smax) else smax
            smin = None if math.isnan(smin) else smin
        return HiResult(merges, smax, smin)
