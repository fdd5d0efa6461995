"""In-memory batch pipeline: growing-window BIC change detection followed by
agglomerative clustering for MANY files in a handful of device launches, keeping
the semantics of the two-script pipeline spk-diarization2.py runs
(spk-diarization2.py:122-128), including the 12-significant-digit text contract
between the two stages: every boundary the change detector emits
is formatted like the recipe writer would and re-parsed like the clustering
script would, so segment frame ranges are the ones the file-based path gets.

All files of a batch live in one resident frame array [sum T, 39]; turns and
segments are absolute frame ranges into it.  One k_gw launch covers every turn of
every file, one k_chunk_stats/k_reduce_sets pair every segment, one
k_cluster_prep/k_matrix/k_ahc triple every file (a clustering problem each).
"""
import numpy as np

from . import hipabi
from .recipe import py2_float_str

DIA2_CD = dict(kind='BIC', lambdac=1.0, threshold=0.0, winsize_s=1.0, winstep_s=3.0, deltaws_s=0.1)
DIA2_CL = dict(variant=1, kind='BIC', lambdac=1.3, threshold=0.0, max_spk=0)


def _roundtrip(values):
    """float(str(x)) with Python-2 str(): what the next stage reads back."""
    return [float(py2_float_str(v)) for v in values]


class BatchFile(object):
    """One file of a batch: frame window in the resident array + its VAD turns
    (start / end seconds as the VAD recipe states them)."""

    def __init__(self, frame_off, n_frames, vad):
        self.frame_off = int(frame_off)
        self.n_frames = int(n_frames)
        self.vad = [(float(s), float(e)) for (s, e) in vad]


def change_detect_batch(ctx, d_frames, total_frames, files, rate=125.0, cd=DIA2_CD, timings=None):
    """Returns, per file, the list of (start_s, end_s) the change-detection recipe
    would contain (already passed through the 12-digit text round trip)."""
    rate = float(rate)
    tb, te, owner, lna0, lna1 = [], [], [], [], []
    for fi, f in enumerate(files):
        for (s, e) in f.vad:
            f0 = min(int(s * rate), f.n_frames)
            f1 = max(f0, min(int(e * rate), f.n_frames))
            tb.append(f.frame_off + f0); te.append(f.frame_off + f1)
            owner.append(fi); lna0.append(s); lna1.append(e)
    p = hipabi.CdParams(hipabi.KINDS[cd['kind']], 0, cd['lambdac'], cd['threshold'],
                        float(np.floor(cd['winsize_s'] * rate)), float(np.floor(cd['winstep_s'] * rate)),
                        float(np.floor(rate * cd['deltaws_s'])), rate)
    r = ctx.gw(d_frames, total_frames, tb, te, p, log_cap=4096)
    if timings is not None:
        timings.setdefault('gw', []).append(ctx.last_ms('gw'))
        timings['gw_frames'] = int(sum(y - x for x, y in zip(tb, te)))
        timings['gw_windows'] = int(r['n_win'].sum())
    if r['status'] == hipabi.SPKD_ENONFINITE:
        raise ValueError('array must not contain infs or NaNs')
    off = r['off']
    out = [[] for _ in files]
    for t in range(len(tb)):
        o = int(off[t])
        nd = int(r['win_det'][o:o + int(r['n_win'][t])].sum())
        ls, le = lna0[t], lna1[t]
        starts = r['det_start'][o:o + nd]
        ends = starts + r['det_maxi'][o:o + nd]
        vals = []
        for k in range(nd):
            vals.append(starts[k] / rate + ls)
            vals.append(ends[k] / rate + ls)
        vals.append(float(r['final_start'][t]) / rate + ls)
        vals.append(((le - ls) * rate) / rate + ls)
        rt = _roundtrip(vals)
        lines = out[owner[t]]
        for k in range(nd + 1):
            lines.append((rt[2 * k], rt[2 * k + 1]))
    return out


def cluster_batch(ctx, d_frames, total_frames, files, segments, rate=125.0, cl=DIA2_CL, timings=None):
    """segments: per file, [(start_s, end_s)] as the clustering script parses
    them.  Returns per file (labels[list of int, 1-based, per segment in input
    order], merges[(a, b, d)])."""
    rate = float(rate)
    b, e = [], []
    seg_off = [0]
    for f, segs in zip(files, segments):
        for (s, t) in segs:
            a0 = max(0, min(int(s * rate), f.n_frames))
            a1 = max(a0, min(int(t * rate), f.n_frames))
            b.append(f.frame_off + a0); e.append(f.frame_off + a1)
        seg_off.append(len(b))
    n = len(b)
    d_stats = ctx.dev_alloc(max(n, 1) * hipabi.REC * 8)
    try:
        ctx.set_stats(d_frames, total_frames, b, e, np.arange(n, dtype=np.int32), n, d_stats)
        if timings is not None:
            timings.setdefault('chunk_stats', []).append(ctx.last_ms('chunk_stats'))
            timings.setdefault('reduce_sets', []).append(ctx.last_ms('reduce_sets'))
            timings['stats_frames'] = int(sum(y - x for x, y in zip(b, e)))
            timings['stats_sets'] = n
        p = hipabi.AhcParams(cl['variant'], hipabi.KINDS[cl['kind']], cl['max_spk'], 0,
                             cl['lambdac'], cl['threshold'])
        r = ctx.ahc(d_stats, seg_off, p)
        if timings is not None:
            for k in ('cluster_prep', 'matrix', 'ahc'):
                timings.setdefault(k, []).append(ctx.last_ms(k))
            npb = np.diff(np.asarray(seg_off, dtype=np.int64))
            nm = r['n_merges'].astype(np.int64)
            timings['matrix_pairs'] = int((npb * (npb - 1) // 2).sum())
            # merge m of a problem with N records recomputes N - 1 - (m + 1) distances
            timings['ahc_pairs'] = int(sum(int(nm[i]) * (int(npb[i]) - 1) - int(nm[i]) * (int(nm[i]) + 1) // 2
                                           for i in range(len(npb))))
    finally:
        ctx.dev_free(d_stats)
    if r['status'] == hipabi.SPKD_ENONFINITE:
        raise ValueError('array must not contain infs or NaNs')
    out = []
    for fi in range(len(files)):
        o, cnt = seg_off[fi], seg_off[fi + 1] - seg_off[fi]
        clusters = [[k] for k in range(cnt)]
        merges = []
        for m in range(int(r['n_merges'][fi])):
            a, bb, d = int(r['a'][o + m]), int(r['b'][o + m]), float(r['d'][o + m])
            merges.append((a, bb, d))
            clusters[a].extend(clusters[bb])
            clusters.pop(bb)
        labels = [0] * cnt
        for k, members in enumerate(clusters):
            for s in members:
                labels[s] = k + 1
        out.append((labels, merges))
    return out


def diarize_batch(ctx, d_frames, total_frames, files, rate=125.0, cd=DIA2_CD, cl=DIA2_CL, timings=None):
    """CD (gw/BIC) + CL (hi/BIC) for a batch; returns per file
    [(start_s, end_s, speaker)] in recipe order."""
    segs = change_detect_batch(ctx, d_frames, total_frames, files, rate, cd, timings)
    res = cluster_batch(ctx, d_frames, total_frames, files, segs, rate, cl, timings)
    out = []
    for s, (labels, _) in zip(segs, res):
        order = sorted(range(len(s)), key=lambda k: (s[k][0] * rate, s[k][1] * rate, k))
        out.append([(s[k][0], s[k][1], labels[k]) for k in order])
    return out
