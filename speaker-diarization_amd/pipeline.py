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
import time

import numpy as np

from . import hipabi
from .recipe import py2_float_str

DIA2_CD = dict(kind='BIC', lambdac=1.0, threshold=0.0, winsize_s=1.0, winstep_s=3.0, deltaws_s=0.1)
DIA2_CL = dict(variant=1, kind='BIC', lambdac=1.3, threshold=0.0, max_spk=0)


class BatchFile(object):
    """One file of a batch: frame window in the resident array + its VAD turns
    (start / end seconds as the VAD recipe states them)."""

    def __init__(self, frame_off, n_frames, vad):
        self.frame_off = int(frame_off)
        self.n_frames = int(n_frames)
        self.vad = [(float(s), float(e)) for (s, e) in vad]
        self.vad_arr = np.array(self.vad, dtype=np.float64).reshape(-1, 2)     # (converted once, not per call)


class FusedStats(object):
    """What the fused change detector leaves for the clustering stage: per recipe line
    (same order as the concatenated per-file segment lists) the record index of its
    statistics in the device buffer and the absolute frame range those statistics
    cover."""

    def __init__(self, d_buf, n_buf, index, begin, end):
        self.d_buf, self.n_buf, self.index, self.begin, self.end = d_buf, n_buf, index, begin, end


def change_detect_batch(ctx, d_frames, total_frames, files, rate=125.0, cd=DIA2_CD, timings=None,
                        text_contract=True, fused=None):
    """Returns, per file, the list of (start_s, end_s) the change-detection recipe
    would contain (already passed through the 12-digit text round trip).
    text_contract=False is the opt-in fused mode of SURVEY.md §8(f) row 4: the times go
    to the clustering stage as the doubles they are, without being printed and re-read
    (A-2) -- NOT the reference's semantics: a boundary within 1e-12 relative of a frame
    edge can land one frame away.
    fused: a list; when given, the detector also leaves the statistics record of every
    segment on the device (spkd_gw_fused) and a FusedStats is appended to the list, so that
    cluster_batch does not read the frames a second time."""
    rate = float(rate)
    _t0 = time.perf_counter()
    nturn = [len(f.vad) for f in files]
    if sum(nturn) == 0:
        return [[] for _ in files]
    vad = np.concatenate([f.vad_arr for f in files])
    owner = np.repeat(np.arange(len(files)), nturn)
    foff = np.array([f.frame_off for f in files], dtype=np.int64)[owner]
    fn = np.array([f.n_frames for f in files], dtype=np.int64)[owner]
    ls, le = vad[:, 0], vad[:, 1]
    f0 = np.minimum((ls * rate).astype(np.int64), fn)            # int() truncation + slice clamp
    f1 = np.maximum(f0, np.minimum((le * rate).astype(np.int64), fn))
    tb, te = foff + f0, foff + f1
    p = hipabi.CdParams(hipabi.KINDS[cd['kind']], 0, cd['lambdac'], cd['threshold'],
                        float(np.floor(cd['winsize_s'] * rate)), float(np.floor(cd['winstep_s'] * rate)),
                        float(np.floor(rate * cd['deltaws_s'])), rate)
    _t1 = time.perf_counter()
    seg_buf = {}

    def seg_alloc(n_rec):
        seg_buf['n'] = n_rec
        seg_buf['p'] = ctx.dev_scratch('fused_segment_stats', max(n_rec, 1) * hipabi.REC * 8)
        return seg_buf['p']

    r = ctx.gw(d_frames, total_frames, tb, te, p, log_cap=4096, tight=True, reuse=True,
               seg_stats=seg_alloc if fused is not None else None)
    _t2 = time.perf_counter()
    if timings is not None:
        timings.setdefault('gw', []).append(ctx.last_ms('gw'))
        timings.setdefault('gw_stream_ms', []).append(ctx.last_ms('call'))     # uploads + kernel + result copies
        timings['gw_frames'] = int((te - tb).sum())
        timings['gw_windows'] = int(r['n_win'].sum())
        timings['gw_dets'] = ctx.last_gw_items()
    if r['status'] == hipabi.SPKD_ENONFINITE:
        raise ValueError('array must not contain infs or NaNs')
    off = r['off']
    nt = len(tb)
    # detections per turn = ones among the turn's n_win window flags (what lies behind them
    # in the reused buffers is not looked at)
    nd = hipabi.count_flags(r['win_det'], off[:-1], r['n_win'])
    # recipe order: per turn its detections (detection j of turn t sits at off[t] + j), then the
    # tail line; times as the script computes them, through the 12-digit text round trip
    lines = hipabi.gw_lines(off[:-1], nd, r['det_start'], r['det_maxi'], r['final_start'], ls, le, tb, te, rate,
                            text_contract=text_contract, want_frames=fused is not None)
    rt, line_turn = lines['times'], lines['turn']
    if fused is not None:
        # the frames each record covers: [int(start), int(start + maxi)) of the turn for a
        # detection, [int(final start), turn end) for the tail
        fused.append(FusedStats(seg_buf['p'], seg_buf['n'], lines['index'], lines['frame_b'], lines['frame_e']))
    line_file = owner[line_turn]
    bounds = np.searchsorted(line_file, np.arange(len(files) + 1))
    out = [rt[bounds[i]:bounds[i + 1]] for i in range(len(files))]
    if timings is not None:
        _t3 = time.perf_counter()
        timings.setdefault('wall_cd_prepare', []).append(1e3 * (_t1 - _t0))
        timings.setdefault('wall_cd_call', []).append(1e3 * (_t2 - _t1))
        timings.setdefault('wall_cd_finish', []).append(1e3 * (_t3 - _t2))
    return out


def segment_stats(ctx, d_frames, total_frames, files, segments, rate=125.0, timings=None, fused=None,
                  scratch_name='segment_stats'):
    """The statistics record of every segment (get_spk_features + the np.cov inputs,
    spk-clustering.py:46-52, 88-94) -> (device pointer to n records in segment order,
    seg_off per file, n, time stamp after the host preparation).  fused: see cluster_batch."""
    cnt = [len(s) for s in segments]
    seg_off = np.zeros(len(files) + 1, dtype=np.int64)
    seg_off[1:] = np.cumsum(cnt)
    n = int(seg_off[-1])
    allseg = np.concatenate([np.asarray(s, dtype=np.float64).reshape(-1, 2) for s in segments]) if n else np.zeros((0, 2))
    owner = np.repeat(np.arange(len(files)), cnt)
    foff = np.array([f.frame_off for f in files], dtype=np.int64)[owner]
    fn = np.array([f.n_frames for f in files], dtype=np.int64)[owner]
    a0 = np.clip((allseg[:, 0] * rate).astype(np.int64), 0, fn)
    a1 = np.maximum(a0, np.clip((allseg[:, 1] * rate).astype(np.int64), 0, fn))
    b, e = foff + a0, foff + a1
    d_stats = ctx.dev_scratch(scratch_name, max(n, 1) * hipabi.REC * 8)
    _t1 = time.perf_counter()
    if fused is None:
        ctx.set_stats(d_frames, total_frames, b, e, np.arange(n, dtype=np.int32), n, d_stats)
        redo = np.arange(n)
    else:
        same = (fused.begin == b) & (fused.end == e)
        keep = np.nonzero(same)[0]
        redo = np.nonzero(~same)[0]
        ctx.gather_stats(fused.d_buf, fused.n_buf, fused.index[keep], d_stats, n, keep)
        if len(redo):
            d_tmp = ctx.dev_scratch(scratch_name + '_redo', len(redo) * hipabi.REC * 8)
            ctx.set_stats(d_frames, total_frames, b[redo], e[redo], np.arange(len(redo), dtype=np.int32),
                          len(redo), d_tmp)
            ctx.gather_stats(d_tmp, len(redo), np.arange(len(redo)), d_stats, n, redo)
    if timings is not None:
        if len(redo):
            timings.setdefault('chunk_stats', []).append(ctx.last_ms('chunk_stats'))
            timings.setdefault('reduce_sets', []).append(ctx.last_ms('reduce_sets'))
        timings['stats_frames'] = int((e[redo] - b[redo]).sum())
        timings['stats_sets'] = n
        timings['stats_recomputed'] = int(len(redo))
    return d_stats, seg_off, n, _t1


def cluster_batch(ctx, d_frames, total_frames, files, segments, rate=125.0, cl=DIA2_CL, timings=None,
                  want_merges=False, fused=None):
    """segments: per file, array [(start_s, end_s)] as the clustering script parses
    them.  Returns per file (labels[int array, 1-based, per segment in input
    order], merges[(a, b, d)]).
    fused: the FusedStats of change_detect_batch for exactly these segments: the records
    of the segments whose frame range (as computed here, from the times) equals the range
    the detector summed are gathered from its buffer; only the others -- a boundary the
    12-digit text round trip moved across a frame edge -- are computed from the frames."""
    rate = float(rate)
    # files without any segment are not clustering problems (spkd_ahc rejects empty ones)
    if fused is None and any(len(sg) == 0 for sg in segments):
        keep = [i for i, sg in enumerate(segments) if len(sg) > 0]
        out = [(np.zeros(0, dtype=np.int32), [] if want_merges else None) for _ in files]
        if keep:
            sub = cluster_batch(ctx, d_frames, total_frames, [files[i] for i in keep],
                                [segments[i] for i in keep], rate, cl, timings, want_merges)
            for i, r in zip(keep, sub):
                out[i] = r
        return out
    _t0 = time.perf_counter()
    cnt = [len(s) for s in segments]
    d_stats, seg_off, n, _t1 = segment_stats(ctx, d_frames, total_frames, files, segments, rate, timings, fused)
    _t2 = time.perf_counter()
    p = hipabi.AhcParams(cl['variant'], hipabi.KINDS[cl['kind']], cl['max_spk'], cl.get('path', 0),
                         cl['lambdac'], cl['threshold'])
    r = ctx.ahc(d_stats, seg_off, p)
    _t3 = time.perf_counter()
    if timings is not None:
        for k in ('cluster_prep', 'matrix', 'ahc'):
            timings.setdefault(k, []).append(ctx.last_ms(k))
        npb = np.diff(seg_off)
        nm = r['n_merges'].astype(np.int64)
        timings['matrix_pairs'] = int((npb * (npb - 1) // 2).sum())
        # merge m of a problem with N records recomputes N - 2 - m distances
        timings['ahc_pairs'] = int((nm * (npb - 2) - nm * (nm - 1) // 2).sum())
    if r['status'] == hipabi.SPKD_ENONFINITE:
        raise ValueError('array must not contain infs or NaNs')
    all_labels = hipabi.labels_from_merges_batch(seg_off, r['n_merges'], r['a'], r['b'])
    out = []
    for fi in range(len(files)):
        o, c = int(seg_off[fi]), cnt[fi]
        merges = None
        if want_merges:
            nm = int(r['n_merges'][fi])
            merges = list(zip(r['a'][o:o + nm].tolist(), r['b'][o:o + nm].tolist(), r['d'][o:o + nm].tolist()))
        out.append((all_labels[o:o + c], merges))
    if timings is not None:
        _t4 = time.perf_counter()
        timings.setdefault('wall_cl_prepare', []).append(1e3 * (_t1 - _t0))
        timings.setdefault('wall_cl_stats_call', []).append(1e3 * (_t2 - _t1))
        timings.setdefault('wall_cl_ahc_call', []).append(1e3 * (_t3 - _t2))
        timings.setdefault('wall_cl_finish', []).append(1e3 * (_t4 - _t3))
    return out


def diarize_batch(ctx, d_frames, total_frames, files, rate=125.0, cd=DIA2_CD, cl=DIA2_CL, timings=None,
                  text_contract=True, fused=False):
    """CD (gw/BIC) + CL (hi/BIC) for a batch; returns per file an array of rows
    [start_s, end_s, speaker] in recipe order.
    fused=True: the frames are read once -- the change detector leaves every segment's
    statistics record for the clustering stage (segments and labels are those of the
    two-pass form; a record differs from the two-pass one only in the order of its
    floating-point sums)."""
    box = [] if fused else None
    segs = change_detect_batch(ctx, d_frames, total_frames, files, rate, cd, timings, text_contract, box)
    fs = box[0] if box else None
    if fused and fs is None:                 # no turn at all in the batch
        return [np.zeros((0, 3)) for _ in files]
    if fused and any(len(sg) == 0 for sg in segs):
        # files without segments are not clustering problems: drop them, keep the line order
        keep = [i for i, sg in enumerate(segs) if len(sg) > 0]
        out = [np.zeros((0, 3)) for _ in files]
        if keep:
            sub = _cluster_and_order(ctx, d_frames, total_frames, [files[i] for i in keep],
                                     [segs[i] for i in keep], rate, cl, timings, fs)
            for i, rws in zip(keep, sub):
                out[i] = rws
        return out
    return _cluster_and_order(ctx, d_frames, total_frames, files, segs, rate, cl, timings, fs)


def _cluster_and_order(ctx, d_frames, total_frames, files, segs, rate, cl, timings, fs):
    res = cluster_batch(ctx, d_frames, total_frames, files, segs, rate, cl, timings, fused=fs)
    # recipe order of spk_cluster_hi's output: per file, sorted by (start*rate, end*rate, line)
    cnt = [len(s) for s in segs]
    n = sum(cnt)
    if n == 0:
        return [np.zeros((0, 3)) for _ in segs]
    allseg = np.concatenate([np.asarray(s, dtype=np.float64).reshape(-1, 2) for s in segs])
    labels = np.concatenate([lab for (lab, _) in res]).astype(np.float64)
    owner = np.repeat(np.arange(len(segs)), cnt)
    k0, k1 = allseg[:, 0] * rate, allseg[:, 1] * rate
    # the change detector emits a file's lines in time order, so the keys are almost always
    # sorted already (a stable sort then changes nothing): one O(n) look instead of the sort
    in_order = (owner[1:] != owner[:-1]) | (k0[1:] > k0[:-1]) | ((k0[1:] == k0[:-1]) & (k1[1:] >= k1[:-1]))
    if bool(in_order.all()):
        rows = np.column_stack([allseg, labels])
    else:
        order = np.lexsort((np.arange(n), k1, k0, owner))
        rows = np.column_stack([allseg[order], labels[order]])
    bounds = np.zeros(len(segs) + 1, dtype=np.int64)
    bounds[1:] = np.cumsum(cnt)
    out = [rows[bounds[i]:bounds[i + 1]] for i in range(len(segs))]
    return out


def in_flight(contexts, n_jobs, job):
    """Runs job(ctx, k) for k = 0 .. n_jobs - 1 with one job in flight per context -- a host
    thread per context, each taking the next k when it is free -- and yields the results in
    order of k.  With two contexts on streams of their own the host part of one batch (recipe
    text, label replay) runs under the kernels of the next, and the tail of one launch beside
    the head of the next: + 5 - 10 % throughput on 256-hour batches (DESIGN.md par. 5).  A
    context is only ever used by its own thread; an exception in a job is re-raised here."""
    import threading
    if len(contexts) == 1 or n_jobs <= 1:
        for k in range(n_jobs):
            yield job(contexts[0], k)
        return
    results = [None] * n_jobs
    done = [threading.Event() for _ in range(n_jobs)]
    nxt = [0]
    lock = threading.Lock()
    failure = []

    def worker(ctx):
        while not failure:
            with lock:
                k = nxt[0]
                nxt[0] += 1
            if k >= n_jobs:
                return
            try:
                results[k] = job(ctx, k)
            except BaseException as e:               # surfaces on the consuming thread
                failure.append(e)
            finally:
                done[k].set()

    threads = [threading.Thread(target=worker, args=(c,)) for c in contexts]
    for th in threads:
        th.start()
    try:
        for k in range(n_jobs):
            done[k].wait()
            if failure:
                break
            r, results[k] = results[k], None
            yield r
    finally:
        if not failure:
            failure.append(None)                     # (an abandoned run: the workers stop after their current job)
        for th in threads:
            th.join()
    if failure and failure[0] is not None:
        raise failure[0]
