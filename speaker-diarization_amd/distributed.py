"""File-sharded multi-GPU execution: one process per GPU, no collective on the
data path (files are independent units); the only exchange is the
gather of the finished recipes on rank 0, done with torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

RCCL has no native gather of ragged byte strings, so the exchange is two
all_gathers: the byte counts, then the payloads padded to the longest one
(KB-scale messages: latency bound, one collective pair per batch)."""
import numpy as np


def shard(items, rank, world):
    """Round-robin: file i goes to rank i mod world (each rank keeps the order)."""
    return [(i, it) for i, it in enumerate(items) if i % world == rank]


def _device_for(dist):
    import torch
    if dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def _gather_payloads(payload, dist):
    """payload: uint8 array of this rank.  Returns on rank 0 the list of every rank's payload
    (memoryviews, in rank order), on the others None.  Two all_gathers: sizes, padded bytes."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _device_for(dist)
    size = torch.tensor([payload.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    maxlen = max(max(sizes), 1)
    buf = torch.zeros(maxlen, dtype=torch.uint8, device=dev)
    if payload.size:
        buf[:payload.size] = torch.from_numpy(np.ascontiguousarray(payload).copy()).to(dev)
    bufs = [torch.zeros(maxlen, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(bufs, buf)
    if rank != 0:
        return None
    return [memoryview(bufs[r].cpu().numpy())[:sizes[r]] for r in range(world)]


def _pack(items):
    """items: [(file_index, bytes-like)] -> uint8 payload: n | (index, length) x n | the blobs"""
    head = np.array([[i, len(b)] for (i, b) in items], dtype=np.int64).reshape(-1, 2)
    return np.frombuffer(np.int64(len(items)).tobytes() + head.tobytes() + b''.join(bytes(b) for (_, b) in items),
                         dtype=np.uint8)


def _unpack(raw):
    """inverse of _pack on a memoryview: yields (file_index, memoryview of the blob)"""
    n = int(np.frombuffer(raw[:8], dtype=np.int64)[0])
    head = np.frombuffer(raw[8:8 + 16 * n], dtype=np.int64).reshape(n, 2)
    pos = 8 + 16 * n
    for i, ln in head:
        yield int(i), raw[pos:pos + int(ln)]
        pos += int(ln)


def gather_texts(local, dist=None):
    """local: list of (file_index, text) produced by this rank.  Returns on rank 0
    the dict {file_index: text} of ALL ranks, on the others None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(local)
    got = _gather_payloads(_pack([(i, t.encode('utf-8')) for (i, t) in local]), dist)
    if got is None:
        return None
    return {i: bytes(b).decode('utf-8') for raw in got for (i, b) in _unpack(raw)}


def gather_rows(local, dist=None):
    """local: list of (file_index, float64 array [n, 3]) -> on rank 0 {file_index: array}.
    The rows travel as their bytes (a text encoding cost rank 0 tens of milliseconds per
    step at eight ranks of 256 files)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return {i: np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 3) for i, a in local}
    got = _gather_payloads(_pack([(i, np.ascontiguousarray(a, dtype=np.float64).tobytes()) for i, a in local]), dist)
    if got is None:
        return None
    return {i: np.frombuffer(b, dtype=np.float64).reshape(-1, 3) for raw in got for (i, b) in _unpack(raw)}


def recipe_text(audio, rows, lna_prefix='a'):
    """rows: [(start_s, end_s, speaker)] -> clustering-stage recipe text
    (writer grammar of spk-clustering.py:66-70, lna renamed a_1, a_2, ...)."""
    from .recipe import py2_float_str
    return ''.join('audio=%s lna=%s_%d start-time=%s end-time=%s speaker=speaker_%d\n' % (
        audio, lna_prefix, k + 1, py2_float_str(s), py2_float_str(e), int(spk))
        for k, (s, e, spk) in enumerate(rows))


# ---------------------------------------------------------------------------
# One LONG file over several GPUs (BASELINE config 5).
#
#   contiguous time shards of the frame array, cut between VAD turns (a segment lies inside a
#   turn, so no segment straddles two shards)
#     -> per rank: change detection + statistics records of ITS turns' segments
#     -> all_gather of the records (N x 6 560 B) and of the segment times
#     -> per rank: a block of rows of the N x N distance matrix, balanced over the triangle
#        (the outer loop of spk-clustering.py:188-200 is what is being split)
#     -> all_gather of the blocks (N x N x 8 B) and of the blocks' max / min
#     -> the merge loop (spk-clustering.py:201-240) replicated on every rank: it is a serial
#        chain of dependent steps, more GPUs cannot shorten it, and replicated it needs no
#        further exchange -- every rank ends with the same rows.
# Every record and every matrix entry is computed by the same kernel code on the same inputs
# whichever rank owns it, so the result equals the one-GPU result bit for bit.
# ---------------------------------------------------------------------------
def shard_turns(vad, world, rate=125.0):
    """vad: [(start_s, end_s)] in time order.  Returns per rank (turn_lo, turn_hi): contiguous
    runs of turns with about equal frame counts."""
    n = len(vad)
    w = np.array([max(0.0, (e - s) * rate) for (s, e) in vad], dtype=np.float64)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, cum[-1] * r / world)))
    cuts.append(n)
    cuts = np.maximum.accumulate(np.clip(cuts, 0, n))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def row_blocks(n, world):
    """Rows of the upper triangle of an n x n matrix (row a holds n - 1 - a pairs) cut into
    `world` contiguous blocks of about equal pair counts -> [(row_begin, row_end)]."""
    if n <= 0:
        return [(0, 0)] * world
    cost = np.arange(n - 1, -1, -1, dtype=np.float64)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, cum[-1] * r / world)))
    cuts.append(n)
    cuts = np.maximum.accumulate(np.clip(cuts, 0, n))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def _all_gather_rows(local, width, dist):
    """local: float64 array [n_local, width] on the host.  Returns the concatenation over the
    ranks (rank order), on every rank, and the per-rank counts.  Two all_gathers: counts,
    padded payloads (on the GPU for RCCL, on the host for gloo)."""
    import torch
    world = dist.get_world_size()
    dev = _device_for(dist)
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    cnts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    cnts = [int(x.item()) for x in cnts]
    nmax = max(max(cnts), 1)
    buf = torch.zeros((nmax, width), dtype=torch.float64, device=dev)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64)).to(dev)
    bufs = [torch.zeros((nmax, width), dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(bufs, buf)
    return torch.cat([bufs[r][:cnts[r]] for r in range(world)]), cnts


def _all_gather_device(ctx, d_ptr, n_local, width, dist, name):
    """The same for records that live in device memory of the library's context (d_ptr: n_local x
    width doubles).  Returns (device pointer to all rows in rank order, total rows, keep-alive)."""
    import torch
    world = dist.get_world_size()
    dev = _device_for(dist)
    cnt = torch.tensor([n_local], dtype=torch.int64, device=dev)
    cnts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    cnts = [int(x.item()) for x in cnts]
    nmax, total = max(max(cnts), 1), sum(cnts)
    if dev.type == 'cuda':
        # RCCL: straight from / to device memory (a device-to-device copy through the library
        # brings the records into a tensor torch.distributed can send)
        buf = torch.zeros((nmax, width), dtype=torch.float64, device=dev)
        if n_local:
            ctx.copy_d2d(buf.data_ptr(), d_ptr, n_local * width * 8)
        bufs = [torch.empty((nmax, width), dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(bufs, buf)
        allr = torch.cat([bufs[r][:cnts[r]] for r in range(world)]) if total else torch.zeros((1, width), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        return allr.data_ptr(), total, allr
    host = np.zeros((nmax, width), dtype=np.float64)
    if n_local:
        ctx.d2h(host[:n_local], d_ptr)
    buf = torch.from_numpy(host)
    bufs = [torch.zeros((nmax, width), dtype=torch.float64) for _ in range(world)]
    dist.all_gather(bufs, buf)
    allh = np.concatenate([bufs[r][:cnts[r]].numpy() for r in range(world)]) if total else np.zeros((1, width))
    d_all = ctx.dev_scratch(name, max(total, 1) * width * 8)
    ctx.h2d(d_all, np.ascontiguousarray(allh))
    return d_all, total, None


def diarize_long_file(ctx, d_shard, shard_frame0, shard_frames, file_frames, vad_local, dist=None, rate=125.0,
                      cd=None, cl=None, timings=None, want_merges=False):
    """One long file whose frames are sharded over the ranks of `dist` (module docstring of this
    section).  d_shard: this rank's frames [shard_frame0, shard_frame0 + shard_frames) of the
    file, in its GPU's memory; vad_local: the VAD turns that lie in it ((start_s, end_s) in file
    time, as the VAD recipe states them).  Returns on EVERY rank the rows [start_s, end_s,
    speaker] of the whole file in recipe order (and the merge log with want_merges); with
    dist = None or one rank it is the single-GPU pipeline through the same entry points."""
    import time as _time
    from . import hipabi, pipeline
    cd = cd or pipeline.DIA2_CD
    cl = cl or pipeline.DIA2_CL
    multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    world = dist.get_world_size() if multi else 1
    rank = dist.get_rank() if multi else 0
    tm = {} if timings is None else timings
    clock = _time.perf_counter
    t0 = clock()
    # ---- this rank's turns: change detection + the statistics record of every segment
    f = pipeline.BatchFile(-int(shard_frame0), file_frames, vad_local)
    box = []
    segs = pipeline.change_detect_batch(ctx, d_shard, shard_frames, [f], rate, cd, tm, True, box)[0] if vad_local else []
    segs = np.asarray(segs, dtype=np.float64).reshape(-1, 2)
    t1 = clock()
    n_loc = len(segs)
    d_loc = 0
    if n_loc:
        d_loc, _, _, _ = pipeline.segment_stats(ctx, d_shard, shard_frames, [f], [segs], rate, tm, box[0] if box else None,
                                                scratch_name='long_file_local_stats')
    t2 = clock()
    # ---- exchange 1: every rank gets every record and every segment
    keep = None
    if multi:
        d_all, n, keep = _all_gather_device(ctx, d_loc, n_loc, hipabi.REC, dist, 'long_file_all_stats')
        allseg, _ = _all_gather_rows(segs, 2, dist)
        allseg = allseg.cpu().numpy()
    else:
        d_all, n, allseg = d_loc, n_loc, segs
    t3 = clock()
    if n == 0:
        return (np.zeros((0, 3)), []) if want_merges else np.zeros((0, 3))
    # ---- this rank's block of matrix rows
    variant, kind = cl['variant'], cl['kind']
    rb, re_ = row_blocks(n, world)[rank]
    d_rows = ctx.dev_scratch('long_file_rows', max(re_ - rb, 1) * n * 8)
    bmax, bmin = ctx.distance_rows(variant, kind, cl['lambdac'], d_all, n, rb, re_, d_rows)
    tm.setdefault('matrix_rows', []).append(ctx.last_ms('matrix'))
    t4 = clock()
    # ---- exchange 2: the blocks -> the whole matrix on every rank
    import torch
    keep2 = None
    if multi:
        d_up, n_rows, keep2 = _all_gather_device(ctx, d_rows, re_ - rb, n, dist, 'long_file_matrix_upper')
        assert n_rows == n
        mm, _ = _all_gather_rows(np.array([[bmax, bmin]]), 2, dist)
        mm = mm.cpu().numpy()
        smax = float(np.nanmax(mm[:, 0])) if np.any(mm[:, 0] == mm[:, 0]) else float('nan')
        smin = float(np.nanmin(mm[:, 1])) if np.any(mm[:, 1] == mm[:, 1]) else float('nan')
    else:
        d_up, smax, smin = d_rows, bmax, bmin
    t5 = clock()
    # ---- the full initial matrix in the form the merge loop expects, then the merge loop
    if not multi:
        d_M = d_up                                         # one rank: its "block" is the whole matrix, mirror included
    elif keep2 is not None:                                # RCCL: assemble on the device
        U = keep2
        if variant == 1:
            up = torch.triu(U, 1)
            M = up + up.t()
            M.fill_diagonal_(9223372036854775808.0)        # float(sys.maxint)
        else:
            M = U
        M = M.contiguous()
        torch.cuda.synchronize()
        d_M = M.data_ptr()
    else:
        host = np.empty((n, n), dtype=np.float64)
        ctx.d2h(host, d_up)
        if variant == 1:
            up = np.triu(host, 1)
            host = up + up.T
            np.fill_diagonal(host, 9223372036854775808.0)
        d_M = ctx.dev_scratch('long_file_matrix', n * n * 8)
        ctx.h2d(d_M, np.ascontiguousarray(host))
    p = hipabi.AhcParams(variant, hipabi.KINDS[kind], cl['max_spk'], cl.get('path', 0), cl['lambdac'], cl['threshold'])
    r = ctx.ahc_matrix(d_all, n, p, d_M, smax, smin)
    tm.setdefault('ahc', []).append(ctx.last_ms('ahc'))
    if r['status'] == hipabi.SPKD_ENONFINITE:
        raise ValueError('array must not contain infs or NaNs')
    t6 = clock()
    seg_off = np.array([0, n], dtype=np.int64)
    labels = hipabi.labels_from_merges_batch(seg_off, r['n_merges'], r['a'], r['b']).astype(np.float64)
    k0, k1 = allseg[:, 0] * rate, allseg[:, 1] * rate
    order = np.lexsort((np.arange(n), k1, k0))             # spk_cluster_hi's output order (spk-clustering.py:244-246)
    rows = np.column_stack([allseg[order], labels[order]])
    t7 = clock()
    tm['long_file_ms'] = {'change_detection': 1e3 * (t1 - t0), 'segment_stats': 1e3 * (t2 - t1),
                          'gather_records': 1e3 * (t3 - t2), 'matrix_rows': 1e3 * (t4 - t3),
                          'gather_rows': 1e3 * (t5 - t4), 'assemble_and_merge_loop': 1e3 * (t6 - t5),
                          'labels': 1e3 * (t7 - t6), 'total': 1e3 * (t7 - t0),
                          'segments': int(n), 'local_segments': int(n_loc), 'rows_of_this_rank': int(re_ - rb),
                          'stat_max': float(r['stat_max'][0]), 'stat_min': float(r['stat_min'][0])}
    del keep, keep2
    if want_merges:
        nm = int(r['n_merges'][0])
        return rows, list(zip(r['a'][:nm].tolist(), r['b'][:nm].tolist(), r['d'][:nm].tolist()))
    return rows
