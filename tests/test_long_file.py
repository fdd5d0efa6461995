"""One long file over several ranks (SURVEY.md 8(e) row 2 / BASELINE config 5):
contiguous time shards -> per-shard change detection and statistics -> all_gather of the
records -> a row block of the N x N matrix per rank -> all_gather of the blocks -> the merge
loop replicated.  The loops being split are spk-clustering.py:188-200 (initial matrix) and
:201-240 (merge loop)."""
import importlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT

dmod = importlib.import_module('speaker-diarization_amd.distributed')


def test_row_blocks_cover_the_triangle_in_balance():
    for n in (0, 1, 2, 7, 387, 3862):
        for world in (1, 2, 3, 8):
            blocks = dmod.row_blocks(n, world)
            assert len(blocks) == world
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            if n >= 64:
                pairs = [sum(n - 1 - a for a in range(b, e)) for b, e in blocks]
                assert max(pairs) - min(pairs) <= 2 * n      # a cut is off by at most one row on either side


def test_shard_turns_are_contiguous_and_complete():
    vad = [(float(i * 10), float(i * 10 + 3 + (i % 5))) for i in range(57)]
    for world in (1, 2, 5, 8, 64):
        sh = dmod.shard_turns(vad, world)
        assert sh[0][0] == 0 and sh[-1][1] == len(vad)
        assert all(sh[i][1] == sh[i + 1][0] and sh[i][0] <= sh[i][1] for i in range(world - 1))
    assert dmod.shard_turns([], 3) == [(0, 0)] * 3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_two_ranks_equal_one_rank_equal_the_batch_pipeline(tmp_path):
    """1 h / 8 speakers (383 segments): two ranks (gloo, CPU collectives, sharing the box's one
    GPU) against one rank through the same entry points against the ordinary batch pipeline
    (spkd_ahc): rows, merge sequence, merge distances and the distance statistics bit for bit."""
    import torch
    hipabi = importlib.import_module('speaker-diarization_amd.hipabi')
    pipeline = importlib.import_module('speaker-diarization_amd.pipeline')
    synth = importlib.import_module('speaker-diarization_amd.synth')
    rec = importlib.import_module('speaker-diarization_amd.recipe')
    seed, seconds, nspk = 8088, 3600, 8
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    child = os.path.join(ROOT, 'tests', 'long_file_child.py')
    two, one = os.path.join(str(tmp_path), 'two.npz'), os.path.join(str(tmp_path), 'one.npz')
    r2 = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), child,
                         str(seed), str(seconds), str(nspk), two, '--gloo', '--share-device'],
                        cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    r1 = subprocess.run([sys.executable, child, str(seed), str(seconds), str(nspk), one],
                        cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-3000:]
    a, b = np.load(two), np.load(one)
    # the ordinary pipeline on the whole file
    feats, vad, _ = synth.make_session(seed, seconds, nspk)
    vad_t = [(float(rec.py2_float_str(s / 125.0)), float(rec.py2_float_str(e / 125.0))) for (s, e) in vad]
    fr = torch.from_numpy(feats).cuda()
    ctx = hipabi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        f = [pipeline.BatchFile(0, feats.shape[0], vad_t)]
        box = []
        segs = pipeline.change_detect_batch(ctx, fr.data_ptr(), feats.shape[0], f, fused=box)
        res = pipeline.cluster_batch(ctx, fr.data_ptr(), feats.shape[0], f, segs, want_merges=True, fused=box[0])
        rows = pipeline.diarize_batch(ctx, fr.data_ptr(), feats.shape[0], f, fused=True)[0]
    finally:
        ctx.close()
    merges = np.array(res[0][1], dtype=np.float64).reshape(-1, 3)
    assert len(rows) > 300 and len(merges) > 300
    for got in (a, b):
        assert np.array_equal(got['rows'], rows)
        assert np.array_equal(got['merges'], merges)          # (a, b, distance): bit for bit
    assert np.array_equal(a['stat'], b['stat'])
    print(r2.stdout.strip().splitlines()[-1])
    print(r1.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize('variant,kind,threshold', [(2, 'BIC', 0.0), (1, 'GLR', 1500.0), (2, 'KL2', 30.0)])
def test_row_blocks_and_supplied_matrix_equal_spkd_ahc(variant, kind, threshold):
    """The two halves a long file is tiled with -- spkd_distance_rows over three row blocks
    (assembled as the ranks would) and spkd_ahc_matrix on the result -- against spkd_ahc on
    the same records, for the other variant and the other distances: merge log, distance
    statistics and the assembled matrix itself bit for bit (spk-clustering2.py:173-222 for
    variant 2: lower triangle +inf, the merge loop rewrites rows only)."""
    hipabi = importlib.import_module('speaker-diarization_amd.hipabi')
    synth = importlib.import_module('speaker-diarization_amd.synth')
    feats, vad, truth = synth.make_session(4242, 150, 3)
    segs = [(s, e) for (s, e, _) in truth]
    n = len(segs)
    ctx = hipabi.Context(0)
    try:
        d_fr = ctx.dev_alloc(feats.nbytes)
        ctx.h2d(d_fr, feats)
        d_st = ctx.dev_alloc(n * hipabi.REC * 8)
        ctx.set_stats(d_fr, feats.shape[0], [s for s, _ in segs], [e for _, e in segs], np.arange(n, dtype=np.int32), n, d_st)
        p = hipabi.AhcParams(variant, hipabi.KINDS[kind], 0, hipabi.AHC_WIDE, 1.3, threshold)
        ref = ctx.ahc(d_st, [0, n], p)
        blocks = dmod.row_blocks(n, 3)
        full = np.empty((n, n), dtype=np.float64)
        smax, smin = [], []
        for rb, re_ in blocks:
            d_rows = ctx.dev_alloc(max(re_ - rb, 1) * n * 8)
            a, b = ctx.distance_rows(variant, kind, 1.3, d_st, n, rb, re_, d_rows)
            part = np.empty((re_ - rb, n), dtype=np.float64)
            if re_ > rb:
                ctx.d2h(part, d_rows)
            full[rb:re_] = part
            ctx.dev_free(d_rows)
            smax.append(a); smin.append(b)
        if variant == 1:
            up = np.triu(full, 1)
            full = up + up.T
            np.fill_diagonal(full, 9223372036854775808.0)
        d_whole = ctx.dev_alloc(n * n * 8)
        ctx.distance_rows(variant, kind, 1.3, d_st, n, 0, n, d_whole)
        whole = np.empty((n, n), dtype=np.float64)
        ctx.d2h(whole, d_whole)
        assert np.array_equal(full, whole, equal_nan=True)           # blocks assembled == one block
        d_m = ctx.dev_alloc(n * n * 8)
        ctx.h2d(d_m, np.ascontiguousarray(full))
        fmax = np.nanmax(smax) if np.any(np.array(smax) == np.array(smax)) else float('nan')
        fmin = np.nanmin(smin) if np.any(np.array(smin) == np.array(smin)) else float('nan')
        got = ctx.ahc_matrix(d_st, n, p, d_m, fmax, fmin)
        nm = int(ref['n_merges'][0])
        assert nm > 5 and int(got['n_merges'][0]) == nm
        for k in ('a', 'b', 'd'):
            assert np.array_equal(ref[k][:nm], got[k][:nm], equal_nan=True), k
        assert np.array_equal(ref['stat_max'], got['stat_max'], equal_nan=True)
        assert np.array_equal(ref['stat_min'], got['stat_min'], equal_nan=True)
    finally:
        ctx.close()
