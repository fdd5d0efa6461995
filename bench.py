#!/usr/bin/env python3
"""Benchmark of the hot path: growing-window BIC change detection + agglomerative
BIC clustering (the exact flags spk-diarization2.py passes) over a batch of
synthetic 1 h / 4-speaker feature files resident in HBM.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torchrun sets RANK / LOCAL_RANK / WORLD_SIZE).  Files are
independent, so ranks share nothing on the data path: each rank owns its own
batch ("weak" scaling) and the only collective is the max-reduce of the timing.
A step = one pass of CD + CL over the rank's whole batch.  Rank 0 prints ONE JSON
line: value = audio hours processed by all ranks per second.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = 'speaker-diarization_amd'

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_FRAME = 156          # 39 float32, each frame read once per stage
BYTES_PER_PAIR = 13128         # two 6 560-B records in, one double out
REC_BYTES = 6560


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--files', type=int, default=256, help='1 h files per GPU per step')
    ap.add_argument('--distinct', type=int, default=4, help='distinct synthetic sessions (tiled to --files)')
    ap.add_argument('--seconds', type=float, default=3600.0)
    ap.add_argument('--speakers', type=int, default=4)
    ap.add_argument('--cpu-sample-seconds', type=float, default=900.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--ahc-path', type=int, default=0, help='0 auto, 1 one workgroup per file, 2 chained launches')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo for rehearsals)')
    ap.add_argument('--share-device', action='store_true',
                    help='rehearsal only: every rank uses GPU 0 (needs --backend gloo)')
    return ap.parse_args()


def cpu_baseline(args, synth, cli):
    """The numpy oracle (np.cov + det per distance, like the reference) on a
    bounded sample of the same workload, on this box's host cores."""
    import io
    import tempfile
    from oracle.numpy_engine import NumpyEngine
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = 1
    secs = args.cpu_sample_seconds
    feats, vad, _ = synth.make_session(777, secs, args.speakers)
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, 'fea'))
        synth.write_fea(os.path.join(tmp, 'fea', 's.fea'), feats)
        with open(os.path.join(tmp, 'vad.recipe'), 'w') as f:
            f.write(synth.vad_recipe_text('s.wav', vad))
        eng = NumpyEngine()
        t0 = time.perf_counter()
        cli.main_change_detection([os.path.join(tmp, 'vad.recipe'), os.path.join(tmp, 'fea') + '/', '-o',
                                   os.path.join(tmp, 'spkc.recipe'), '-m', 'gw', '-d', 'BIC', '-w', '1.0',
                                   '-st', '3.0', '-dws', '0.1', '-l', '1.0'], engine=eng, stdout=io.StringIO())
        cli.main_clustering([os.path.join(tmp, 'spkc.recipe'), os.path.join(tmp, 'fea') + '/', '-o',
                             os.path.join(tmp, 'out.recipe'), '-m', 'hi', '-l', '1.3'], variant=1,
                            engine=eng, stdout=io.StringIO())
        dt = time.perf_counter() - t0
    return {'value': (secs / 3600.0) / dt, 'unit': 'hours-audio/s', 'cores': int(cores), 'kind': 'port',
            'sample': '%.0f s of the same synthetic %d-speaker audio, CD gw/BIC + CL hi/BIC through '
                      'oracle/numpy_engine.py (np.cov + det per distance), %.1f s wall' % (secs, args.speakers, dt)}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (torch.cuda.is_available() is False)')
    if args.share_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)
    hipabi = importlib.import_module(PKG + '.hipabi')
    synth = importlib.import_module(PKG + '.synth')
    pipeline = importlib.import_module(PKG + '.pipeline')
    cli = importlib.import_module(PKG + '.cli')

    # ---- synthetic batch, resident in HBM before the timed region
    sessions = []
    for i in range(max(1, min(args.distinct, args.files))):
        feats, vad, _ = synth.make_session(1000003 * (rank + 1) + i, args.seconds, args.speakers)
        sessions.append((feats, [(s / 125.0, e / 125.0) for (s, e) in vad]))
    T = sessions[0][0].shape[0]
    # the distinct sessions go up once and are tiled on the device (an 18 GB host copy
    # per rank would be 144 GB on an 8-GPU node)
    dev_sessions = [torch.from_numpy(s[0]).to(dev) for s in sessions]
    frames = torch.cat([dev_sessions[i % len(sessions)] for i in range(args.files)])
    del dev_sessions
    files = []
    for i in range(args.files):
        vad = sessions[i % len(sessions)][1]
        # times as the VAD recipe would state them (12 significant digits)
        rec = importlib.import_module(PKG + '.recipe')
        vad = [(float(rec.py2_float_str(s)), float(rec.py2_float_str(e))) for (s, e) in vad]
        files.append(pipeline.BatchFile(i * T, T, vad))
    total = args.files * T
    stream = torch.cuda.current_stream().cuda_stream
    ctx = hipabi.Context(local, stream)
    ptr = frames.data_ptr()

    distributed = importlib.import_module(PKG + '.distributed')

    def step(tm=None):
        rows = pipeline.diarize_batch(ctx, ptr, total, files, cl=dict(pipeline.DIA2_CL, path=args.ahc_path), timings=tm)
        if world == 1:
            return rows
        # the one exchange of the multi-GPU path: finished recipes -> rank 0 (RCCL)
        return distributed.gather_rows([(rank + world * i, r) for i, r in enumerate(rows)], dist)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    out = None
    for _ in range(args.warmup):
        out = step()
    timings = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(timings)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tdt = torch.tensor([dt], device=dev if args.backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        dt = float(tdt.item())
        if rank == 0:
            assert out is not None and len(out) == world * args.files, 'recipe gather incomplete'
    hours = world * args.files * (args.seconds / 3600.0) * args.steps
    value = hours / dt

    if rank == 0:
        avg = lambda k: float(np.mean(timings[k])) if timings.get(k) else 0.0
        kernels = {
            'k_gw': (avg('gw'), timings.get('gw_frames', 0) * BYTES_PER_FRAME),
            'k_chunk_stats': (avg('chunk_stats'), timings.get('stats_frames', 0) * BYTES_PER_FRAME +
                              timings.get('stats_sets', 0) * REC_BYTES),
            'k_matrix': (avg('matrix'), timings.get('matrix_pairs', 0) * BYTES_PER_PAIR),
            'k_ahc': (avg('ahc'), timings.get('ahc_pairs', 0) * BYTES_PER_PAIR),
        }
        dom = max(kernels, key=lambda k: kernels[k][0])
        per_kernel = {}
        for k, (ms, byts) in kernels.items():
            gbs = (byts / 1e9) / (ms / 1e3) if ms > 0 else 0.0
            per_kernel[k] = {'ms_per_launch': round(ms, 4), 'algorithmic_bytes': int(byts),
                             'achieved_GBps': round(gbs, 2), 'frac_of_hbm_peak': round(gbs / HBM_PEAK_GBS, 5)}
        dms, dbytes = kernels[dom]
        achieved = (dbytes / 1e9) / (dms / 1e3) if dms > 0 else 0.0
        pair_rate = 0.0
        if avg('matrix') > 0:
            pair_rate = timings.get('matrix_pairs', 0) / (avg('matrix') / 1e3)
        # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, separate runs; 2 x FETCH + WRITE: the gfx950 FETCH_SIZE halving was
        # calibrated on k_chunk_stats, whose read volume is known exactly) -- only valid
        # for the workload those passes were taken on
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'r01e_bench256_hbm_traffic.json')
        if (args.files, args.seconds, args.speakers) == (256, 3600.0, 4) and os.path.exists(tfile):
            with open(tfile) as f:
                tk = json.load(f)['kernels'].get(dom)
            if tk:
                traffic = int(tk['hbm_bytes_per_launch_fetch_doubled'])
        res = {
            'metric': 'diarized audio throughput (CD gw/BIC + CL hi/BIC)',
            'value': value, 'unit': 'hours-audio/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'xRT': value * 3600.0,
            'bic_pair_dists_per_s': pair_rate,
            'config': {'workload': '%d x %.0f s synthetic 16 kHz-equivalent features (39-dim, 125 fps), '
                                   '%d speakers, per GPU per step; %d distinct sessions tiled; DIA2 flags '
                                   '(CD -m gw -d BIC -w 1.0 -st 3.0 -dws 0.1 -l 1.0; CL -m hi -l 1.3)'
                                   % (args.files, args.seconds, args.speakers, len(sessions)),
                       'files_per_gpu': args.files, 'frames_per_file': int(T),
                       'segments_per_step': timings.get('stats_sets', 0),
                       'parallelism': 'file-sharded x%d, no data-path collective' % world},
            'roofline': {'kernel': dom, 'bound': 'hbm', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic},
            'kernels': per_kernel,
            'wall_ms': {k[5:]: round(float(np.mean(v)), 2) for k, v in timings.items() if k.startswith('wall_')},
            'device_ms_per_step': round(sum(v[0] for v in kernels.values()) + avg('cluster_prep') + avg('reduce_sets'), 3),
        }
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(args, synth, cli)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
