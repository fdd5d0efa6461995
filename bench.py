#!/usr/bin/env python3
"""Benchmark of the hot path: growing-window BIC change detection + agglomerative
BIC clustering (the exact flags spk-diarization2.py passes) over a batch of
synthetic 1 h / 4-speaker feature files resident in HBM.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torchrun sets RANK / LOCAL_RANK / WORLD_SIZE).  Files are
independent, so ranks share nothing on the data path: each rank owns its own
batch ("weak" scaling) and the only collective is the gather of the finished
recipes on rank 0 (+ the max-reduce of the timing).  A step = one pass of CD + CL
over the rank's whole batch.  Rank 0 prints ONE JSON line: value = audio hours
processed by all ranks per second.

What the line carries besides the contract's keys:
  roofline            the kernel with the largest share of the step: algorithmic bytes
                       / its HIP-event duration on the launch stream;
                      `traffic` only from a PMC profile of THIS build (the profile's
                      kernel time must match the one measured now within 5 %), else null
  kernels             the same for every hot kernel
  hbm_copy_GBps       device-to-device copy rate measured in this run (the achievable
                      HBM rate next to the 8 TB/s spec)
  verified            the batch result was compared with the file-based drop-in scripts
                      before timing, and every timed step reproduced the same digest
  other_configs       latency of BASELINE.json configs 2, 3 and 5 as single-file calls
  cpu_baseline        oracle/numpy_engine.py (np.cov + det per distance, like the
                      reference) on a bounded sample, this box's host cores
  cpu_baseline_stats  oracle/spkd_oracle.c (fp64 sufficient statistics, the algorithmic
                      twin of the GPU path) on one thread and on the job's CPU share (<= 16)
"""
import argparse
import hashlib
import importlib
import json
import os
import re
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
# fp64 work ("report achieved fp64 FLOP/s alongside"): useful multiply-adds,
# not issued lanes.  A 39x39 symmetric determinant: sum_j j (39 - j) = 9 880 FMAs of
# elimination + 780 of forming the covariance (the rank-one mean correction) + 780 of the
# linear combination of the records = 11 440 FMAs = 22 880 flop; a frame added to the
# running moment sums: 820 FMAs = 1 640 flop.  Peak: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x
# 2.4 GHz = 78.6 TFLOP/s (fp64 vector; the fp64 matrix rate is the same on gfx950).
FLOP_PER_DET = 2 * (9880 + 780 + 780)
FLOP_PER_FRAME = 2 * 820
FP64_PEAK_TFLOPS = 78.6
TRAFFIC_PROFILE = os.path.join(ROOT, 'profiles', 'r03_bench256_hbm_traffic.json')

CD_ARGS = ['-m', 'gw', '-d', 'BIC', '-w', '1.0', '-st', '3.0', '-dws', '0.1', '-l', '1.0']
CL_ARGS = ['-m', 'hi', '-l', '1.3']


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--files', type=int, default=256, help='1 h files per GPU per step')
    ap.add_argument('--distinct', type=int, default=0,
                    help='distinct synthetic sessions (0 = every file its own; fewer are tiled)')
    ap.add_argument('--seconds', type=float, default=3600.0)
    ap.add_argument('--speakers', type=int, default=4)
    ap.add_argument('--in-flight', type=int, default=1,
                    help='batches in flight per GPU: that many host threads, each with a library context on a '
                         'stream of its own, take the steps in turn; kernels of different batches then share '
                         'the GPU (tails overlap with the next launch) and their HIP-event durations are no '
                         'longer those of a kernel alone.  Default 1: strictly one step at a time; the '
                         'two-in-flight rate is reported beside it as "two_batches_in_flight"')
    ap.add_argument('--two-pass', action='store_true',
                    help='read the frames once per stage (k_chunk_stats) instead of the fused single read')
    ap.add_argument('--cpu-sample-seconds', type=float, default=900.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the other-config latencies and the copy rate')
    ap.add_argument('--ahc-path', type=int, default=0, help='0 auto, 1 one workgroup per file, 2 chained launches')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo for rehearsals)')
    ap.add_argument('--share-device', action='store_true',
                    help='rehearsal only: every rank uses GPU 0 (needs --backend gloo)')
    ap.add_argument('--dump-rows', default='', help='rank 0 writes the gathered rows of the last step here (.npz)')
    ap.add_argument('--long-file', action='store_true',
                    help='BASELINE config 5 instead of the batch: ONE file of --seconds (default 36000) / --speakers '
                         '(default 8) sharded in time over the ranks (distributed.diarize_long_file: all_gather of the '
                         'segment records, a row block of the N x N matrix per rank, merge loop replicated); the line '
                         'carries the per-phase times of the slowest rank')
    return ap.parse_args()


def cpu_info():
    model = ''
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def _write_session(tmp, synth, feats, vad):
    os.makedirs(os.path.join(tmp, 'fea'), exist_ok=True)
    synth.write_fea(os.path.join(tmp, 'fea', 's.fea'), feats)
    with open(os.path.join(tmp, 'vad.recipe'), 'w') as f:
        f.write(synth.vad_recipe_text('s.wav', vad))


def _run_scripts(tmp, cli, eng):
    import io
    cli.main_change_detection([os.path.join(tmp, 'vad.recipe'), os.path.join(tmp, 'fea') + '/', '-o',
                               os.path.join(tmp, 'spkc.recipe')] + CD_ARGS, engine=eng, stdout=io.StringIO())
    cli.main_clustering([os.path.join(tmp, 'spkc.recipe'), os.path.join(tmp, 'fea') + '/', '-o',
                         os.path.join(tmp, 'out.recipe')] + CL_ARGS, variant=1, engine=eng, stdout=io.StringIO())
    return open(os.path.join(tmp, 'out.recipe')).read()


def cpu_baseline(args, synth, cli):
    """The numpy oracle (np.cov + det per distance, like the reference) on a
    bounded sample of the same workload, on this box's host cores."""
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
        _write_session(tmp, synth, feats, vad)
        t0 = time.perf_counter()
        _run_scripts(tmp, cli, NumpyEngine())
        dt = time.perf_counter() - t0
    model, ncpu = cpu_info()
    return {'value': (secs / 3600.0) / dt, 'unit': 'hours-audio/s', 'cores': int(cores), 'kind': 'port',
            'cpu': model, 'host_cpus': ncpu,
            'sample': '%.0f s of the same synthetic %d-speaker audio, CD gw/BIC + CL hi/BIC through '
                      'oracle/numpy_engine.py (np.cov + det per distance; BLAS threads = cores), %.1f s wall'
                      % (secs, args.speakers, dt)}


def cpu_baseline_stats(args, synth, cli):
    """The C restatement on sufficient statistics (the fair comparison for the GPU kernels:
    most of the gap to the numpy path is algorithmic), one thread and all cores, on one
    whole file of the benchmark's workload."""
    import tempfile
    from oracle.c_engine import COracleEngine
    feats, vad, _ = synth.make_session(777, args.seconds, args.speakers)
    out = {'unit': 'hours-audio/s', 'kind': 'port', 'cpu': cpu_info()[0],
           'sample': 'one %.0f s %d-speaker file, CD gw/BIC + CL hi/BIC through oracle/spkd_oracle.c '
                     '(turns and pair distances spread over OpenMP / host threads); job_cpu_share = the CPUs this '
                     'job may use on the box, at most 16 (gpurun gives a one-GPU job a 16-CPU share of the host) '
                     '-- NOT all %d host CPUs' % (args.seconds, args.speakers, os.cpu_count() or 1)}
    with tempfile.TemporaryDirectory() as tmp:
        _write_session(tmp, synth, feats, vad)
        for key, threads in (('one_thread', 1), ('job_cpu_share', 0)):
            eng = COracleEngine(threads)
            t0 = time.perf_counter()
            _run_scripts(tmp, cli, eng)
            dt = time.perf_counter() - t0
            out[key] = {'value': (args.seconds / 3600.0) / dt, 'threads': int(eng.threads), 'wall_s': round(dt, 2)}
    return out


def rows_digest(rows_by_file):
    h = hashlib.sha256()
    for i in sorted(rows_by_file):
        h.update(np.int64(i).tobytes())
        h.update(np.ascontiguousarray(rows_by_file[i], dtype=np.float64).tobytes())
    return h.hexdigest()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process, which has not touched
    the GPU (torch is not even imported yet), starts N ranks of itself as a FRESH child
    process tree through torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1), relays what rank 0 prints and exits with the children's code.  Never an
    exec: the parent stays alive as a plain waiter."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, cwd=os.getcwd())
    try:
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        return proc.wait()


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch
    import torch.distributed as dist
    if world != max(1, args.gpus):
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.long_file:
        return long_file_main(args, rank, local, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (torch.cuda.is_available() is False)')
    if args.share_device:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit('bench.py: rank %d has no GPU (%d visible); one process per GPU'
                         % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)
    hipabi = importlib.import_module(PKG + '.hipabi')
    synth = importlib.import_module(PKG + '.synth')
    synth_device = importlib.import_module(PKG + '.synth_device')
    pipeline = importlib.import_module(PKG + '.pipeline')
    cli = importlib.import_module(PKG + '.cli')
    rec = importlib.import_module(PKG + '.recipe')
    distributed = importlib.import_module(PKG + '.distributed')
    engine_mod = importlib.import_module(PKG + '.engine')

    # ---- synthetic batch, generated on the device (bit-identical to synth.make_session),
    # resident in HBM before the timed region.  Seeds depend on the GLOBAL file index, so a
    # file is the same whichever rank owns it.
    n_distinct = args.files if args.distinct <= 0 else max(1, min(args.distinct, args.files))

    def vad_times(vad):
        # times as the VAD recipe would state them (12 significant digits)
        return [(float(rec.py2_float_str(s / 125.0)), float(rec.py2_float_str(e / 125.0))) for (s, e) in vad]

    sessions = []
    for i in range(n_distinct):
        gi = rank + world * i                        # global index of this rank's i-th file
        feats, vad, _ = synth_device.make_session_device(1000003 + gi, args.seconds, args.speakers, device=dev)
        sessions.append((feats, vad))
    T = int(sessions[0][0].shape[0])
    frames = torch.cat([sessions[i % n_distinct][0] for i in range(args.files)])
    files = [pipeline.BatchFile(i * T, T, vad_times(sessions[i % n_distinct][1])) for i in range(args.files)]
    first_host = (sessions[0][0].cpu().numpy(), sessions[0][1]) if rank == 0 else None
    del sessions
    total = args.files * T
    torch.cuda.synchronize()
    # The library launches on the stream it is given.  One context per batch in flight; with
    # more than one, each sits on a (non-blocking) torch stream of its own and is driven by
    # its own host thread (ctypes and numpy release the GIL).
    depth = max(1, args.in_flight)
    if depth == 1:
        lanes = [hipabi.Context(local, torch.cuda.current_stream().cuda_stream)]
        lane_streams = []
    else:
        lane_streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        lanes = [hipabi.Context(local, s.cuda_stream) for s in lane_streams]
    ctx = lanes[0]
    ptr = frames.data_ptr()
    fused = not args.two_pass
    cl = dict(pipeline.DIA2_CL, path=args.ahc_path)

    def compute(lane, tm=None, lanes_=None):
        """One pass of the hot path over the batch -> rows per local file."""
        return pipeline.diarize_batch((lanes_ or lanes)[lane], ptr, total, files, cl=cl, timings=tm, fused=fused)

    gather_s = []                                    # wall time of each gather on this rank

    def deliver(rows):
        if world == 1:
            return {i: r for i, r in enumerate(rows)}
        # the one exchange of the multi-GPU path: finished recipes -> rank 0 (RCCL)
        tg = time.perf_counter()
        got = distributed.gather_rows([(rank + world * i, r) for i, r in enumerate(rows)], dist)
        gather_s.append(time.perf_counter() - tg)
        return got

    def step(tm=None):
        return deliver(compute(0, tm))

    def run_steps(k_steps, tm, lanes_=None):
        """k_steps passes, at most one per lane in flight (pipeline.in_flight); the gathers run
        on this thread in step order (collectives must be issued in the same order on every
        rank)."""
        use = lanes_ or lanes
        tms = [{} for _ in use]
        lane_of = {id(c): i for i, c in enumerate(use)}

        def job(c, k):
            return pipeline.diarize_batch(c, ptr, total, files, cl=cl, timings=tms[lane_of[id(c)]], fused=fused)

        out_ = None
        for rows in pipeline.in_flight(use, k_steps, job):
            out_ = deliver(rows)
        for lane_tm in tms:                          # per-launch kernel times of all lanes
            for key, v in lane_tm.items():
                if isinstance(v, list):
                    tm.setdefault(key, []).extend(v)
                else:
                    tm[key] = v
        return out_

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # ---- before timing: the batch result of this rank's first file == the file-based
    # drop-in scripts on the same file (rank 0 only; the scripts re-read the .fea)
    verified = None
    out = step()
    if rank == 0:
        import re
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            _write_session(tmp, synth, first_host[0], first_host[1])
            eng = engine_mod.HipEngine(local)
            try:
                final = _run_scripts(tmp, cli, eng)
            finally:
                eng.close()
        want = [(float(a), float(b), int(c)) for a, b, c in
                re.findall(r'start-time=(\S+) end-time=(\S+) speaker=speaker_(\d+)', final)]
        got = [(a, b, int(c)) for a, b, c in out[0].tolist()]
        if got != want:
            raise SystemExit('bench.py: the batch pipeline and the file-based scripts disagree on file 0')
        verified = {'file0_rows': len(want), 'against': 'spk-change-detection.py + spk-clustering.py (file based, HIP engine)'}
    digest0 = rows_digest(out) if out is not None else None
    for _ in range(max(0, args.warmup - 1)):
        out = step()
    if depth > 1:                                    # every lane's scratch is sized before the clock starts
        for lane in range(1, depth):
            compute(lane)
    timings = {}
    barrier()
    del gather_s[:]
    t0 = time.perf_counter()
    out = run_steps(args.steps, timings)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0                # this rank's steps, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    ranks_info = None
    if world > 1:
        # [step time incl. the closing barrier, this rank's own time, its time inside the gathers]
        mine = torch.tensor([dt, dt_own, sum(gather_s)], device=dev if args.backend == 'nccl' else 'cpu',
                            dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = np.array([e.cpu().numpy() for e in every])
        dt = float(every[:, 0].max())
        k = max(1, args.steps)
        # a rank's gather time includes waiting for the slowest rank to arrive (compute skew):
        # the fastest rank's own time minus its gather time is the pure compute of the fastest,
        # the smallest gather time is the closest to the cost of the exchange itself
        ranks_info = {'ms_per_step_min': round(1e3 * float(every[:, 1].min()) / k, 3),
                      'ms_per_step_max': round(1e3 * float(every[:, 1].max()) / k, 3),
                      'compute_ms_per_step_min': round(1e3 * float((every[:, 1] - every[:, 2]).min()) / k, 3),
                      'compute_ms_per_step_max': round(1e3 * float((every[:, 1] - every[:, 2]).max()) / k, 3),
                      'gather_ms': round(1e3 * float(every[0, 2]) / k, 3),
                      'gather_ms_min_over_ranks': round(1e3 * float(every[:, 2].min()) / k, 3),
                      'gather_ms_max_over_ranks': round(1e3 * float(every[:, 2].max()) / k, 3)}
    if rank == 0:
        assert out is not None and len(out) == world * args.files, 'recipe gather incomplete'
        if rows_digest(out) != digest0:
            raise SystemExit('bench.py: a timed step produced different rows than the verified step')
        verified['digest'] = digest0[:16]
        verified['files_gathered'] = len(out)
        if args.dump_rows:
            np.savez(args.dump_rows, **{'f%d' % i: r for i, r in out.items()})
    hours = world * args.files * (args.seconds / 3600.0) * args.steps
    value = hours / dt

    if rank == 0:
        avg = lambda k: float(np.mean(timings[k])) if timings.get(k) else 0.0
        n_sets = timings.get('stats_sets', 0)
        gw_bytes = timings.get('gw_frames', 0) * BYTES_PER_FRAME + (n_sets * REC_BYTES if fused else 0)
        kernels = {
            'k_gw': (avg('gw'), gw_bytes),
            'k_chunk_stats': (avg('chunk_stats'), timings.get('stats_frames', 0) * BYTES_PER_FRAME +
                              (timings.get('stats_recomputed', n_sets)) * REC_BYTES),
            'k_matrix': (avg('matrix'), timings.get('matrix_pairs', 0) * BYTES_PER_PAIR),
            'k_ahc': (avg('ahc'), timings.get('ahc_pairs', 0) * BYTES_PER_PAIR),
        }
        flops = {
            'k_gw': timings.get('gw_dets', 0) * FLOP_PER_DET + timings.get('gw_frames', 0) * FLOP_PER_FRAME,
            'k_chunk_stats': timings.get('stats_frames', 0) * FLOP_PER_FRAME,
            'k_matrix': timings.get('matrix_pairs', 0) * FLOP_PER_DET,
            'k_ahc': timings.get('ahc_pairs', 0) * FLOP_PER_DET,
        }
        dom = max(kernels, key=lambda k: kernels[k][0])
        per_kernel = {}
        for k, (ms, byts) in kernels.items():
            gbs = (byts / 1e9) / (ms / 1e3) if ms > 0 else 0.0
            tfs = (flops[k] / 1e12) / (ms / 1e3) if ms > 0 else 0.0
            per_kernel[k] = {'ms_per_launch': round(ms, 4), 'algorithmic_bytes': int(byts),
                             'achieved_GBps': round(gbs, 2), 'frac_of_hbm_peak': round(gbs / HBM_PEAK_GBS, 5),
                             'fp64_flop': int(flops[k]), 'fp64_TFLOPs': round(tfs, 3),
                             'frac_of_fp64_peak': round(tfs / FP64_PEAK_TFLOPS, 4)}
        dms, dbytes = kernels[dom]
        achieved = (dbytes / 1e9) / (dms / 1e3) if dms > 0 else 0.0
        pair_rate = 0.0
        if avg('matrix') > 0:
            pair_rate = timings.get('matrix_pairs', 0) / (avg('matrix') / 1e3)
        # HBM traffic per launch from the PMC passes of THIS build (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, separate runs; 2 x FETCH + WRITE: the gfx950 FETCH_SIZE halving was
        # calibrated on k_chunk_stats, whose read volume is known exactly).  The profile stores
        # the kernel's duration; a build whose kernel runs at another speed is another build.
        traffic, traffic_note = None, 'no PMC profile for this workload'
        if (args.files, args.seconds, args.speakers, fused) == (256, 3600.0, 4, True) and os.path.exists(TRAFFIC_PROFILE):
            with open(TRAFFIC_PROFILE) as f:
                tk = json.load(f)['kernels'].get(dom)
            if tk and tk.get('kernel_ms') and abs(tk['kernel_ms'] - dms) <= 0.05 * dms:
                traffic = int(tk['hbm_bytes_per_launch_fetch_doubled'])
                traffic_note = 'profiles/%s (kernel %.1f ms there, %.1f ms now)' % (
                    os.path.basename(TRAFFIC_PROFILE), tk['kernel_ms'], dms)
            elif tk:
                traffic_note = 'profile is of another build (kernel %.1f ms there, %.1f ms now)' % (
                    tk.get('kernel_ms') or 0.0, dms)
        res = {
            'metric': 'diarized audio throughput (CD gw/BIC + CL hi/BIC)',
            'value': value, 'unit': 'hours-audio/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'xRT': value * 3600.0,
            'bic_pair_dists_per_s': pair_rate,
            'config': {'workload': '%d x %.0f s synthetic 16 kHz-equivalent features (39-dim, 125 fps), '
                                   '%d speakers, per GPU per step; %d distinct sessions; DIA2 flags '
                                   '(CD -m gw -d BIC -w 1.0 -st 3.0 -dws 0.1 -l 1.0; CL -m hi -l 1.3); %s'
                                   % (args.files, args.seconds, args.speakers, n_distinct,
                                      'frames read once (segment statistics emitted by the change detector)'
                                      if fused else 'frames read once per stage') +
                                   '; frames RESIDENT IN HBM when the timed region starts: reading the .fea files '
                                   'and the host-to-device copy are excluded (see value_incl_h2d)',
                       'files_per_gpu': args.files, 'frames_per_file': int(T),
                       'segments_per_step': n_sets,
                       'segments_recomputed_from_frames': timings.get('stats_recomputed', n_sets),
                       'parallelism': 'file-sharded x%d, no data-path collective' % world,
                       'batches_in_flight_per_gpu': depth},
            # achieved / frac: ALGORITHMIC bytes per launch over the measured launch
            # duration; traffic: counted HBM bytes per launch (PMC), traffic_GBps the same over
            # the same duration -- what the memory system actually carries for this kernel
            'roofline': {'kernel': dom, 'bound': 'hbm', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic,
                         'traffic_GBps': round(traffic / 1e9 / (dms / 1e3), 1) if traffic and dms > 0 else None,
                         'traffic_source': traffic_note,
                         # the same kernel against the fp64 vector peak (useful flop, see FLOP_PER_DET)
                         'flops': {'achieved': per_kernel[dom]['fp64_TFLOPs'], 'peak': FP64_PEAK_TFLOPS,
                                   'unit': 'TFLOP/s', 'frac': per_kernel[dom]['frac_of_fp64_peak'],
                                   'determinants': int(timings.get('gw_dets', 0)) if dom == 'k_gw' else None}},
            'kernels': per_kernel,
            'wall_ms': {k[5:]: round(float(np.mean(v)), 2) for k, v in timings.items() if k.startswith('wall_')},
            'gw_stream_ms': round(avg('gw_stream_ms'), 2),
            'device_ms_per_step': round(sum(v[0] for v in kernels.values()) + avg('cluster_prep') + avg('reduce_sets'), 3),
            'verified': verified,
        }
        if ranks_info:
            res['gather_ms'] = ranks_info['gather_ms']           # rank 0, per step (incl. waiting for the slowest rank)
            res['ranks'] = ranks_info
        if world == 1 and not args.no_extras and depth == 1:
            # the same job with two batches in flight: a second context on a stream of its own, a
            # host thread per context; the kernels of consecutive batches share the GPU (the tail of
            # one launch runs beside the head of the next), so this is a throughput figure
            # only -- per-kernel durations under it are not those of a kernel alone
            side = torch.cuda.Stream(device=dev)     # (non-blocking: independent of the default stream)
            two = [ctx, hipabi.Context(local, side.cuda_stream)]
            try:
                compute(1, None, two)                # sizes the second context's scratch
                torch.cuda.synchronize()
                k2 = max(4, args.steps)
                t2 = time.perf_counter()
                out2 = run_steps(k2, {}, two)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t2
            finally:
                two[1].close()
            if rows_digest(out2) != digest0:
                raise SystemExit('bench.py: the two-in-flight run produced different rows')
            res['two_batches_in_flight'] = {'value': args.files * (args.seconds / 3600.0) * k2 / dt2,
                                            'unit': 'hours-audio/s', 'steps': k2,
                                            'ms_per_step': round(1e3 * dt2 / k2, 2)}
        if world == 1 and not args.no_extras and depth == 1:
            res['value_incl_h2d'] = with_uploads(torch, dev, args, frames, total, files, pipeline, ctx, cl, fused, digest0)
            res['dropin_scripts_wall_s'] = dropin_wall(args, synth, first_host)
            res['host_driven_modes_1h'] = host_driven_modes(args, synth, first_host, local)
        if world == 1 and not args.no_extras:
            res['hbm_copy_GBps'] = round(copy_rate(torch, dev), 1)
            res['other_configs'] = other_configs(torch, dev, hipabi, pipeline, synth_device, vad_times, ctx, local)
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(args, synth, cli)
            res['cpu_baseline_stats'] = cpu_baseline_stats(args, synth, cli)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def long_file_main(args, rank, local, world):
    """One long file over the ranks (BASELINE config 5): a step is the
    whole file from resident frames to labelled rows on every rank."""
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
    synth_device = importlib.import_module(PKG + '.synth_device')
    rec = importlib.import_module(PKG + '.recipe')
    dmod = importlib.import_module(PKG + '.distributed')
    seconds = args.seconds if args.seconds != 3600.0 else 36000.0
    speakers = args.speakers if args.speakers != 4 else 8
    feats, vad, _ = synth_device.make_session_device(424242, seconds, speakers, device=dev)
    T = int(feats.shape[0])
    vad_t = [(float(rec.py2_float_str(s / 125.0)), float(rec.py2_float_str(e / 125.0))) for (s, e) in vad]
    lo, hi = dmod.shard_turns(vad_t, world)[rank]
    f0 = 0 if lo == 0 else min(T, int(vad_t[lo - 1][1] * 125.0))
    f1 = T if hi == len(vad_t) else min(T, int(vad_t[hi - 1][1] * 125.0))
    shard = feats[f0:f1].clone()                      # this rank keeps ITS time shard only
    del feats
    torch.cuda.synchronize()
    ctx = hipabi.Context(local, torch.cuda.current_stream().cuda_stream)
    grp = dist if world > 1 else None

    def step(tm):
        return dmod.diarize_long_file(ctx, shard.data_ptr(), f0, f1 - f0, T, vad_t[lo:hi], grp, timings=tm)

    rows0 = step({})
    for _ in range(max(0, args.warmup - 1)):
        step({})
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    phases = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tm = {}
        rows = step(tm)
        phases.append(tm['long_file_ms'])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if not np.array_equal(rows, rows0):
        raise SystemExit('bench.py --long-file: a timed step produced different rows')
    keys = ['change_detection', 'segment_stats', 'gather_records', 'matrix_rows', 'gather_rows',
            'assemble_and_merge_loop', 'labels', 'total']
    mine = np.array([[np.mean([p[k] for p in phases]) for k in keys] + [dt]])
    if world > 1:
        t = torch.from_numpy(mine).to(dev if args.backend == 'nccl' else 'cpu')
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        every = np.concatenate([e.cpu().numpy() for e in every])
    else:
        every = mine
    dt = float(every[:, -1].max())
    if rank == 0:
        hours = (seconds / 3600.0) * args.steps
        res = {'metric': 'diarized audio throughput, ONE long file sharded in time over the GPUs (CD gw/BIC + CL hi/BIC)',
               'value': hours / dt, 'unit': 'hours-audio/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
               'dtype': 'f64', 'data': 'synthetic', 'xRT': hours / dt * 3600.0,
               'config': {'workload': 'one %.0f s / %d-speaker file (BASELINE config 5), %d frames, %d segments; frames '
                                      'resident in HBM, time-sharded between VAD turns; DIA2 flags'
                                      % (seconds, speakers, T, phases[-1]['segments']),
                          'parallelism': 'time shards x%d: all_gather of segment records, matrix row blocks, '
                                         'merge loop replicated' % world},
               # per phase: the slowest rank's mean over the steps (ms)
               'phases_ms_max_over_ranks': {k: round(float(every[:, i].max()), 3) for i, k in enumerate(keys)},
               'phases_ms_rank0': {k: round(float(every[0, i]), 3) for i, k in enumerate(keys)},
               'segments': phases[-1]['segments'], 'rows_of_rank0': phases[-1]['rows_of_this_rank'],
               'note': 'the merge loop is a serial chain replicated on every rank: only change_detection, '
                       'segment_stats and matrix_rows shrink with the rank count (DESIGN.md par. 6)'}
        print(json.dumps(res))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def with_uploads(torch, dev, args, frames, total, files, pipeline, ctx, cl, fused, digest0):
    """What the headline leaves out, measured: the same steps with every batch coming from
    PINNED HOST memory.  Two device buffers; the upload of batch k + 1 runs on a copy stream
    under the kernels of batch k (the library is synchronous to its caller, so the copy is
    queued before the step starts).  Reading the .fea files from disk is still not in it."""
    host = torch.empty(frames.shape, dtype=frames.dtype, pin_memory=True)
    host.copy_(frames)
    bufs = [frames, torch.empty_like(frames)]
    copier = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream()
    torch.cuda.synchronize()
    k = max(4, args.steps)

    def upload(i):
        ev = torch.cuda.Event()
        with torch.cuda.stream(copier):
            bufs[i % 2].copy_(host, non_blocking=True)
            ev.record(copier)
        return ev

    out = None
    t_up0 = time.perf_counter()
    ev = upload(0)
    ev.synchronize()
    t_up = time.perf_counter() - t_up0               # one upload alone
    t0 = time.perf_counter()
    ev = upload(0)
    for i in range(k):
        main.wait_event(ev)                          # batch i is on the device
        ev.synchronize()
        if i + 1 < k:
            nxt = upload(i + 1)                      # overwrites the buffer batch i - 1 used (its step has returned)
        rows = pipeline.diarize_batch(ctx, bufs[i % 2].data_ptr(), total, files, cl=cl, fused=fused)
        out = {j: r for j, r in enumerate(rows)}
        if i + 1 < k:
            ev = nxt
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rows_digest(out) != digest0:
        raise SystemExit('bench.py: the upload-inclusive run produced different rows')
    gb = host.numel() * 4 / 1e9
    del bufs[1], host
    return {'value': args.files * (args.seconds / 3600.0) * k / dt, 'unit': 'hours-audio/s', 'steps': k,
            'ms_per_step': round(1e3 * dt / k, 2), 'upload_GB_per_step': round(gb, 2),
            'upload_alone_ms': round(1e3 * t_up, 1), 'h2d_GBps': round(gb / t_up, 1),
            'note': 'frames from pinned host memory every step, upload of batch k+1 on a copy stream under the '
                    'kernels of batch k; .fea file reading excluded'}


def dropin_wall(args, synth, first_host):
    """Wall time of the drop-in executables themselves on ONE file, the way spk-diarization2.py
    runs them (spk-diarization2.py:122-128): two fresh child processes of this one (never an
    exec), each paying interpreter start, library load, context creation, .fea read and
    upload.  BASELINE.md: the reference needs ~3 811 s for the same two steps on 1 h."""
    import subprocess
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        _write_session(tmp, synth, first_host[0], first_host[1])
        env = dict(os.environ)
        t0 = time.perf_counter()
        r1 = subprocess.run([sys.executable, os.path.join(ROOT, 'spk-change-detection.py'), os.path.join(tmp, 'vad.recipe'),
                             os.path.join(tmp, 'fea') + '/', '-o', os.path.join(tmp, 'spkc.recipe')] + CD_ARGS,
                            cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        t1 = time.perf_counter()
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'spk-clustering.py'), os.path.join(tmp, 'spkc.recipe'),
                             os.path.join(tmp, 'fea') + '/', '-o', os.path.join(tmp, 'out.recipe')] + CL_ARGS,
                            cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        t2 = time.perf_counter()
        if r1.returncode or r2.returncode:
            return {'error': (r1.stderr + r2.stderr)[-400:]}
        lines = open(os.path.join(tmp, 'out.recipe')).read().count('\n')
    out = {'change_detection_s': round(t1 - t0, 3), 'clustering_s': round(t2 - t1, 3), 'total_s': round(t2 - t0, 3),
           'file_seconds': args.seconds, 'speaker_lines': lines,
           'xRT': round(args.seconds / (t2 - t0), 1),
           'note': './spk-change-detection.py + ./spk-clustering.py as child processes on one file '
                   '(process start, imports, library load, context, .fea read and upload included)'}
    return out


def host_driven_modes(args, synth, first_host, local):
    """The reference's host-driven modes on ONE 1 h file, in process, on an engine of their own:
    `-m m` (merge_rec, spk-change-detection.py:136-177: one pair distance and one decision per
    recipe line, each decision behind a synchronisation) over the growing-window output, with BIC
    and GLR, and `-m in` (spk_cluster_in, spk-clustering.py:136-175: every segment against every
    cluster so far).  Wall time of the library calls only (recipes in memory files, features
    uploaded before the clock starts is NOT possible here: the drivers load the .fea themselves,
    so the 70 MB read + upload is inside)."""
    import importlib
    import io
    import tempfile
    cli = importlib.import_module(PKG + '.cli')
    engine = importlib.import_module(PKG + '.engine')
    out = {}
    eng = engine.HipEngine(local)
    try:
        with tempfile.TemporaryDirectory() as tmp:
            _write_session(tmp, synth, first_host[0], first_host[1])
            fea = os.path.join(tmp, 'fea') + '/'
            spkc = os.path.join(tmp, 'spkc.recipe')
            cli.main_change_detection([os.path.join(tmp, 'vad.recipe'), fea, '-o', spkc] + CD_ARGS,
                                      engine=eng, stdout=io.StringIO())
            n_lines = open(spkc).read().count('\n')
            for tag, extra in (('merge_bic', ['-m', 'm', '-d', 'BIC', '-l', '1.3']),
                               ('merge_glr', ['-m', 'm', '-d', 'GLR', '-t', '1500'])):
                walls = []
                for it in range(3):
                    t0 = time.perf_counter()
                    cli.main_change_detection([spkc, fea, '-o', os.path.join(tmp, tag + '.recipe')] + extra,
                                              engine=eng, stdout=io.StringIO())
                    walls.append(time.perf_counter() - t0)
                kept = open(os.path.join(tmp, tag + '.recipe')).read().count('\n')
                out[tag] = {'ms': round(1e3 * float(np.median(walls)), 2), 'decisions': n_lines - 1, 'lines_out': kept,
                            'us_per_decision': round(1e6 * float(np.median(walls)) / max(n_lines - 1, 1), 1)}
            walls = []
            for it in range(3):
                t0 = time.perf_counter()
                cli.main_clustering([spkc, fea, '-o', os.path.join(tmp, 'in.recipe'), '-m', 'in', '-l', '1.3'],
                                    variant=1, engine=eng, stdout=io.StringIO())
                walls.append(time.perf_counter() - t0)
            spk = len(set(re.findall(r'speaker=(\S+)', open(os.path.join(tmp, 'in.recipe')).read())))
            out['cluster_in_bic'] = {'ms': round(1e3 * float(np.median(walls)), 2), 'segments': n_lines, 'speakers': spk,
                                     'us_per_segment': round(1e6 * float(np.median(walls)) / max(n_lines, 1), 1)}
    except Exception as e:                               # a measurement beside the line, never the line itself
        out['error'] = repr(e)[-300:]
    finally:
        eng.close()
    out['note'] = ('in-process runs of the drop-in drivers on a 1 h file (.fea read + upload inside): merge mode takes the '
                   'terms of all adjacent pairs in one call and goes back to the device only behind a merge; '
                   'spk_cluster_in is one device-resident chain (spkd_cluster_in), replayed on the host')
    return out


def copy_rate(torch, dev):
    """Device-to-device copy of 2 GiB: bytes read + bytes written per second."""
    n = 1 << 29
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 5 * 2 * a.numel() * 4 / 1e9 / (e0.elapsed_time(e1) / 1e3)


def other_configs(torch, dev, hipabi, pipeline, synth_device, vad_times, ctx, local):
    """BASELINE.json configs 2, 3 and 5 as single-file calls on the same context:
    wall latency of CD + CL per file (median of 5), with the kernel shares."""
    out = {}
    for name, secs, spk in (('config2_1h_4spk', 3600.0, 4), ('config3_1h_8spk', 3600.0, 8),
                            ('config5_10h_8spk', 36000.0, 8)):
        feats, vad, _ = synth_device.make_session_device(424242, secs, spk, device=dev)
        f = [pipeline.BatchFile(0, int(feats.shape[0]), vad_times(vad))]
        torch.cuda.synchronize()
        walls, tm = [], {}
        rows = None
        for it in range(6):
            tm = {}
            t0 = time.perf_counter()
            rows = pipeline.diarize_batch(ctx, feats.data_ptr(), int(feats.shape[0]), f, timings=tm, fused=True)
            walls.append(1e3 * (time.perf_counter() - t0))
        g = lambda k: round(float(np.mean(tm[k])), 3) if tm.get(k) else 0.0
        out[name] = {'ms': round(float(np.median(walls[1:])), 3), 'segments': int(len(rows[0])),
                     'speakers_found': int(rows[0][:, 2].max()) if len(rows[0]) else 0,
                     'xRT': round(secs / (float(np.median(walls[1:])) / 1e3), 0),
                     'k_gw_ms': g('gw'), 'k_matrix_ms': g('matrix'), 'ahc_ms': g('ahc'),
                     'wall_ms': {k[5:]: round(float(np.mean(v)), 3) for k, v in tm.items() if k.startswith('wall_')}}
        if name == 'config2_1h_4spk':
            # the reference's DEFAULT stand-alone mode on the same file: -m sw -d GLR -w 5.0 -st 0.5
            # (spk-change-detection.py:436-447): distances of every sliding window of every VAD turn
            tb = np.array([int(s * 125.0) for (s, e) in f[0].vad], dtype=np.int64)
            te = np.array([int(e * 125.0) for (s, e) in f[0].vad], dtype=np.int64)
            p = hipabi.CdParams(hipabi.KINDS['GLR'], 0, 1.3, 0.0, float(np.floor(5.0 * 125.0)),
                                float(np.floor(0.5 * 125.0)), float(np.floor(125.0 * 0.05)), 125.0)
            sw_walls = []
            for it in range(4):
                t0 = time.perf_counter()
                _, off, d = ctx.sw(feats.data_ptr(), int(feats.shape[0]), tb, te, p)
                sw_walls.append(1e3 * (time.perf_counter() - t0))
            out['sw_glr_1h'] = {'ms': round(float(np.median(sw_walls[1:])), 3), 'windows': int(len(d)),
                                'kernels_ms': round(ctx.last_ms('sw'), 3),
                                'note': 'sliding-window GLR distances (-w 5.0 -st 0.5) of every VAD turn of the 1 h file: '
                                        'two statistics records per window from the frames + the pair kernels'}
        del feats
    return out


if __name__ == '__main__':
    main()
