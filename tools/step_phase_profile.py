#!/usr/bin/env python3
"""Where a round of the hand-off-free merge chain (k_ahc_step) spends its cycles: one 1 h / 4-speaker
file through the profiling build (make -C speaker-diarization_amd/csrc libspkd_hip_prof.so);
clocks of thread 0 of workgroup 0, summed over the rounds.  Development tool only."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'speaker-diarization_amd'


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3600.0
    import torch
    hipabi = importlib.import_module(PKG + '.hipabi')
    lib = hipabi.load_library(os.path.join(ROOT, PKG, 'csrc', 'libspkd_hip_prof.so'))
    hipabi._lib = lib
    sd = importlib.import_module(PKG + '.synth_device')
    pipeline = importlib.import_module(PKG + '.pipeline')
    rec = importlib.import_module(PKG + '.recipe')
    feats, vad, _ = sd.make_session_device(424242, secs, 4 if secs <= 3600 else 8, device='cuda')
    vt = [(float(rec.py2_float_str(s / 125.0)), float(rec.py2_float_str(e / 125.0))) for (s, e) in vad]
    f = [pipeline.BatchFile(0, int(feats.shape[0]), vt)]
    ctx = hipabi.Context(0, torch.cuda.current_stream().cuda_stream)
    a8 = (C.c_ulonglong * 8)()
    f8 = (C.c_ulonglong * 8)()
    for it in range(2):
        lib.spkd_debug_ahc_prof(a8)                    # (zeroes the counters)
        lib.spkd_debug_step_prof(f8)
        tm = {}
        pipeline.diarize_batch(ctx, feats.data_ptr(), int(feats.shape[0]), f, timings=tm, fused=True)
        lib.spkd_debug_ahc_prof(a8)
        v = list(a8)
        n = max(v[1], 1)
        print('merge chain %.2f ms, %d rounds, %.2f us each' % (tm['ahc'][-1], v[1], 1e3 * tm['ahc'][-1] / n))
        tot = float(sum(v[2:7]))
        for name, x in zip(('selection pass + reductions', 'partner list + merged record', 'bookkeeping (workgroup 0)',
                            'the pass (loads + elimination)', 'logs, distances, caches, rescans'), v[2:7]):
            print('  %-34s %5.1f %%  %8.0f cycles/round' % (name, 100.0 * x / tot, x / n))
        lib.spkd_debug_step_prof(f8)
        print('  inside the selection: state + offsets %.0f, the pass over the clusters %.0f, wave reduction %.0f, '
              'barrier + fold %.0f cycles/round' % tuple(x / n for x in list(f8)[:4]))


if __name__ == '__main__':
    main()
