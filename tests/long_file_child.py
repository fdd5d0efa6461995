"""Worker of tests/test_long_file.py: one rank of the sharded long-file pipeline
(speaker-diarization_amd/distributed.py).  Started by
torch.distributed.run (or alone: world 1).  Every rank builds the same synthetic
session, keeps only ITS time shard on the GPU, and rank 0 writes rows + merge log."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    seed, seconds, nspk, out = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    share = '--share-device' in sys.argv
    backend = 'gloo' if '--gloo' in sys.argv else 'nccl'
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = 0 if share else int(os.environ.get('LOCAL_RANK', '0'))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group(backend)
    pkg = 'speaker-diarization_amd'
    hipabi = importlib.import_module(pkg + '.hipabi')
    synth = importlib.import_module(pkg + '.synth')
    rec = importlib.import_module(pkg + '.recipe')
    dmod = importlib.import_module(pkg + '.distributed')
    feats, vad, _ = synth.make_session(seed, seconds, nspk)
    vad_t = [(float(rec.py2_float_str(s / 125.0)), float(rec.py2_float_str(e / 125.0))) for (s, e) in vad]
    lo, hi = dmod.shard_turns(vad_t, world)[rank]
    mine = vad_t[lo:hi]
    T = feats.shape[0]
    # the shard: from the end of the previous rank's last turn to the end of this rank's last turn
    f0 = 0 if lo == 0 else min(T, int(vad_t[lo - 1][1] * 125.0))
    f1 = T if hi == len(vad_t) else min(T, int(vad_t[hi - 1][1] * 125.0))
    shard = torch.from_numpy(np.ascontiguousarray(feats[f0:f1])).cuda()
    ctx = hipabi.Context(local, torch.cuda.current_stream().cuda_stream)
    tm = {}
    rows, merges = dmod.diarize_long_file(ctx, shard.data_ptr(), f0, f1 - f0, T, mine, dist if world > 1 else None,
                                          timings=tm, want_merges=True)
    if rank == 0:
        np.savez(out, rows=rows, merges=np.array(merges, dtype=np.float64).reshape(-1, 3),
                 stat=np.array([tm['long_file_ms']['stat_max'], tm['long_file_ms']['stat_min']]))
        print('long_file', world, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in tm['long_file_ms'].items()})
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
