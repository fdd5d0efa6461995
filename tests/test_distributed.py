"""world_size-2 gloo test (CPU) of the file sharding + recipe gather used by the
multi-GPU path.  The per-file work is a stand-in here (the sharding and the
exchange are what is under test); on the GPU box bench.py runs the same code
over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_files, q):
    import importlib
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dmod = importlib.import_module('speaker-diarization_amd.distributed')
    files = ['file%03d.wav' % i for i in range(n_files)]
    mine = dmod.shard(files, rank, world)
    local = []
    for i, name in mine:
        rows = [(1.0 + k, 2.5 + k, (i + k) % 3 + 1) for k in range(i % 5 + (0 if i == 3 else 1))]
        local.append((i, dmod.recipe_text(name, rows)))
    got = dmod.gather_texts(local, dist)
    if rank == 0:
        q.put(got)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_files', [0, 1, 7])
def test_shard_and_gather_two_ranks(n_files):
    import importlib
    dmod = importlib.import_module('speaker-diarization_amd.distributed')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_files, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(got) == list(range(n_files))
    for i in range(n_files):
        rows = [(1.0 + k, 2.5 + k, (i + k) % 3 + 1) for k in range(i % 5 + (0 if i == 3 else 1))]
        assert got[i] == dmod.recipe_text('file%03d.wav' % i, rows)


def test_single_process_gather_is_identity():
    import importlib
    dmod = importlib.import_module('speaker-diarization_amd.distributed')
    assert dmod.gather_texts([(2, 'x\n'), (0, '')]) == {2: 'x\n', 0: ''}
    assert dmod.shard(list('abcde'), 1, 2) == [(1, 'b'), (3, 'd')]


@pytest.mark.gpu
@pytest.mark.parametrize('launcher', ['self', 'torchrun'])
def test_bench_two_ranks_gather_equals_one_rank(tmp_path, launcher):
    """The multi-GPU path of bench.py, driver-style: two fresh child processes, one rank
    each, sharing the box's single GPU (gloo instead of RCCL for that reason), started
    either by `python bench.py --gpus 2` ITSELF (the parent spawns torch.distributed.run
    before anything touches the GPU) or by torch.distributed.run from outside.  Rank 0
    must have gathered world x files recipes, identical to what one rank produces for
    the same global file indices, and the line carries the gather time and the ranks'
    step times."""
    import subprocess
    import numpy as np
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    common = ['--steps', '1', '--warmup', '0', '--seconds', '120', '--no-cpu-baseline', '--no-extras']
    two = os.path.join(str(tmp_path), 'two.npz')
    one = os.path.join(str(tmp_path), 'one.npz')
    head = [sys.executable, 'bench.py'] if launcher == 'self' else [
        sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), 'bench.py']
    env.pop('WORLD_SIZE', None)
    r2 = subprocess.run(head + ['--gpus', '2', '--backend', 'gloo', '--share-device', '--files', '2',
                                '--dump-rows', two] + common, cwd=ROOT, env=env, capture_output=True, text=True,
                        timeout=600)
    assert r2.returncode == 0, r2.stderr[-3000:]
    r1 = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--files', '4', '--dump-rows', one] + common,
                        cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-3000:]
    import json
    line = json.loads(r2.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['verified']['files_gathered'] == 4
    assert line['gather_ms'] >= 0.0 and line['ranks']['ms_per_step_max'] >= line['ranks']['ms_per_step_min'] > 0.0
    a, b = np.load(two), np.load(one)
    assert sorted(a.files) == sorted(b.files) == ['f0', 'f1', 'f2', 'f3']
    for k in a.files:
        assert np.array_equal(a[k], b[k]) and len(a[k]) > 3


def test_bench_gpus_flag_starts_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start two ranks
    by itself (as child processes, before it touches the GPU).  Without a GPU every rank
    stops at its own device check: the ranks' message and torchrun's failure report for
    two local ranks are the evidence that the spawn happened."""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip('covered by test_bench_two_ranks_gather_equals_one_rank[self] on a GPU box')
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--backend', 'gloo', '--share-device',
                        '--files', '1', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-extras'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    # (the launcher stops the other rank as soon as the first one has failed: one message is
    # certain, and its failure report names both local ranks)
    assert r.stderr.count('bench.py needs an MI355X') >= 1, r.stderr[-2000:]
    assert 'local_rank: 0' in r.stderr and 'local_rank: 1' in r.stderr, r.stderr[-2000:]
