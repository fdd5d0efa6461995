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
