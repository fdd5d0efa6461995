"""The synthetic feature generator of synth.py with the samples produced on the GPU
(torch tensors), bit for bit the same bytes: integer hashing in two's-complement int64
(wrap-around products, logical shifts by masking), exactly representable fp64 scalings,
and the same order of the separately rounded fp64 multiply and add per output element.
``fea_sha256`` of the result equals that of synth.make_session (tests assert it).

Used where many long sessions are wanted quickly and already resident in HBM: bench.py
builds its batch of distinct one-hour files with it in a few seconds instead of three
host-seconds per file.  torch is needed only here, never by the scripts.
"""
import numpy as np

from . import synth

_MASK = {k: (1 << (64 - k)) - 1 for k in (11, 16, 27, 30, 31, 32, 48)}


def _i64(x):
    """Python int (any uint64 value) -> the int64 with the same bits."""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


_M1, _M2, _G, _S = (_i64(int(v)) for v in (synth._M1, synth._M2, synth._G, synth._S))


def _lsr(x, k):
    return (x >> k) & _MASK[k]


def _mix(x):
    x = (x ^ _lsr(x, 30)) * _M1
    x = (x ^ _lsr(x, 27)) * _M2
    return x ^ _lsr(x, 31)


def _base(seed, stream):
    """synth._raw's scalar part, on the host in exact integer arithmetic."""
    m = (1 << 64) - 1
    x = (int(seed) + int(synth._S) * int(stream)) & m
    x = ((x ^ (x >> 30)) * int(synth._M1)) & m
    x = ((x ^ (x >> 27)) * int(synth._M2)) & m
    return _i64(x ^ (x >> 31))


def _gauss_pieces(torch, seed, streams, lengths, device):
    """synth._gauss(seed, stream_i, lengths_i) for every piece i, concatenated: element e
    of piece i is hashed with its index inside the piece and the piece's own stream."""
    n = torch.as_tensor(lengths, dtype=torch.int64, device=device)
    piece = torch.repeat_interleave(torch.arange(len(lengths), device=device), n)
    first = torch.cumsum(n, 0) - n
    idx = torch.arange(int(n.sum().item()), dtype=torch.int64, device=device) - first[piece]
    acc = torch.zeros_like(idx)
    for rep in range(3):
        base = torch.tensor([_base(seed, st * 4 + rep + 1000003) for st in streams],
                            dtype=torch.int64, device=device)[piece]
        r = _mix(_mix(idx * _G + base) + _G)
        acc += r & 0xFFFF
        for sh in (16, 32, 48):
            acc += (r >> sh) & 0xFFFF
    return (acc - 393210).to(torch.float64) / 65536.0


def make_session_device(seed, seconds, n_speakers=4, device='cuda', **kw):
    """Returns (features: float32 tensor [T, 39] on `device`, vad_turns, truth), equal
    to synth.make_session(seed, seconds, n_speakers, **kw).  A handful of large kernels
    per session (all pieces hashed at once, one 39-step accumulation per speaker)."""
    import torch
    total, pieces, vad, truth = synth.plan_session(seed, seconds, n_speakers, **kw)
    dim = synth.DIM
    z = _gauss_pieces(torch, seed, [p[2] for p in pieces], [p[1] * dim for p in pieces], device).reshape(total, dim)
    spk = torch.repeat_interleave(torch.tensor([p[3] for p in pieces], device=device),
                                  torch.tensor([p[1] for p in pieces], device=device))
    feats = torch.empty((total, dim), dtype=torch.float32, device=device)
    rows = torch.nonzero(spk < 0).squeeze(1)
    feats[rows] = (0.05 * z[rows]).to(torch.float32)
    for k in range(n_speakers):
        rows = torch.nonzero(spk == k).squeeze(1)
        if rows.numel() == 0:
            continue
        mu, a = synth._speaker_model(seed, k)
        mu = torch.from_numpy(np.ascontiguousarray(mu)).to(device)
        a = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        zk = z[rows]
        out = torch.zeros((rows.numel(), dim), dtype=torch.float64, device=device)
        for j in range(dim):
            # one rounded product, then one rounded sum, like the numpy original
            out += zk[:, j:j + 1] * a[:, j][None, :]
        out += mu[None, :]
        feats[rows] = out.to(torch.float32)
    return feats, vad, truth
