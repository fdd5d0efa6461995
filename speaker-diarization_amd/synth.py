"""Bit-reproducible synthetic feature files (feacat ``.fea`` layout) plus the VAD
recipe that goes with them.

There is no feacat / audio in the build environment, so every workload enters the
hot path at the feature level: d = 39, 125 frames/s, float32.
The generator uses only integer hashing and exactly-rounded IEEE add/mul in a
fixed order (no libm, no BLAS), so the same seed gives the same bytes on every
machine; ``fea_sha256`` lets tests assert that.

Statistical model: K speakers; speaker k emits frames  A_k z + mu_k  with
z ~ approx N(0, I) (Irwin-Hall, 12 uniforms), mu_k ~ 0.5 N(0, I),
A_k = diag(U(0.6, 1.4)) (I + 0.15 G).  Speaker turns last U(3 s, 15 s) and never
repeat back to back; VAD turns are runs of 2..6 speaker turns separated by 2 s of
low-variance "silence" frames that are in the file but not in the recipe.
"""
import hashlib
import numpy as np

DIM = 39
RATE = 125

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)
_S = np.uint64(0xD1B54A32D192ED03)


def _mix(x):
    with np.errstate(over='ignore'):
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


def _raw(seed, stream, idx):
    """Counter-based 64-bit hash: (seed, stream, idx) -> uint64 array."""
    with np.errstate(over='ignore'):
        base = _mix(np.uint64(seed) + _S * np.uint64(stream))
        return _mix(_mix(base + _G * idx.astype(np.uint64)) + _G)


def _uniform(seed, stream, n):
    """n doubles in [0, 1) with 53 random bits."""
    r = _raw(seed, stream, np.arange(n, dtype=np.uint64))
    return (r >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _gauss(seed, stream, n):
    """n approx-normal doubles: sum of twelve 16-bit uniforms, centred, scaled by
    2**-16 (all exact in binary64)."""
    idx = np.arange(n, dtype=np.uint64)
    acc = np.zeros(n, dtype=np.int64)
    for rep in range(3):
        r = _raw(seed, stream * 4 + rep + 1000003, idx)
        for sh in (0, 16, 32, 48):
            acc += ((r >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
    return (acc - 393210).astype(np.float64) / 65536.0


def _speaker_model(seed, k):
    mu = 0.5 * _gauss(seed, 10 + 3 * k, DIM)
    scale = 0.6 + 0.8 * _uniform(seed, 11 + 3 * k, DIM)
    g = _gauss(seed, 12 + 3 * k, DIM * DIM).reshape(DIM, DIM)
    a = scale[:, None] * (np.eye(DIM) + 0.15 * g)
    return mu, a


def _emit(seed, stream, n, mu, a):
    """n frames of A z + mu, accumulated column by column (fixed order)."""
    z = _gauss(seed, stream, n * DIM).reshape(n, DIM)
    out = np.zeros((n, DIM), dtype=np.float64)
    for j in range(DIM):
        out += z[:, j:j + 1] * a[:, j][None, :]
    out += mu[None, :]
    return out.astype(np.float32)


def plan_session(seed, seconds, n_speakers=4, min_turn=3.0, max_turn=15.0,
                 sil_seconds=2.0, group=(2, 6), lead_silence=1.0):
    """The layout of a session without its samples: (total_frames, pieces, vad_turns,
    truth) with pieces = [(pos, length, stream, speaker or -1 for silence)] in file order;
    vad_turns = [(start_frame, end_frame)] and truth = [(start, end, speaker)]."""
    total = int(round(seconds * RATE))
    u = _uniform(seed, 1, 4 * (total // (int(min_turn * RATE)) + 16) + 64)
    ui = 0
    pieces, vad, truth = [], [], []
    pos = 0
    stream = 100
    prev = -1

    def silence(n):
        nonlocal pos, stream
        n = min(n, total - pos)
        if n <= 0:
            return
        pieces.append((pos, n, stream, -1))
        stream += 1
        pos += n

    silence(int(lead_silence * RATE))
    while pos < total:
        nturns = group[0] + int(u[ui] * (group[1] - group[0] + 1)); ui += 1
        vstart = pos
        for _ in range(nturns):
            length = int((min_turn + u[ui] * (max_turn - min_turn)) * RATE); ui += 1
            length = min(length, total - pos)
            if length < int(min_turn * RATE):
                # not enough room for a real turn: pad with silence and stop
                break
            if n_speakers > 1:
                k = int(u[ui] * (n_speakers - 1)); ui += 1
                if prev >= 0 and k >= prev:
                    k += 1
            else:
                k = 0
            pieces.append((pos, length, stream, k))
            stream += 1
            truth.append((pos, pos + length, k))
            pos += length
            prev = k
        if pos > vstart:
            vad.append((vstart, pos))
        if pos < total and total - pos < int(min_turn * RATE) + int(sil_seconds * RATE):
            silence(total - pos)
        else:
            silence(int(sil_seconds * RATE))
    return total, pieces, vad, truth


def make_session(seed, seconds, n_speakers=4, min_turn=3.0, max_turn=15.0,
                 sil_seconds=2.0, group=(2, 6), lead_silence=1.0):
    """Returns (features float32 [T, 39], vad_turns, truth) where
    vad_turns = [(start_frame, end_frame)] and truth = [(start, end, speaker)]."""
    total, pieces, vad, truth = plan_session(seed, seconds, n_speakers, min_turn, max_turn,
                                             sil_seconds, group, lead_silence)
    models = [_speaker_model(seed, k) for k in range(n_speakers)]
    feats = np.empty((total, DIM), dtype=np.float32)
    for (pos, n, stream, k) in pieces:
        if k < 0:
            z = _gauss(seed, stream, n * DIM).reshape(n, DIM)
            feats[pos:pos + n] = (0.05 * z).astype(np.float32)
        else:
            mu, a = models[k]
            feats[pos:pos + n] = _emit(seed, stream, n, mu, a)
    return feats, vad, truth


def write_fea(path, feats):
    """feacat ``-H --raw-output`` layout: int32 dim, then float32 frames, little
    endian (spk-change-detection.py:37-41 reads exactly this)."""
    feats = np.ascontiguousarray(feats, dtype='<f4')
    with open(path, 'wb') as f:
        np.array([feats.shape[1]], dtype='<i4').tofile(f)
        feats.tofile(f)


def fea_sha256(feats):
    h = hashlib.sha256()
    h.update(np.array([feats.shape[1]], dtype='<i4').tobytes())
    h.update(np.ascontiguousarray(feats, dtype='<f4').tobytes())
    return h.hexdigest()


def vad_recipe_text(audio, vad, rate=RATE):
    """VAD-stage recipe (voice-detection2.py:117-122 grammar): lna names a_1.."""
    from .recipe import py2_float_str
    lines = []
    for n, (s, e) in enumerate(vad):
        lines.append('audio=%s lna=a_%d start-time=%s end-time=%s\n' % (
            audio, n + 1, py2_float_str(s / float(rate)), py2_float_str(e / float(rate))))
    return ''.join(lines)
