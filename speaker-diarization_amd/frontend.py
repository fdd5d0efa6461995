"""feacat-shaped feature extraction on the MI355X: the step of
spk-diarization2.py:98-100, `feacat -c fconfig.cfg -H --raw-output x.wav > fea/x.fea`.
The reference's own configuration file is read (feaconfig.py); the arithmetic runs in
libspkd_hip.so (spkd_mfcc).  PARITY UNPINNED -- feacat itself is not available; every
choice the configuration file leaves open is listed in the test suite's numpy restatement
(mfcc_numpy.py in the checker directory) and in include/spkd.h.

`main` mirrors the one feacat command line the reference uses: -c CONFIG -H --raw-output WAV,
feature file (int32 dim + float32 frames, spk-change-detection.py:37-41) on stdout.
"""
import argparse
import sys
import wave

import numpy as np

from . import hipabi
from .feaconfig import FeatureConfig

N_FFT, N_MEL = 512, 21


def mel_filterbank(sample_rate, n_fft=N_FFT, n_mel=N_MEL):
    """Triangular filters equally spaced on the mel scale 2595 log10(1 + f / 700), 0 .. Nyquist."""
    hz2mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    mel2hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    edges = mel2hz(np.linspace(hz2mel(0.0), hz2mel(sample_rate / 2.0), n_mel + 2))
    freqs = np.arange(n_fft // 2 + 1) * (sample_rate / float(n_fft))
    fb = np.zeros((n_mel, n_fft // 2 + 1))
    for m in range(n_mel):
        lo, mid, hi = edges[m], edges[m + 1], edges[m + 2]
        fb[m] = np.maximum(0.0, np.minimum((freqs - lo) / (mid - lo), (hi - freqs) / (hi - mid)))
    return fb.astype(np.float32)


def dct_matrix(n_cep, n_mel=N_MEL):
    k = np.arange(1, n_cep + 1)[:, None]
    m = np.arange(n_mel)[None, :]
    return (np.sqrt(2.0 / n_mel) * np.cos(np.pi * k * (m + 0.5) / n_mel)).astype(np.float32)


def read_wav(path):
    """16-bit mono PCM samples of a .wav file (what `ffmpeg -ar 16000 -ac 1` leaves,
    spk-diarization2.py:83-84) and its sample rate."""
    with wave.open(path, 'rb') as w:
        if w.getsampwidth() != 2 or w.getnchannels() != 1:
            raise ValueError('%s: 16-bit mono PCM expected' % path)
        return np.frombuffer(w.readframes(w.getnframes()), dtype='<i2'), w.getframerate()


def extract(pcm, cfg, ctx=None, device=0):
    """int16 samples -> float32 [T, 39] features (device computation, result on the host)."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    own = ctx is None
    if own:
        ctx = hipabi.Context(device)
    try:
        T = len(pcm) // cfg.hop
        out = np.zeros((T, cfg.dim), dtype=np.float32)
        if T == 0:
            return out
        d_pcm = ctx.dev_alloc(max(pcm.nbytes, 16))
        d_out = ctx.dev_alloc(max(out.nbytes, 16))
        try:
            ctx.h2d(d_pcm, pcm)
            p = hipabi.MfccParams(cfg.sample_rate, cfg.frame_rate, cfg.window_width, N_FFT, N_MEL, cfg.n_cep,
                                  cfg.cms_left, cfg.cms_right, (hipabi.C.c_int32 * 2)(*cfg.delta_width),
                                  cfg.pre_emph, (hipabi.C.c_float * 2)(*cfg.delta_norm))
            n = ctx.mfcc(d_pcm, len(pcm), p, mel_filterbank(cfg.sample_rate), dct_matrix(cfg.n_cep), cfg.mean,
                         cfg.scale, cfg.transform, d_out)
            assert n == T
            ctx.d2h(out, d_out)
        finally:
            ctx.dev_free(d_pcm)
            ctx.dev_free(d_out)
        return out
    finally:
        if own:
            ctx.close()


def main(argv=None, stdout=None):
    ap = argparse.ArgumentParser(description='feacat-shaped feature extraction (the options spk-diarization2.py uses).')
    ap.add_argument('-c', dest='config', required=True, help='feature configuration (fconfig.cfg)')
    ap.add_argument('-H', dest='header', action='store_true', help='write the int32 dimension header')
    ap.add_argument('--raw-output', dest='raw', action='store_true', help='raw float32 frames')
    ap.add_argument('wav')
    args = ap.parse_args(argv)
    cfg = FeatureConfig.load(args.config)
    pcm, rate = read_wav(args.wav)
    if rate != cfg.sample_rate:
        raise ValueError('%s is sampled at %d Hz, the configuration wants %d' % (args.wav, rate, cfg.sample_rate))
    feats = extract(pcm, cfg)
    out = stdout or sys.stdout.buffer
    if args.header:
        out.write(np.array([feats.shape[1]], dtype='<i4').tobytes())
    out.write(np.ascontiguousarray(feats, dtype='<f4').tobytes())
    return 0
