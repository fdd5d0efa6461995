"""CPU restatement (numpy, float64 inside, float32 out) of the feacat-shaped MFCC
front-end of speaker-diarization_amd/frontend.py -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: feacat (AaltoASR, C++) is neither in the reference tree nor installable
here, and the reference holds no feature file, so nothing pins these numbers.  What the
reference DOES fix is its configuration file (fconfig.cfg:1-101): every parameter it
states is honoured; where the file is silent the standard definitions are used, and each
such choice is listed here so that a maintainer with feacat at hand can correct it:

  audiofile   16 kHz, 125 frames/s (hop 128), window 400 samples, pre-emphasis 0.97
              (fconfig.cfg:5-8).  CHOICE: frame t covers samples [128 t - 200, 128 t + 200)
              (centred), samples outside the file are the border sample (copy_borders 1),
              y[n] = x[n] - 0.97 x[n-1], Hamming window, N = T_samples // 128 frames.
  fft         magnitude spectrum (magnitude 1).  CHOICE: 512-point transform (window
              zero-padded), 257 bins.
  mel         CHOICE: 21 triangular filters, equally spaced on the mel scale
              2595 log10(1 + f / 700) between 0 and 8 kHz, unit peak; log of the summed
              magnitudes (floor 1e-10).
  power       CHOICE: log of the sum of the squared magnitude spectrum (floor 1e-10).
  dct         dim 12, zeroth 0 (fconfig.cfg:35-41): c_k = sqrt(2/M) sum_m L_m cos(pi k (m + 1/2) / M),
              k = 1..12.
  cms         mean over the frames [t - 75, t + 75] that exist (fconfig.cfg:51-58).
  delta       width 2 (fconfig.cfg:60-76): d[t] = sum_{k=1..2} k (x[t+k] - x[t-k]) / normalization,
              border frames repeated; normalization 1 for the deltas, 10 for the delta-deltas
              AS THE FILE STATES.
  merge, normalization (x - mean) * scale, lin_transform y = M x (fconfig.cfg:78-100).
"""
import numpy as np

N_FFT = 512
N_MEL = 21


def mel_filterbank(sample_rate, n_fft=N_FFT, n_mel=N_MEL):
    hz2mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    mel2hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    edges = mel2hz(np.linspace(hz2mel(0.0), hz2mel(sample_rate / 2.0), n_mel + 2))
    freqs = np.arange(n_fft // 2 + 1) * (sample_rate / float(n_fft))
    fb = np.zeros((n_mel, n_fft // 2 + 1))
    for m in range(n_mel):
        lo, mid, hi = edges[m], edges[m + 1], edges[m + 2]
        up = (freqs - lo) / (mid - lo)
        down = (hi - freqs) / (hi - mid)
        fb[m] = np.maximum(0.0, np.minimum(up, down))
    return fb


def dct_matrix(n_cep, n_mel=N_MEL):
    k = np.arange(1, n_cep + 1)[:, None]
    m = np.arange(n_mel)[None, :]
    return np.sqrt(2.0 / n_mel) * np.cos(np.pi * k * (m + 0.5) / n_mel)


def static_features(pcm, cfg):
    """int16 / float samples -> [T, n_cep + 1] (cepstra 1..n_cep, log power), float64."""
    x = np.asarray(pcm, dtype=np.float64)
    n_frames = len(x) // cfg.hop
    half = cfg.window_width // 2
    idx = np.arange(n_frames)[:, None] * cfg.hop - half + np.arange(cfg.window_width)[None, :]
    cur = x[np.clip(idx, 0, len(x) - 1)]
    prev = x[np.clip(idx - 1, 0, len(x) - 1)]
    y = (cur - cfg.pre_emph * prev) * np.hamming(cfg.window_width)[None, :]
    mag = np.abs(np.fft.rfft(y, n=N_FFT, axis=1))
    logmel = np.log(np.maximum(mag @ mel_filterbank(cfg.sample_rate).T, 1e-10))
    cep = logmel @ dct_matrix(cfg.n_cep).T
    power = np.log(np.maximum((mag ** 2).sum(axis=1), 1e-10))
    return np.concatenate([cep, power[:, None]], axis=1)


def _delta(x, width, norm):
    T = x.shape[0]
    out = np.zeros_like(x)
    t = np.arange(T)
    for k in range(1, width + 1):
        out += k * (x[np.minimum(t + k, T - 1)] - x[np.maximum(t - k, 0)])
    return out / norm


def features(pcm, cfg):
    """The whole chain: float32 [T, 39]."""
    s = static_features(pcm, cfg)
    T = s.shape[0]
    if T == 0:
        return np.zeros((0, cfg.dim), dtype=np.float32)
    c = np.concatenate([np.zeros((1, s.shape[1])), np.cumsum(s, axis=0)])
    lo = np.maximum(np.arange(T) - cfg.cms_left, 0)
    hi = np.minimum(np.arange(T) + cfg.cms_right + 1, T)
    cms = s - (c[hi] - c[lo]) / (hi - lo)[:, None]
    d1 = _delta(cms, cfg.delta_width[0], cfg.delta_norm[0])
    d2 = _delta(d1, cfg.delta_width[1], cfg.delta_norm[1])
    z = (np.concatenate([cms, d1, d2], axis=1) - cfg.mean[None, :]) * cfg.scale[None, :]
    return (z @ cfg.transform.astype(np.float64).T).astype(np.float32)
