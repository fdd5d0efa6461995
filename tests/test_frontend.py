"""feacat-shaped MFCC front-end.  PARITY UNPINNED: neither feacat
nor any feature file of the reference exists here, so these tests check (a) the reader of
the reference's configuration grammar against the parameters parsed from the reference's
own fconfig.cfg (tests/golden/feaconfig.json), (b) the numpy restatement against
properties any MFCC chain has, and (c) -- on the GPU -- the HIP kernels against that
restatement on the same samples."""
import io
import json
import os
import wave

import numpy as np
import pytest

from conftest import pkg
from helpers import ROOT

GOLD = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'feaconfig.json')))
REF_CFG = os.path.join(os.environ.get('SPKD_REFERENCE', '/root/reference'), 'fconfig.cfg')


def _arrays():
    """The trained arrays of the chain (normalization mean / scale, 39x39 transform).  The
    reference's own are model data and are not stored in this repository (only their SHA-256 is,
    tests/golden/feaconfig.json): synthetic ones of the same shapes and magnitudes stand in --
    nothing here pins the numbers anyway (PARITY UNPINNED)."""
    rng = np.random.default_rng(39)
    dim = GOLD['dim']
    mean = np.zeros(dim)
    scale = 0.15 + 0.05 * np.arange(dim) + 0.01 * rng.random(dim)
    transform = np.eye(dim) + 0.1 * rng.standard_normal((dim, dim))
    return mean, scale, transform.ravel()


def _cfg_text(g):
    """A configuration file in the feacat grammar with the golden parameters (written
    from scratch: module order and names as the chain needs them)."""
    nums = lambda v: ' '.join(repr(float(x)) for x in v)
    mean, scale, transform = _arrays()
    g = dict(g, mean=mean, scale=scale, transform=transform)
    return '\n'.join([
        'module\n{\n  name audiofile\n  type audiofile\n  pre_emph_coef %r\n  sample_rate %d\n  frame_rate %d\n'
        '  window_width %d\n  copy_borders %d\n  raw 1\n}' % (g['pre_emph'], g['sample_rate'], g['frame_rate'],
                                                              g['window_width'], g['copy_borders']),
        'module\n{\n  name fft\n  type fft\n  magnitude %d\n  sources audiofile\n}' % g['magnitude'],
        'module\n{\n  name mel\n  type mel\n  sources fft\n}',
        'module\n{\n  name power\n  type power\n  sources fft\n}',
        'module\n{\n  name mfcc\n  type dct\n  dim %d\n  zeroth %d\n  sources mel\n}' % (g['n_cep'], g['zeroth']),
        'module\n{\n  name mfcc_power\n  type merge\n  sources mfcc power\n}',
        'module\n{\n  name cms\n  type mean_subtractor\n  left %d\n  right %d\n  sources mfcc_power\n}' % (
            g['cms_left'], g['cms_right']),
        'module\n{\n  name delta1\n  type delta\n  width %d\n  normalization %r\n  sources cms\n}' % (
            g['delta_width'][0], g['delta_norm'][0]),
        'module\n{\n  name delta2\n  type delta\n  width %d\n  normalization %r\n  sources delta1\n}' % (
            g['delta_width'][1], g['delta_norm'][1]),
        'module\n{\n  name all\n  type merge\n  sources cms delta1 delta2\n}',
        'module\n{\n  name normalization\n  type normalization\n  mean %s\n  scale %s\n  sources all\n}' % (
            nums(g['mean']), nums(g['scale'])),
        'module\n{\n  name transform\n  type lin_transform\n  dim %d\n  matrix %s\n  sources normalization\n}' % (
            g['dim'], nums(g['transform'])),
    ]) + '\n'


@pytest.fixture(scope='module')
def cfg():
    return pkg('feaconfig').FeatureConfig(_cfg_text(GOLD))


def _signal(seconds=4.0, rate=16000, seed=3):
    rng = np.random.default_rng(seed)
    t = np.arange(int(seconds * rate)) / rate
    x = 3000 * np.sin(2 * np.pi * 440 * t) + 1500 * np.sin(2 * np.pi * 1870 * t + 1.0) * (t > 1.5)
    x += 300 * rng.standard_normal(len(t))
    return np.clip(x, -32768, 32767).astype(np.int16)


def test_config_reader_round_trips_the_reference_parameters(cfg):
    assert (cfg.sample_rate, cfg.frame_rate, cfg.window_width, cfg.hop) == (16000, 125, 400, 128)
    assert (cfg.n_cep, cfg.cms_left, cfg.cms_right, cfg.dim) == (12, 75, 75, 39)
    assert cfg.delta_width == [2, 2] and cfg.delta_norm == [1.0, 10.0] and cfg.pre_emph == pytest.approx(0.97)
    mean, scale, transform = _arrays()
    assert np.allclose(cfg.scale, scale) and np.allclose(cfg.transform.ravel(), transform) and np.allclose(cfg.mean, mean)
    with pytest.raises(ValueError):
        pkg('feaconfig').FeatureConfig('module\n{\n  name a\n  type audiofile\n  sample_rate 16000\n  frame_rate 125\n'
                                       '  window_width 400\n}\n')


def test_unsupported_configurations_are_refused_not_reinterpreted():
    """A configuration whose switches or module chain differ from the one the device code
    computes must raise, not silently yield the shipped chain's features."""
    fc = pkg('feaconfig')
    good = _cfg_text(GOLD)
    fc.FeatureConfig(good)
    for a, b in (('magnitude 1', 'magnitude 0'), ('zeroth 0', 'zeroth 1'), ('copy_borders 1', 'copy_borders 0'),
                 ('sources mfcc power', 'sources power mfcc'), ('sources cms delta1 delta2', 'sources delta1 cms delta2'),
                 ('sources delta1\n', 'sources cms\n'), ('window_width 400', 'window_width 320'),
                 ('sample_rate 16000', 'sample_rate 0')):
        assert a in good
        with pytest.raises((ValueError, ZeroDivisionError)):
            fc.FeatureConfig(good.replace(a, b))


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason='the reference tree is not on this machine')
def test_reference_configuration_file_parses_to_the_golden_parameters():
    import hashlib
    c = pkg('feaconfig').FeatureConfig.load(REF_CFG)
    for k in ('sample_rate', 'frame_rate', 'window_width', 'copy_borders', 'magnitude', 'n_cep', 'zeroth',
              'cms_left', 'cms_right', 'delta_width', 'delta_norm', 'dim'):
        assert getattr(c, k) == GOLD[k], k
    assert c.pre_emph == pytest.approx(GOLD['pre_emph'])
    arr = np.concatenate([c.mean, c.scale, c.transform.ravel()]).astype('<f4').tobytes()
    assert hashlib.sha256(arr).hexdigest() == GOLD['arrays_sha256']


def test_numpy_restatement_has_the_properties_of_an_mfcc_chain(cfg):
    from oracle import mfcc_numpy as m
    pcm = _signal()
    f = m.features(pcm, cfg)
    assert f.shape == (len(pcm) // 128, 39) and f.dtype == np.float32 and np.all(np.isfinite(f))
    # mean subtraction: a gain change (x2) only moves log power / c0-like terms, which the
    # +-75-frame mean removes except where the window is clipped -> features nearly equal
    g = m.features((pcm.astype(np.int32) // 2).astype(np.int16), cfg)
    assert np.max(np.abs(f[100:-100] - g[100:-100])) < 0.05
    # the static block is the cosine transform of the log mel spectrum: a pure tone's energy
    # sits in the filter that covers it
    s = m.static_features(_signal(1.0)[:16000], cfg)
    assert s.shape[1] == 13
    fb = m.mel_filterbank(16000)
    assert fb.shape == (21, 257) and np.all(fb >= 0) and np.all(fb.max(axis=1) > 0.5)
    assert np.argmax(fb[:, round(440 / (16000 / 512))]) == np.argmax(fb @ np.abs(np.fft.rfft(
        np.sin(2 * np.pi * 440 * np.arange(512) / 16000) * np.hamming(512))))
    # deltas of a constant are zero, of a ramp constant
    ramp = np.arange(50, dtype=np.float64)[:, None] * np.ones((1, 13))
    d = m._delta(ramp, 2, 1.0)
    assert np.allclose(d[5:-5], 10.0)          # sum k * 2k = 2 (1 + 4)


def test_feacat_stand_in_writes_the_fea_layout_on_cpu_path_free_host(tmp_path, cfg):
    """The reader of .wav files and the argument surface (no GPU: extraction is patched out)."""
    fe = pkg('frontend')
    p = os.path.join(str(tmp_path), 'a.wav')
    pcm = _signal(0.5)
    with wave.open(p, 'wb') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(pcm.tobytes())
    got, rate = fe.read_wav(p)
    assert rate == 16000 and np.array_equal(got, pcm)
    assert fe.mel_filterbank(16000).shape == (21, 257) and fe.dct_matrix(12).shape == (12, 21)


@pytest.mark.gpu
def test_hip_front_end_matches_the_numpy_restatement(tmp_path, cfg):
    from oracle import mfcc_numpy as m
    fe = pkg('frontend')
    for seconds, seed in ((4.0, 3), (0.9, 5), (21.3, 7)):
        pcm = _signal(seconds, seed=seed)
        want = m.features(pcm, cfg)
        got = fe.extract(pcm, cfg)
        assert got.shape == want.shape and np.all(np.isfinite(got))
        # fp32 DFT of 400 samples against float64 numpy: 1e-3 of the feature scale
        scale = max(1.0, float(np.abs(want).max()))
        assert float(np.max(np.abs(got - want))) < 2e-3 * scale, (seconds, float(np.max(np.abs(got - want))))
    assert fe.extract(np.zeros(100, dtype=np.int16), cfg).shape == (0, 39)
    # the feacat command line of spk-diarization2.py:98-100, feature file on stdout
    cfgp = os.path.join(str(tmp_path), 'fconfig.cfg')
    with open(cfgp, 'w') as f:
        f.write(_cfg_text(GOLD))
    wav = os.path.join(str(tmp_path), 'm.wav')
    pcm = _signal(2.0)
    with wave.open(wav, 'wb') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(pcm.tobytes())
    out = io.BytesIO()
    fe.main(['-c', cfgp, '-H', '--raw-output', wav], stdout=out)
    raw = out.getvalue()
    assert np.frombuffer(raw[:4], dtype='<i4')[0] == 39
    feats = np.frombuffer(raw[4:], dtype='<f4').reshape(-1, 39)
    assert feats.shape[0] == len(pcm) // 128
    assert np.max(np.abs(feats - m.features(pcm, cfg))) < 2e-3 * max(1.0, float(np.abs(feats).max()))
