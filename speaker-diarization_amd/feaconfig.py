"""Reader of the AKU feature-configuration files feacat takes (`-c fconfig.cfg`): a list
of `module { name .. type .. key values.. sources .. }` blocks (fconfig.cfg:1-101).
Returns the parameters of the 39-dimensional MFCC chain the reference ships:

    audiofile -> fft (magnitude) -> mel -> dct(12) + power -> merge(13) -> mean_subtractor
    -> delta, delta-delta -> merge(39) -> normalization (mean, scale) -> lin_transform 39x39

The device code (csrc/spkd_mfcc.hpp) computes THIS chain and no other: a configuration
that names another topology, another merge order or other switches (power spectrum instead
of magnitude, a zeroth cepstrum, no border copies) is refused here rather than silently
turned into the shipped chain's features.
"""
import re

import numpy as np


def parse_modules(text):
    """[{'name':..., 'type':..., key: [values...]}] in file order."""
    mods = []
    for body in re.findall(r'module\s*\{(.*?)\}', text, flags=re.S):
        m = {}
        for line in body.strip().splitlines():
            toks = line.split()
            if toks:
                m[toks[0]] = toks[1:]
        mods.append(m)
    return mods


class FeatureConfig(object):
    def __init__(self, text):
        mods = parse_modules(text)
        by_type = {}
        for m in mods:
            by_type.setdefault(m['type'][0], []).append(m)

        def one(t):
            if t not in by_type:
                raise ValueError('feature configuration without a %s module' % t)
            return by_type[t][0]

        a = one('audiofile')
        self.sample_rate = int(a['sample_rate'][0])
        self.frame_rate = int(a['frame_rate'][0])
        self.window_width = int(a['window_width'][0])
        self.pre_emph = float(a.get('pre_emph_coef', ['0'])[0])
        self.copy_borders = int(a.get('copy_borders', ['0'])[0])
        self.magnitude = int(one('fft').get('magnitude', ['0'])[0])
        d = one('dct')
        self.n_cep = int(d['dim'][0])
        self.zeroth = int(d.get('zeroth', ['0'])[0])
        c = one('mean_subtractor')
        self.cms_left, self.cms_right = int(c['left'][0]), int(c['right'][0])
        deltas = by_type.get('delta', [])
        if len(deltas) != 2:
            raise ValueError('expected a delta and a delta-delta module')
        self.delta_width = [int(x['width'][0]) for x in deltas]
        self.delta_norm = [float(x['normalization'][0]) for x in deltas]
        n = one('normalization')
        self.mean = np.array([float(x) for x in n['mean']], dtype=np.float32)
        self.scale = np.array([float(x) for x in n['scale']], dtype=np.float32)
        t = one('lin_transform')
        self.dim = int(t['dim'][0])
        mat = np.array([float(x) for x in t['matrix']], dtype=np.float32)
        if mat.size != self.dim * self.dim or self.mean.size != self.dim or self.scale.size != self.dim:
            raise ValueError('normalization / transform sizes do not match dim %d' % self.dim)
        self.transform = mat.reshape(self.dim, self.dim)
        if self.dim != 3 * (self.n_cep + 1):
            raise ValueError('the chain gives %d features, the transform wants %d' % (3 * (self.n_cep + 1), self.dim))
        if self.sample_rate <= 0 or self.frame_rate <= 0 or self.sample_rate % self.frame_rate:
            raise ValueError('sample rate must be a positive multiple of the frame rate')
        self.hop = self.sample_rate // self.frame_rate
        self._check_supported(mods)

    # (type, sources by module TYPE) of the one chain the device code computes, in file order
    CHAIN = [('audiofile', []), ('fft', ['audiofile']), ('mel', ['fft']), ('power', ['fft']), ('dct', ['mel']),
             ('merge', ['dct', 'power']), ('mean_subtractor', ['merge']), ('delta', ['mean_subtractor']),
             ('delta', ['delta']), ('merge', ['mean_subtractor', 'delta', 'delta']),
             ('normalization', ['merge']), ('lin_transform', ['normalization'])]

    def _check_supported(self, mods):
        """Raises ValueError unless the configuration is the supported chain with the supported
        switches: magnitude 1, zeroth 0, copy_borders 1, n_mel / n_fft / window as built."""
        for key, want in (('magnitude', 1), ('zeroth', 0), ('copy_borders', 1)):
            if getattr(self, key) != want:
                raise ValueError('feature configuration: %s %d is not supported (this build computes %s %d)'
                                 % (key, getattr(self, key), key, want))
        if self.window_width != 400 or self.n_cep != 12:
            raise ValueError('feature configuration: this build does 400-sample windows and 12 cepstra')
        by_name = {m['name'][0]: m for m in mods if 'name' in m}
        got = []
        for m in mods:
            srcs = []
            for nm in m.get('sources', []):
                if nm not in by_name:
                    raise ValueError('feature configuration: module %s reads unknown source %s' % (m.get('name', ['?'])[0], nm))
                srcs.append(by_name[nm]['type'][0])
            got.append((m['type'][0], srcs))
        if got != self.CHAIN:
            raise ValueError('feature configuration: unsupported module chain %r; this build computes %r' % (got, self.CHAIN))
        # the second delta reads the first, the final merge is (cms, delta1, delta2) in that order
        deltas = [m for m in mods if m['type'][0] == 'delta']
        merges = [m for m in mods if m['type'][0] == 'merge']
        cms = [m for m in mods if m['type'][0] == 'mean_subtractor'][0]['name'][0]
        if deltas[1]['sources'] != [deltas[0]['name'][0]] or deltas[0]['sources'] != [cms] or \
                merges[1]['sources'] != [cms, deltas[0]['name'][0], deltas[1]['name'][0]]:
            raise ValueError('feature configuration: the delta / merge wiring differs from the supported chain')

    @classmethod
    def load(cls, path):
        with open(path) as f:
            return cls(f.read())
