#!/usr/bin/env python3
"""Writes tests/golden/feaconfig.json: the STRUCTURAL parameters of the reference's feature
configuration (fconfig.cfg:1-101) as parsed by speaker-diarization_amd/feaconfig.py, and a
SHA-256 of its trained arrays (normalization mean / scale, the 39x39 transform: 1 599
numbers of model data, which are NOT stored -- ADVICE r2).  Runs only where /root/reference
exists; the tests read the real arrays from there when it does and use synthetic ones
otherwise."""
import hashlib

import numpy as np
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = os.environ.get('SPKD_REFERENCE', '/root/reference')


def main():
    fc = importlib.import_module('speaker-diarization_amd.feaconfig')
    cfg = fc.FeatureConfig.load(os.path.join(REF, 'fconfig.cfg'))
    d = dict(sample_rate=cfg.sample_rate, frame_rate=cfg.frame_rate, window_width=cfg.window_width,
             pre_emph=cfg.pre_emph, copy_borders=cfg.copy_borders, magnitude=cfg.magnitude, n_cep=cfg.n_cep,
             zeroth=cfg.zeroth, cms_left=cfg.cms_left, cms_right=cfg.cms_right, delta_width=cfg.delta_width,
             delta_norm=cfg.delta_norm, dim=cfg.dim,
             arrays_sha256=hashlib.sha256(np.concatenate([cfg.mean, cfg.scale, cfg.transform.ravel()])
                                          .astype('<f4').tobytes()).hexdigest(),
             source="structural parameters parsed from the reference's fconfig.cfg:1-101 by "
                    "speaker-diarization_amd/feaconfig.py (tests/golden/make_golden_feaconfig.py); the trained "
                    "arrays (mean, scale, transform) are represented by the SHA-256 of their float32 values only")
    with open(os.path.join(HERE, 'feaconfig.json'), 'w') as f:
        json.dump(d, f, indent=1)


if __name__ == '__main__':
    main()
