"""feacat raw feature files: ``int32 dim`` + ``float32[T][dim]``, little endian.

Reference: spk-change-detection.py:31-43 (path joined with os.path.join),
spk-clustering.py:31-43 (plain string concatenation, hence the caller's trailing
'/' fix-up at spk-clustering.py:357-358), spk-clustering2.py:32-44.
``T = size // dim`` (py2 integer division); a trailing partial frame makes the
reference's reshape raise, and so does this loader.
"""
import os.path as op
import numpy as np


def fea_path(audio, feapath, ext, join=True):
    base = op.splitext(op.basename(audio))[0] + ext
    return op.join(feapath, base) if join else feapath + base


def load_features(path):
    with open(path, 'rb') as f:
        head = np.fromfile(f, dtype='<i4', count=1)
        if head.size != 1:
            raise IOError('feature file too short: ' + path)
        dim = int(head[0])
        data = np.fromfile(f, dtype='<f4')
    if dim <= 0 or data.size % dim != 0:
        raise ValueError('cannot reshape array of size %d into shape (%d,%d)' % (
            data.size, data.size // max(dim, 1), dim))
    return dim, data.reshape(data.size // dim, dim)
