import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# PyTorch wheels bundle a HIP runtime of their own (torch/lib/libamdhip64.so).  A process
# that uses both torch and libspkd_hip.so must load torch's first: the library then binds
# to the runtime already in the process.  The other way round torch finds a foreign runtime
# under its soname and reports "No HIP GPUs are available" (seen when a test that only
# uses the library ran before the first test that touches torch.cuda).
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pkg(name=None):
    full = 'speaker-diarization_amd' + ('.' + name if name else '')
    return importlib.import_module(full)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
