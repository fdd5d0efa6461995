"""CPU-side checks of the C-ABI boundary: the shared library loads and exports
every symbol include/spkd.h declares (no compute calls without a GPU)."""
import os
import re

from helpers import ROOT
from conftest import pkg


def _declared():
    text = open(os.path.join(ROOT, 'include', 'spkd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(spkd_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported():
    hipabi = pkg('hipabi')
    lib = hipabi.load_library()
    names = _declared()
    assert 'spkd_gw' in names and 'spkd_ahc' in names
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(hipabi.EXPORTS) == names
    assert lib.spkd_abi_version() == 1


def test_capacity_helpers_need_no_gpu():
    lib = pkg('hipabi').load_library()
    assert lib.spkd_gw_event_capacity(1000, 125.0) >= 1000 // 25
    assert lib.spkd_gw_event_capacity(1000, 5.0) == -1
    assert lib.spkd_sw_window_count(1250, 625.0, 62.0) == 1
    assert lib.spkd_sw_window_count(1249, 625.0, 62.0) == 0
    assert lib.spkd_sw_window_count(1250 + 62, 625.0, 62.0) == 2


def test_product_has_no_cpu_path():
    """The product package must not import the oracle."""
    base = os.path.join(ROOT, 'speaker-diarization_amd')
    for fn in os.listdir(base):
        if fn.endswith('.py'):
            src = open(os.path.join(base, fn)).read()
            assert 'oracle' not in src.replace('the CPU oracle', '').replace('CPU oracle', ''), fn
