"""CPU-side checks of the C-ABI boundary: the shared library loads and exports
every symbol include/spkd.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

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
    assert lib.spkd_abi_version() == 2


def _dynamic_exports(path):
    """Defined global symbols of an ELF64 shared object's .dynsym (no binutils needed)."""
    import struct
    data = open(path, 'rb').read()
    assert data[:4] == b'\x7fELF' and data[4] == 2
    shoff, = struct.unpack_from('<Q', data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from('<HHH', data, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from(
            '<IIQQQQIIQQ', data, shoff + i * shentsize)
        secs.append((typ, off, size, link, entsize))
    out = []
    for typ, off, size, link, entsize in secs:
        if typ != 11:                       # SHT_DYNSYM
            continue
        stroff = secs[link][1]
        for k in range(size // entsize):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from(
                '<IBBHQQ', data, off + k * entsize)
            if st_shndx == 0 or (st_info >> 4) not in (1, 2):     # undefined / not GLOBAL or WEAK
                continue
            end = data.index(b'\0', stroff + st_name)
            out.append(data[stroff + st_name:end].decode())
    return sorted(out)


def test_library_exports_only_the_c_abi():
    """libspkd_hip.so exports the functions of include/spkd.h and nothing else (kernel
    stubs, templates and helpers are local: csrc/libspkd_hip.map)."""
    hipabi = pkg('hipabi')
    assert _dynamic_exports(hipabi.LIB_PATH) == _declared()


def test_capacity_helpers_need_no_gpu():
    lib = pkg('hipabi').load_library()
    assert lib.spkd_gw_event_capacity(1000, 125.0) >= 1000 // 25
    assert lib.spkd_gw_event_capacity(1000, 5.0) == -1
    hipabi = pkg('hipabi')
    # DIA2 flags (winstep 375 frames): the same bound as the two-argument form
    p = hipabi.CdParams(0, 0, 1.0, 0.0, 125.0, 375.0, 12.0, 125.0)
    assert lib.spkd_gw_event_capacity_p(37500, p) == lib.spkd_gw_event_capacity(37500, 125.0)
    # -st 0.1 (12 frames): a 300 s turn without a change needs ~3000 windows
    p = hipabi.CdParams(0, 0, 1.0, 0.0, 125.0, 12.0, 12.0, 125.0)
    assert lib.spkd_gw_event_capacity_p(37500, p) >= 3018
    p = hipabi.CdParams(0, 0, 1.0, 0.0, 125.0, 0.0, 12.0, 125.0)
    assert lib.spkd_gw_event_capacity_p(37500, p) == -1
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


@pytest.mark.gpu
def test_library_before_torch_in_one_process():
    """PyTorch wheels bundle a HIP runtime of their own; a process that created a library
    context before importing torch used to end up with two runtimes and torch reported "No
    HIP GPUs are available".  hipabi loads the wheel's runtime first when a torch
    installation is present: both orders work (a fresh process: this one imported torch in
    conftest.py)."""
    pytest.importorskip('torch')
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'order_child.py')
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'then torch: True' in r.stdout and 'library again: merges' in r.stdout
