"""ctypes binding of include/spkd.h (libspkd_hip.so).  No torch here: device
memory is addressed by raw pointers (``tensor.data_ptr()`` or ``dev_alloc``)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# (SPKD_HIP_LIBRARY: another build of the same library, for A/B measurements)
LIB_PATH = os.environ.get('SPKD_HIP_LIBRARY') or os.path.join(HERE, 'csrc', 'libspkd_hip.so')

SPKD_OK, SPKD_EINVAL, SPKD_EHIP, SPKD_ENONFINITE, SPKD_EOVERFLOW, SPKD_ENOMEM = range(6)
KINDS = {'BIC': 0, 'GLR': 1, 'KL2': 2}
WANT_GLR, WANT_KL2 = 1, 2
TIMERS = {n: i for i, n in enumerate(['call', 'chunk_stats', 'reduce_sets', 'pair_terms',
                                        'cluster_prep', 'matrix', 'ahc', 'gw', 'sw'])}
REC = 820
DIM = 39

EXPORTS = ['spkd_abi_version', 'spkd_create', 'spkd_create_on_stream', 'spkd_destroy', 'spkd_last_error', 'spkd_sync',
           'spkd_malloc', 'spkd_free', 'spkd_memcpy_h2d', 'spkd_memcpy_d2h', 'spkd_memcpy_d2d',
           'spkd_last_kernel_ms', 'spkd_last_gw_items', 'spkd_set_stats', 'spkd_pair_terms',
           'spkd_distance_matrix', 'spkd_gw_event_capacity', 'spkd_gw_event_capacity_p', 'spkd_gw', 'spkd_gw_ex', 'spkd_gw_fused', 'spkd_gather_stats', 'spkd_mfcc',
           'spkd_sw_window_count', 'spkd_sw', 'spkd_ahc', 'spkd_ahc_matrix', 'spkd_distance_rows', 'spkd_cluster_in', 'spkd_py2_roundtrip',
           'spkd_labels_from_merges', 'spkd_labels_from_merges_batch', 'spkd_count_flags', 'spkd_gw_lines']


class CdParams(C.Structure):
    _fields_ = [('kind', C.c_int32), ('trace', C.c_int32), ('lambdac', C.c_double),
                ('threshold', C.c_double), ('winsize', C.c_double), ('winstep', C.c_double),
                ('deltaws', C.c_double), ('rate', C.c_double)]


class CandLog(C.Structure):
    _fields_ = [('turn', C.c_int32), ('coarse', C.c_int32), ('seq', C.c_int64),
                ('start', C.c_double), ('i', C.c_double), ('d', C.c_double),
                ('n1', C.c_int64), ('n2', C.c_int64)]


AHC_AUTO, AHC_MONO, AHC_WIDE = 0, 1, 2


class MfccParams(C.Structure):
    _fields_ = [('sample_rate', C.c_int32), ('frame_rate', C.c_int32), ('window_width', C.c_int32),
                ('n_fft', C.c_int32), ('n_mel', C.c_int32), ('n_cep', C.c_int32), ('cms_left', C.c_int32),
                ('cms_right', C.c_int32), ('delta_width', C.c_int32 * 2), ('pre_emph', C.c_float),
                ('delta_norm', C.c_float * 2)]


class AhcParams(C.Structure):
    _fields_ = [('variant', C.c_int32), ('kind', C.c_int32), ('max_spk', C.c_int32),
                ('path', C.c_int32), ('lambdac', C.c_double), ('threshold', C.c_double)]


class SpkdError(RuntimeError):
    def __init__(self, status, text):
        RuntimeError.__init__(self, 'spkd status %d: %s' % (status, text))
        self.status = status


_lib = None


def _one_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch wheels bundle their own
    (torch/lib/libamdhip64.so): if this library is loaded first it pulls in the system
    runtime, and a later `import torch` finds that foreign runtime under its soname and
    reports "No HIP GPUs are available".  So when a PyTorch installation is present but not
    imported yet, its runtime is loaded first (the wheel's file only: torch itself is not
    imported), and libspkd_hip.so binds to it -- the same situation as after `import
    torch`.  SPKD_SYSTEM_HIP=1 keeps the system runtime."""
    import sys
    if 'torch' in sys.modules or os.environ.get('SPKD_SYSTEM_HIP'):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
        where = list(spec.submodule_search_locations)[0] if spec and spec.submodule_search_locations else None
    except (ImportError, ValueError, AttributeError):
        where = None
    if where:
        rt = os.path.join(where, 'lib', 'libamdhip64.so')
        if os.path.exists(rt):
            try:
                C.CDLL(rt, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def mapped_hip_runtimes():
    """The distinct libamdhip64 files mapped into this process (from /proc/self/maps)."""
    found = set()
    try:
        with open('/proc/self/maps') as f:
            for line in f:
                path = line.rsplit(None, 1)[-1] if '/' in line else ''
                if os.path.basename(path).startswith('libamdhip64.so'):
                    found.add(os.path.realpath(path))
    except OSError:
        pass
    return sorted(found)


def _check_one_runtime():
    """After libspkd_hip.so is loaded exactly one HIP runtime must be mapped: the preload of
    the PyTorch wheel's runtime only helps when its soname is the one this library was linked
    against -- otherwise the loader brings in the system runtime as well, silently, which is
    the very situation the preload is meant to prevent.  Two runtimes: an error
    naming both, and the way out (SPKD_SYSTEM_HIP=1 for torch-free processes; importing torch
    before this package otherwise)."""
    rts = mapped_hip_runtimes()
    if len(rts) > 1:
        raise ImportError('two HIP runtimes are mapped into this process (%s): libspkd_hip.so and the PyTorch '
                          'wheel link against different libamdhip64 sonames.  Set SPKD_SYSTEM_HIP=1 for a '
                          'torch-free process, or import torch before this package.' % ', '.join(rts))


def load_library(path=None):
    """Loads libspkd_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError('HIP extension %s is not built: run `python -c "import __graft_entry__ as g; '
                          'g.build()"` (or make -C speaker-diarization_amd/csrc)' % p)
    _one_hip_runtime()
    lib = C.CDLL(p)
    _check_one_runtime()
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    P = C.POINTER
    lib.spkd_abi_version.restype = C.c_int
    lib.spkd_create.argtypes = [C.c_int, vp, P(vp)]
    lib.spkd_create_on_stream.argtypes = [C.c_int, vp, P(vp)]
    lib.spkd_destroy.argtypes = [vp]
    lib.spkd_destroy.restype = None
    lib.spkd_last_error.argtypes = [vp]
    lib.spkd_last_error.restype = C.c_char_p
    lib.spkd_sync.argtypes = [vp]
    lib.spkd_malloc.argtypes = [vp, C.c_size_t, P(vp)]
    lib.spkd_free.argtypes = [vp, vp]
    lib.spkd_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.spkd_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    lib.spkd_memcpy_d2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.spkd_last_kernel_ms.argtypes = [vp, C.c_int, P(C.c_float)]
    lib.spkd_last_gw_items.argtypes = [vp, P(C.c_int64)]
    lib.spkd_set_stats.argtypes = [vp, vp, i64, vp, vp, vp, i64, i64, vp]
    lib.spkd_pair_terms.argtypes = [vp, vp, vp, vp, i64, C.c_int, vp]
    lib.spkd_distance_matrix.argtypes = [vp, C.c_int, dbl, vp, i64, vp]
    lib.spkd_gw_event_capacity.argtypes = [i64, dbl]
    lib.spkd_gw_event_capacity.restype = i64
    lib.spkd_gw_event_capacity_p.argtypes = [i64, P(CdParams)]
    lib.spkd_gw_event_capacity_p.restype = i64
    lib.spkd_gw.argtypes = [vp, vp, i64, vp, vp, i64, P(CdParams), vp, vp, vp, vp, vp, vp, vp, vp,
                            vp, i64, P(i64)]
    lib.spkd_gw_ex.argtypes = [vp, vp, i64, vp, vp, i64, P(CdParams), vp, C.c_int, vp, vp, vp, vp, vp, vp,
                               vp, vp, i64, P(i64)]
    lib.spkd_gw_fused.argtypes = [vp, vp, i64, vp, vp, i64, P(CdParams), vp, C.c_int, vp, vp, vp, vp, vp, vp,
                                  vp, vp, vp, i64, P(i64)]
    lib.spkd_gather_stats.argtypes = [vp, vp, i64, vp, vp, i64, i64, vp]
    lib.spkd_mfcc.argtypes = [vp, vp, i64, P(MfccParams), vp, vp, vp, vp, vp, vp, P(i64)]
    lib.spkd_sw_window_count.argtypes = [i64, dbl, dbl]
    lib.spkd_sw_window_count.restype = i64
    lib.spkd_sw.argtypes = [vp, vp, i64, vp, vp, i64, P(CdParams), vp, vp]
    lib.spkd_ahc.argtypes = [vp, vp, vp, i64, P(AhcParams), vp, vp, vp, vp, vp, vp]
    lib.spkd_ahc_matrix.argtypes = [vp, vp, i64, P(AhcParams), vp, dbl, dbl, vp, vp, vp, vp, vp, vp]
    lib.spkd_distance_rows.argtypes = [vp, C.c_int, C.c_int, dbl, vp, i64, i64, i64, vp, P(dbl), P(dbl)]
    lib.spkd_cluster_in.argtypes = [vp, vp, i64, C.c_int, dbl, dbl, vp, vp, i64, vp, P(i64), P(i64)]
    lib.spkd_py2_roundtrip.argtypes = [vp, i64]
    lib.spkd_py2_roundtrip.restype = None
    lib.spkd_labels_from_merges.argtypes = [i64, i64, vp, vp, vp]
    lib.spkd_labels_from_merges_batch.argtypes = [i64, vp, vp, vp, vp, vp]
    lib.spkd_count_flags.argtypes = [vp, vp, vp, i64, vp]
    lib.spkd_gw_lines.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, dbl, C.c_int, i64, vp, vp, vp, vp, vp]
    if lib.spkd_abi_version() != 2:
        raise ImportError('libspkd_hip.so ABI version mismatch')
    if path is None:
        _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def py2_roundtrip(values):
    """float(str(v)) with Python-2 str() for an array of values (host helper)."""
    v = np.ascontiguousarray(values, dtype=np.float64).copy()
    if v.size:
        load_library().spkd_py2_roundtrip(_ptr(v), v.size)
    return v


def labels_from_merges(n, a, b):
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    out = np.zeros(n, dtype=np.int32)
    st = load_library().spkd_labels_from_merges(n, len(a), _ptr(a), _ptr(b), _ptr(out))
    if st != SPKD_OK:
        raise SpkdError(st, 'bad merge log')
    return out


def count_flags(flags, off, n):
    """out[t] = number of non-zero flags among flags[off[t] : off[t] + n[t]] (host side)."""
    flags = np.ascontiguousarray(flags, dtype=np.int32)
    off = np.ascontiguousarray(off, dtype=np.int64)
    n = np.ascontiguousarray(n, dtype=np.int32)
    if len(off) != len(n):
        raise SpkdError(SPKD_EINVAL, 'one offset per group')
    if len(n) and (int((off + n).max()) > len(flags) or int(n.min()) < 0 or int(off.min()) < 0):
        raise SpkdError(SPKD_EINVAL, 'flag ranges outside the array')
    out = np.zeros(len(n), dtype=np.int32)
    st = load_library().spkd_count_flags(_ptr(flags), _ptr(off), _ptr(n), len(n), _ptr(out))
    if st != SPKD_OK:
        raise SpkdError(st, 'bad flag ranges')
    return out


def gw_lines(off, n_det, det_start, det_maxi, final_start, turn_start_s, turn_end_s, turn_begin, turn_end,
             rate, text_contract=True, want_frames=False):
    """The recipe lines of a gw() result in recipe order (spkd_gw_lines): dict with 'times'
    [n_lines, 2], 'turn' [n_lines] and, with want_frames, 'frame_b', 'frame_e', 'index'."""
    c = np.ascontiguousarray
    off = c(off, dtype=np.int64); n_det = c(n_det, dtype=np.int32)
    nt = len(n_det)
    arrs = [c(det_start, dtype=np.float64), c(det_maxi, dtype=np.float64), c(final_start, dtype=np.float64),
            c(turn_start_s, dtype=np.float64), c(turn_end_s, dtype=np.float64), c(turn_begin, dtype=np.int64),
            c(turn_end, dtype=np.int64)]
    if len(off) != nt or any(len(a) != nt for a in arrs[2:]):
        raise SpkdError(SPKD_EINVAL, 'one entry per turn')
    if nt and (int(n_det.min()) < 0 or int(off.min()) < 0 or int((off + n_det).max()) > min(len(arrs[0]), len(arrs[1]))):
        raise SpkdError(SPKD_EINVAL, 'detections outside the event arrays')
    n_lines = int(n_det.sum(dtype=np.int64)) + nt
    times = np.empty((n_lines, 2), dtype=np.float64)
    turn = np.empty(n_lines, dtype=np.int32)
    fb = np.empty(n_lines, dtype=np.int64) if want_frames else None
    fe = np.empty(n_lines, dtype=np.int64) if want_frames else None
    ix = np.empty(n_lines, dtype=np.int64) if want_frames else None
    p = lambda a: _ptr(a) if a is not None else None
    st = load_library().spkd_gw_lines(nt, _ptr(off), _ptr(n_det), *[_ptr(a) for a in arrs], float(rate),
                                      1 if text_contract else 0, n_lines, _ptr(times), p(fb), p(fe), p(ix), _ptr(turn))
    if st != SPKD_OK:
        raise SpkdError(st, 'bad growing-window result')
    return {'times': times, 'turn': turn, 'frame_b': fb, 'frame_e': fe, 'index': ix}


def labels_from_merges_batch(seg_off, n_merges, a, b):
    """Final 1-based cluster index of every record of every problem of an spkd_ahc
    result (same layout as its outputs)."""
    seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
    n_merges = np.ascontiguousarray(n_merges, dtype=np.int32)
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    out = np.zeros(int(seg_off[-1]), dtype=np.int32)
    st = load_library().spkd_labels_from_merges_batch(len(seg_off) - 1, _ptr(seg_off), _ptr(n_merges),
                                                      _ptr(a), _ptr(b), _ptr(out))
    if st != SPKD_OK:
        raise SpkdError(st, 'bad merge log')
    return out


class Context(object):
    """One (device, stream) context; owns nothing but the library's scratch."""

    def __init__(self, device=0, stream=None):
        """stream=None: the context owns a (blocking) stream.  stream=<handle>, 0 included:
        launch on exactly that hipStream_t (0 = the legacy default stream, which is what
        ``torch.cuda.current_stream().cuda_stream`` returns for torch's default stream)."""
        self.lib = load_library()
        h = C.c_void_p()
        if stream is None:
            st = self.lib.spkd_create(int(device), None, C.byref(h))
        else:
            st = self.lib.spkd_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(h))
        if st != SPKD_OK:
            raise SpkdError(st, 'spkd_create failed (no usable HIP device %d?)' % device)
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, 'h', None):
            for ptr, _ in self.__dict__.pop('_dev_scratch', {}).values():
                self.lib.spkd_free(self.h, C.c_void_p(ptr))
            self.lib.spkd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, st, allow=()):
        if st != SPKD_OK and st not in allow:
            raise SpkdError(st, self.lib.spkd_last_error(self.h).decode())
        return st

    def last_ms(self, which='call'):
        ms = C.c_float()
        self.check(self.lib.spkd_last_kernel_ms(self.h, TIMERS[which], C.byref(ms)))
        return ms.value

    def last_gw_items(self):
        """39x39 determinants evaluated by the most recent growing-window call."""
        n = C.c_int64()
        self.check(self.lib.spkd_last_gw_items(self.h, C.byref(n)))
        return int(n.value)

    # ---- device memory for torch-less hosts
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self.check(self.lib.spkd_malloc(self.h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, ptr):
        if ptr:
            self.check(self.lib.spkd_free(self.h, C.c_void_p(ptr)))

    def dev_scratch(self, name, nbytes):
        """Device buffer kept by the context and reused by later calls under the same
        name (hipMalloc / hipFree of a batch-sized buffer cost milliseconds each)."""
        cache = self.__dict__.setdefault('_dev_scratch', {})
        ptr, cap = cache.get(name, (None, 0))
        if ptr is None or cap < nbytes:
            if ptr is not None:
                self.dev_free(ptr)
            ptr, cap = self.dev_alloc(nbytes), nbytes
            cache[name] = (ptr, cap)
        return ptr

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.spkd_memcpy_h2d(self.h, C.c_void_p(dptr), _ptr(arr), arr.nbytes))

    def copy_d2d(self, d_dst, d_src, nbytes):
        self.check(self.lib.spkd_memcpy_d2d(self.h, C.c_void_p(d_dst), C.c_void_p(d_src), int(nbytes)))

    def d2h(self, arr, dptr):
        assert arr.flags['C_CONTIGUOUS']
        self.check(self.lib.spkd_memcpy_d2h(self.h, _ptr(arr), C.c_void_p(dptr), arr.nbytes))

    # ---- (1)
    def set_stats(self, d_frames, n_frames, begins, ends, sets, n_sets, d_stats):
        b = np.ascontiguousarray(begins, dtype=np.int64)
        e = np.ascontiguousarray(ends, dtype=np.int64)
        s = np.ascontiguousarray(sets, dtype=np.int32)
        self.check(self.lib.spkd_set_stats(self.h, C.c_void_p(d_frames), n_frames, _ptr(b), _ptr(e),
                                           _ptr(s), len(b), n_sets, C.c_void_p(d_stats)))

    # ---- (2)
    def pair_terms(self, d_stats, ia, ib, flags=0):
        ia = np.ascontiguousarray(ia, dtype=np.int32)
        ib = np.ascontiguousarray(ib, dtype=np.int32)
        out = np.empty((len(ia), 8), dtype=np.float64)
        st = self.lib.spkd_pair_terms(self.h, C.c_void_p(d_stats), _ptr(ia), _ptr(ib), len(ia),
                                      flags, _ptr(out))
        self.check(st, allow=(SPKD_ENONFINITE,))
        return out, st

    def distance_matrix(self, kind, lambdac, d_stats, n, d_matrix):
        st = self.lib.spkd_distance_matrix(self.h, KINDS[kind], lambdac, C.c_void_p(d_stats), n,
                                           C.c_void_p(d_matrix))
        return self.check(st, allow=(SPKD_ENONFINITE,))

    # ---- (3)
    def _buf(self, name, n, dtype):
        """Host output buffer reused across calls (fresh pages cost more than the copy)."""
        cache = self.__dict__.setdefault('_bufs', {})
        arr = cache.get(name)
        if arr is None or arr.shape[0] < n or arr.dtype != np.dtype(dtype):
            arr = cache[name] = np.empty(max(n, 1), dtype=dtype)
        return arr[:n]

    def gather_stats(self, d_src, n_src, src_index, d_dst, n_dst, dst_index=None):
        """d_dst[dst_index[i]] = d_src[src_index[i]] (whole statistics records)."""
        si = np.ascontiguousarray(src_index, dtype=np.int64)
        di = None if dst_index is None else np.ascontiguousarray(dst_index, dtype=np.int64)
        self.check(self.lib.spkd_gather_stats(self.h, C.c_void_p(d_src), n_src, _ptr(si),
                                              None if di is None else _ptr(di), len(si), n_dst,
                                              C.c_void_p(d_dst)))

    def gw(self, d_frames, n_frames, begins, ends, params, log_cap=4096, tight=False, reuse=False,
           seg_stats=None, first_guess_scale=1.0):
        """Event capacity: spkd_gw_event_capacity_p is a FIRST GUESS, not a bound -- it counts
        the scans of a window end that only moves forward, but after every detection the
        reference resets the end to start + 2 * winsize (spk-change-detection.py:264-266) and
        the window regrows over frames it has already scanned: a change point found early in a
        long-grown window can make a turn need more scans than turn_len / step + 8.  The
        kernel reports that safely (SPKD_EOVERFLOW, nothing written out of bounds) and the
        call is repeated here with doubled capacities until it fits (the reference simply
        produces output for such a turn; so does this).  tight=True starts from a quarter of
        the first guess: four times less data to allocate and copy in the typical case.
        first_guess_scale (tests): scales the first guess, to exercise the doubling.
        reuse=True returns views of per-context buffers: valid until the next gw call
        on this context.
        seg_stats: a callable n_records -> device pointer; turns the call into the fused
        form (spkd_gw_fused): the statistics record of segment j of turn t is left at
        record off[t] + j of that buffer."""
        b = np.ascontiguousarray(begins, dtype=np.int64)
        e = np.ascontiguousarray(ends, dtype=np.int64)
        nt = len(b)
        if not (params.rate >= 10.0 and params.winstep >= 1.0):
            raise SpkdError(SPKD_EINVAL, 'unsupported growing-window parameters (frame rate >= 10 and '
                                         'a window step of at least one frame needed)')
        # == spkd_gw_event_capacity_p(len, params), vectorised
        step = min(0.2 * params.rate, 0.5 * params.rate, params.winstep)
        full = ((e - b).astype(np.float64) / step).astype(np.int64) + 8
        if first_guess_scale != 1.0:
            full = np.maximum((full * float(first_guess_scale)).astype(np.int64), 2)
        grow = 1
        while True:
            caps = (full // 4 + 8) if tight else full * grow
            off = np.zeros(nt + 1, dtype=np.int64)
            off[1:] = np.cumsum(caps)
            nev = int(off[-1])
            mk = self._buf if reuse else (lambda name, n, dt: np.empty(n, dtype=dt))
            n_win = mk('n_win', nt, np.int32)
            win_maxd = mk('win_maxd', nev, np.float64)
            win_det = mk('win_det', nev, np.int32)
            det_start = mk('det_start', nev, np.float64)
            det_maxi = mk('det_maxi', nev, np.float64)
            det_d = mk('det_d', nev, np.float64)
            final_start = mk('final_start', nt, np.float64)
            log = (CandLog * max(log_cap, 1))()
            cnt = C.c_int64(0)
            if seg_stats is None:
                st = self.lib.spkd_gw_ex(self.h, C.c_void_p(d_frames), n_frames, _ptr(b), _ptr(e), nt,
                                         C.byref(params), _ptr(off), 0, _ptr(n_win),
                                         _ptr(win_maxd), _ptr(win_det), _ptr(det_start), _ptr(det_maxi),
                                         _ptr(det_d), _ptr(final_start), C.cast(log, C.c_void_p), log_cap,
                                         C.byref(cnt))
            else:
                d_seg = seg_stats(nev)
                st = self.lib.spkd_gw_fused(self.h, C.c_void_p(d_frames), n_frames, _ptr(b), _ptr(e), nt,
                                            C.byref(params), _ptr(off), 0, _ptr(n_win),
                                            _ptr(win_maxd), _ptr(win_det), _ptr(det_start), _ptr(det_maxi),
                                            _ptr(det_d), _ptr(final_start), C.c_void_p(d_seg),
                                            C.cast(log, C.c_void_p), log_cap, C.byref(cnt))
            if st == SPKD_EOVERFLOW and cnt.value > log_cap:
                log_cap = int(cnt.value) + 16      # the run is deterministic: retry with room
                continue
            if st == SPKD_EOVERFLOW and tight:
                tight = False                      # an unusually busy turn: the full first guess
                continue
            if st == SPKD_EOVERFLOW and grow < 1024:
                grow *= 2                          # a turn that re-scans (see above): double until it fits
                continue
            self.check(st, allow=(SPKD_ENONFINITE,))
            break
        return dict(status=st, off=off, n_ev=nev, n_win=n_win, win_maxd=win_maxd, win_det=win_det,
                    det_start=det_start, det_maxi=det_maxi, det_d=det_d, final_start=final_start,
                    log=log, log_count=min(int(cnt.value), log_cap))

    def sw(self, d_frames, n_frames, begins, ends, params):
        b = np.ascontiguousarray(begins, dtype=np.int64)
        e = np.ascontiguousarray(ends, dtype=np.int64)
        nt = len(b)
        cnt = [self.lib.spkd_sw_window_count(int(e[t] - b[t]), params.winsize, params.winstep)
               for t in range(nt)]
        if any(c < 0 for c in cnt):
            raise SpkdError(SPKD_EINVAL, 'window size and step must be at least one frame')
        off = np.zeros(nt + 1, dtype=np.int64)
        off[1:] = np.cumsum(cnt)
        d = np.zeros(int(off[-1]), dtype=np.float64)
        st = self.lib.spkd_sw(self.h, C.c_void_p(d_frames), n_frames, _ptr(b), _ptr(e), nt,
                              C.byref(params), _ptr(off), _ptr(d))
        self.check(st, allow=(SPKD_ENONFINITE,))
        return st, off, d

    # ---- (6)
    def mfcc(self, d_pcm, n_samples, params, melfb, dct, mean, scale, transform, d_features):
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (melfb, dct, mean, scale, transform)]
        n = C.c_int64(0)
        self.check(self.lib.spkd_mfcc(self.h, C.c_void_p(d_pcm), n_samples, C.byref(params), *[_ptr(a) for a in arrs],
                                      C.c_void_p(d_features), C.byref(n)))
        return int(n.value)

    # ---- (4)
    def ahc(self, d_stats, seg_off, params):
        seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
        npb = len(seg_off) - 1
        nt = int(seg_off[-1])
        n_merges = np.zeros(npb, dtype=np.int32)
        ma = np.zeros(nt, dtype=np.int32)
        mb = np.zeros(nt, dtype=np.int32)
        md = np.zeros(nt, dtype=np.float64)
        smax = np.zeros(npb, dtype=np.float64)
        smin = np.zeros(npb, dtype=np.float64)
        st = self.lib.spkd_ahc(self.h, C.c_void_p(d_stats), _ptr(seg_off), npb, C.byref(params),
                               _ptr(n_merges), _ptr(ma), _ptr(mb), _ptr(md), _ptr(smax), _ptr(smin))
        self.check(st, allow=(SPKD_ENONFINITE,))
        return dict(status=st, n_merges=n_merges, a=ma, b=mb, d=md, stat_max=smax, stat_min=smin)

    def distance_rows(self, variant, kind, lambdac, d_stats, n, row_begin, row_end, d_rows):
        """Rows [row_begin, row_end) of the initial matrix of one n-record problem -> d_rows
        (device, (row_end - row_begin) * n doubles); returns (max, min) over the block's finite
        distances (NaN: none)."""
        smax, smin = C.c_double(), C.c_double()
        st = self.lib.spkd_distance_rows(self.h, int(variant), KINDS[kind] if isinstance(kind, str) else int(kind),
                                         float(lambdac), C.c_void_p(d_stats), int(n), int(row_begin), int(row_end),
                                         C.c_void_p(d_rows), C.byref(smax), C.byref(smin))
        self.check(st, allow=(SPKD_ENONFINITE,))
        if st == SPKD_ENONFINITE:
            raise ValueError('array must not contain infs or NaNs')
        return smax.value, smin.value

    def cluster_in(self, d_stats, n, kind, lambdac, threshold):
        """spk_cluster_in over n records in recipe order as one device chain -> (labels[n],
        per-record distance lists (done records only), records done, clusters, status).  status is
        SPKD_ENONFINITE when a covariance with infs or NaNs stopped the chain at record `done`."""
        n = int(n)
        label = np.zeros(max(n, 1), dtype=np.int32)
        off = np.zeros(n + 1, dtype=np.int64)
        done, nclu = C.c_int64(0), C.c_int64(0)
        full = n * (n - 1) // 2
        cap = min(full, 32 * n)
        while True:
            dist = np.empty(max(cap, 1), dtype=np.float64)
            st = self.lib.spkd_cluster_in(self.h, C.c_void_p(d_stats), n, KINDS[kind] if isinstance(kind, str) else int(kind),
                                          float(lambdac), float(threshold), _ptr(label), _ptr(dist), cap, _ptr(off),
                                          C.byref(done), C.byref(nclu))
            if st == SPKD_EOVERFLOW and cap < full:
                cap = min(full, 4 * cap)
                continue
            break
        self.check(st, allow=(SPKD_ENONFINITE,))
        nd = int(done.value)
        dists = [dist[int(off[s]):int(off[s + 1])].copy() for s in range(nd)]
        return label[:n], dists, nd, int(nclu.value), st

    def ahc_matrix(self, d_stats, n, params, d_matrix, stat_max_in=float('nan'), stat_min_in=float('nan')):
        """The merge loop of one n-record problem on a caller-supplied initial matrix."""
        n = int(n)
        n_merges = np.zeros(1, dtype=np.int32)
        ma = np.zeros(n, dtype=np.int32)
        mb = np.zeros(n, dtype=np.int32)
        md = np.zeros(n, dtype=np.float64)
        smax = np.zeros(1, dtype=np.float64)
        smin = np.zeros(1, dtype=np.float64)
        st = self.lib.spkd_ahc_matrix(self.h, C.c_void_p(d_stats), n, C.byref(params), C.c_void_p(d_matrix),
                                      float(stat_max_in), float(stat_min_in), _ptr(n_merges), _ptr(ma), _ptr(mb),
                                      _ptr(md), _ptr(smax), _ptr(smin))
        self.check(st, allow=(SPKD_ENONFINITE,))
        return dict(status=st, n_merges=n_merges, a=ma, b=mb, d=md, stat_max=smax, stat_min=smin)
