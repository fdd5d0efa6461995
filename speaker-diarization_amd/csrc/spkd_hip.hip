// libspkd_hip.so — C ABI (include/spkd.h) over the HIP kernels.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/spkd.h"
#include "spkd_cd.hpp"
#include "spkd_cluster.hpp"
#include "spkd_device.hpp"
#include "spkd_stats.hpp"
#include "spkd_mfcc.hpp"

using namespace spkd;

namespace {
constexpr int N_SLOTS = 32;
}

struct spkd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool gw_lds_ok = false;
    int gw_waves = 0;
    unsigned long long init_keys[2] = {0ull, ~0ull};
    int step_waves = 0;          // step chain: waves per workgroup (0 = by problem size, 4, 8)
    int step_partners = 0;       // step chain: partners per workgroup (0 = by problem size, 3, 7, 15)
    int ahc_chain = 0;           // wide merge loop: 0 / 2 = the step chain (no hand-offs), 1 = the ticket chain
    int64_t last_gw_items = 0;
    std::string err;
    int* d_err = nullptr;
    unsigned long long* d_counter = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    // per-kernel timers (HIP events on the launch stream), see spkd_last_kernel_ms
    hipEvent_t ka[SPKD_N_TIMERS] = {}, kb[SPKD_N_TIMERS] = {};
    bool kused[SPKD_N_TIMERS] = {};
    float kms[SPKD_N_TIMERS] = {};
    void* slot[N_SLOTS] = {};
    size_t slot_bytes[N_SLOTS] = {};
};

namespace {

spkd_status fail(spkd_ctx* c, spkd_status s, const std::string& msg) {
    if (c) c->err = msg;
    return s;
}

#define HIPCHK(c, call)                                                              \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess)                                                        \
            return fail((c), SPKD_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// grow-only scratch buffers (no hipMalloc in the steady state)
spkd_status scratch(spkd_ctx* c, int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    if (c->slot_bytes[slot] < bytes) {
        if (c->slot[slot]) HIPCHK(c, hipFree(c->slot[slot]));
        c->slot[slot] = nullptr;
        c->slot_bytes[slot] = 0;
        size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&c->slot[slot], want);
        if (e != hipSuccess) {
            want = bytes;
            e = hipMalloc(&c->slot[slot], want);
        }
        if (e != hipSuccess) return fail(c, SPKD_ENOMEM, "scratch allocation failed");
        c->slot_bytes[slot] = want;
    }
    *out = c->slot[slot];
    return SPKD_OK;
}

template <class T>
spkd_status upload(spkd_ctx* c, int slot, const std::vector<T>& h, T** d) {
    void* p = nullptr;
    spkd_status s = scratch(c, slot, h.size() * sizeof(T), &p);
    if (s != SPKD_OK) return s;
    if (!h.empty()) HIPCHK(c, hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *d = (T*)p;
    return SPKD_OK;
}

#define TIMED(c, idx, launch)                                   \
    do {                                                         \
        (void)hipEventRecord((c)->ka[idx], (c)->stream);         \
        launch;                                                  \
        (void)hipEventRecord((c)->kb[idx], (c)->stream);         \
        (c)->kused[idx] = true;                                  \
    } while (0)

spkd_status begin_call(spkd_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < SPKD_N_TIMERS; ++i) c->kused[i] = false;
    HIPCHK(c, hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    return SPKD_OK;
}

// records the end event, waits, folds the device error word into a status
spkd_status end_call(spkd_ctx* c) {
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    int herr = 0;
    HIPCHK(c, hipMemcpyAsync(&herr, c->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
    c->kms[0] = c->last_ms;
    for (int i = 1; i < SPKD_N_TIMERS; ++i)
        if (c->kused[i]) HIPCHK(c, hipEventElapsedTime(&c->kms[i], c->ka[i], c->kb[i]));
    if (herr & ERR_SWEEP) return fail(c, SPKD_EHIP, "internal error: growing-window sweep order");
    if (herr & 4) return fail(c, SPKD_EOVERFLOW, "device scratch capacity exceeded");
    if (herr & ERR_DEGENERATE_MERGE) return fail(c, SPKD_EINVAL, "degenerate merge: a diagonal cell was the minimum");
    if (herr & ERR_NONFINITE) return fail(c, SPKD_ENONFINITE, "array must not contain infs or NaNs");
    return SPKD_OK;
}

enum {
    S_CHUNKS = 0, S_SETOFF, S_PARTIAL, S_IDXA, S_IDXB, S_TERMS, S_TURNS, S_SNAP, S_CAND,
    S_EV_I32A, S_EV_I32B, S_EV_D0, S_EV_D1, S_EV_D2, S_EV_D3, S_EV_D4, S_LOG,
    S_AHC_STATS, S_AHC_LD, S_AHC_AUX, S_AHC_MAT, S_AHC_MISC, S_AHC_OUT, S_AHC_OFF, S_AHC_PROB, S_AHC_PACKED, S_MFCC_TAB, S_MFCC_STATIC,
    S_STEP_EXM, S_STEP_PKM, S_STEP_MISC, S_COUNT
};
static_assert(S_COUNT <= N_SLOTS, "scratch slot table too small");

}  // namespace

extern "C" {

int spkd_abi_version(void) { return SPKD_ABI_VERSION; }

static spkd_status create_ctx(int device, void* stream, bool borrow, spkd_ctx** out) {
    if (!out) return SPKD_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return SPKD_EHIP;
    spkd_ctx* c = new spkd_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return SPKD_EHIP; }
    if (borrow) {
        c->stream = (hipStream_t)stream;           // NULL = the legacy default stream itself
    } else {
        // a BLOCKING stream: ordered after (and before) work on the legacy default stream,
        // which is where torch's default stream puts the producers of d_frames
        if (hipStreamCreateWithFlags(&c->stream, hipStreamDefault) != hipSuccess) { delete c; return SPKD_EHIP; }
        c->own_stream = true;
    }
    if (hipMalloc(&c->d_err, sizeof(int)) != hipSuccess ||
        hipMalloc(&c->d_counter, 2 * sizeof(unsigned long long)) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        spkd_destroy(c);
        return SPKD_EHIP;
    }
    for (int i = 0; i < SPKD_N_TIMERS; ++i)
        if (hipEventCreate(&c->ka[i]) != hipSuccess || hipEventCreate(&c->kb[i]) != hipSuccess) {
            spkd_destroy(c);
            return SPKD_EHIP;
        }
    // k_gw carves its LDS dynamically: the size must fit the device and be admitted
    // for the kernel, or every later launch fails -- checked once, reported by spkd_gw
    int lds_max = 0;
    c->gw_lds_ok = hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess &&
                   Gw<4>::LDS_BYTES <= lds_max &&
                   hipFuncSetAttribute((const void*)k_gw<8>, hipFuncAttributeMaxDynamicSharedMemorySize, Gw<8>::LDS_BYTES) == hipSuccess &&
                   hipFuncSetAttribute((const void*)k_gw<4>, hipFuncAttributeMaxDynamicSharedMemorySize, Gw<4>::LDS_BYTES) == hipSuccess &&
                   hipFuncSetAttribute((const void*)k_gw<2>, hipFuncAttributeMaxDynamicSharedMemorySize, Gw<2>::LDS_BYTES) == hipSuccess &&
                   hipFuncSetAttribute((const void*)k_gw<1>, hipFuncAttributeMaxDynamicSharedMemorySize, Gw<1>::LDS_BYTES) == hipSuccess;
    // waves per turn of the growing-window kernel: 0 = by the number of turns (gw_impl)
    if (const char* e = getenv("SPKD_GW_WAVES")) c->gw_waves = atoi(e);
    if (const char* e = getenv("SPKD_AHC_CHAIN")) c->ahc_chain = atoi(e);
    if (const char* e = getenv("SPKD_STEP_WAVES")) { const int v = atoi(e); c->step_waves = (v == 4 || v == 8) ? v : 0; }
    if (const char* e = getenv("SPKD_STEP_PARTNERS")) { const int v = atoi(e); c->step_partners = (v == 3 || v == 7 || v == 15) ? v : 0; }
    *out = c;
    return SPKD_OK;
}

spkd_status spkd_create(int device, void* stream, spkd_ctx** out) {
    return create_ctx(device, stream, stream != nullptr, out);
}

spkd_status spkd_create_on_stream(int device, void* stream, spkd_ctx** out) {
    return create_ctx(device, stream, true, out);
}

void spkd_destroy(spkd_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < N_SLOTS; ++i)
        if (c->slot[i]) (void)hipFree(c->slot[i]);
    if (c->d_err) (void)hipFree(c->d_err);
    if (c->d_counter) (void)hipFree(c->d_counter);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (int i = 0; i < SPKD_N_TIMERS; ++i) {
        if (c->ka[i]) (void)hipEventDestroy(c->ka[i]);
        if (c->kb[i]) (void)hipEventDestroy(c->kb[i]);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* spkd_last_error(const spkd_ctx* c) { return c ? c->err.c_str() : "null context"; }

spkd_status spkd_sync(spkd_ctx* c) {
    if (!c) return SPKD_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPKD_OK;
}

spkd_status spkd_malloc(spkd_ctx* c, size_t bytes, void** d_ptr) {
    if (!c || !d_ptr) return SPKD_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(c, SPKD_ENOMEM, hipGetErrorString(e));
    return SPKD_OK;
}

spkd_status spkd_free(spkd_ctx* c, void* d_ptr) {
    if (!c) return SPKD_EINVAL;
    HIPCHK(c, hipFree(d_ptr));
    return SPKD_OK;
}

spkd_status spkd_memcpy_h2d(spkd_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
    if (!c) return SPKD_EINVAL;
    HIPCHK(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPKD_OK;
}

spkd_status spkd_memcpy_d2h(spkd_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
    if (!c) return SPKD_EINVAL;
    HIPCHK(c, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPKD_OK;
}

spkd_status spkd_memcpy_d2d(spkd_ctx* c, void* d_dst, const void* d_src, size_t bytes) {
    if (!c) return SPKD_EINVAL;
    HIPCHK(c, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPKD_OK;
}

spkd_status spkd_last_gw_items(spkd_ctx* c, int64_t* items) {
    if (!c || !items) return SPKD_EINVAL;
    *items = c->last_gw_items;
    return SPKD_OK;
}

spkd_status spkd_last_kernel_ms(spkd_ctx* c, int which, float* ms) {
    if (!c || !ms || which < 0 || which >= SPKD_N_TIMERS) return SPKD_EINVAL;
    *ms = c->kms[which];
    return SPKD_OK;
}

// ------------------------------------------------------------------ (1) stats
namespace {
// the launches of spkd_set_stats, without the call bracket (spkd_sw uses them too)
spkd_status set_stats_launch(spkd_ctx* c, const float* d_frames, int64_t n_frames,
                             const int64_t* h_begin, const int64_t* h_end, const int32_t* h_set,
                             int64_t n_ranges, int64_t n_sets, double* d_stats,
                             std::vector<Chunk>& chunks, std::vector<int64_t>& set_off) {   // both must outlive the stream work
    chunks.clear();
    set_off.assign((size_t)n_sets + 1, 0);
    int32_t prev = 0;
    for (int64_t r = 0; r < n_ranges; ++r) {
        const int64_t b = h_begin[r], e = h_end[r];
        const int32_t s = h_set[r];
        if (b < 0 || e < b || e > n_frames || s < prev || s >= n_sets)
            return fail(c, SPKD_EINVAL, "bad frame range or set id");
        prev = s;
        for (int64_t t = b; t < e; t += STATS_CHUNK) {
            Chunk ch;
            ch.begin = t;
            ch.len = (int32_t)std::min<int64_t>(STATS_CHUNK, e - t);
            ch.set = s;
            chunks.push_back(ch);
            set_off[(size_t)s + 1]++;
        }
    }
    for (int64_t s = 0; s < n_sets; ++s) set_off[(size_t)s + 1] += set_off[(size_t)s];
    spkd_status st;
    Chunk* d_chunks = nullptr;
    int64_t* d_setoff = nullptr;
    void* d_partial = nullptr;
    if ((st = upload(c, S_CHUNKS, chunks, &d_chunks)) != SPKD_OK) return st;
    if ((st = upload(c, S_SETOFF, set_off, &d_setoff)) != SPKD_OK) return st;
    if ((st = scratch(c, S_PARTIAL, chunks.size() * REC * sizeof(double), &d_partial)) != SPKD_OK) return st;
    if (!chunks.empty())
        TIMED(c, SPKD_T_CHUNK_STATS,
              hipLaunchKernelGGL(k_chunk_stats, dim3((unsigned)chunks.size()), dim3(STATS_TPB), 0, c->stream,
                                 d_frames, d_chunks, (double*)d_partial));
    TIMED(c, SPKD_T_REDUCE_SETS,
          hipLaunchKernelGGL(k_reduce_sets, dim3((unsigned)n_sets), dim3(STATS_TPB), 0, c->stream,
                             (const double*)d_partial, d_setoff, d_stats));
    HIPCHK(c, hipGetLastError());
    return SPKD_OK;
}
}  // namespace

spkd_status spkd_set_stats(spkd_ctx* c, const float* d_frames, int64_t n_frames,
                           const int64_t* h_begin, const int64_t* h_end, const int32_t* h_set,
                           int64_t n_ranges, int64_t n_sets, double* d_stats) {
    if (!c || !d_stats || n_sets < 0 || n_ranges < 0) return SPKD_EINVAL;
    if (n_sets == 0) return SPKD_OK;
    if (n_ranges > 0 && (!h_begin || !h_end || !h_set || !d_frames)) return fail(c, SPKD_EINVAL, "null range arrays");
    spkd_status st = begin_call(c);
    if (st != SPKD_OK) return st;
    std::vector<Chunk> chunks;
    std::vector<int64_t> set_off;
    if ((st = set_stats_launch(c, d_frames, n_frames, h_begin, h_end, h_set, n_ranges, n_sets, d_stats, chunks, set_off)) != SPKD_OK) {
        (void)hipStreamSynchronize(c->stream);
        return st;
    }
    return end_call(c);
}

// ------------------------------------------------------------------ (2) pair terms
spkd_status spkd_pair_terms(spkd_ctx* c, const double* d_stats, const int32_t* h_a, const int32_t* h_b,
                            int64_t n_pairs, int flags, double* h_terms) {
    if (!c || n_pairs < 0) return SPKD_EINVAL;
    if (n_pairs == 0) return SPKD_OK;
    if (!d_stats || !h_a || !h_b || !h_terms) return fail(c, SPKD_EINVAL, "null argument");
    spkd_status st = begin_call(c);
    if (st != SPKD_OK) return st;
    std::vector<int32_t> va(h_a, h_a + n_pairs), vb(h_b, h_b + n_pairs);
    int32_t *d_a = nullptr, *d_b = nullptr;
    void* d_terms = nullptr;
    if ((st = upload(c, S_IDXA, va, &d_a)) != SPKD_OK) return st;
    if ((st = upload(c, S_IDXB, vb, &d_b)) != SPKD_OK) return st;
    if ((st = scratch(c, S_TERMS, (size_t)n_pairs * 8 * sizeof(double), &d_terms)) != SPKD_OK) return st;
    const unsigned blocks = (unsigned)((n_pairs + PT2_WAVES - 1) / PT2_WAVES);
    TIMED(c, SPKD_T_PAIR_TERMS,
          hipLaunchKernelGGL(k_pair_terms, dim3(blocks), dim3(PT2_WAVES * WAVE), 0, c->stream,
                             d_stats, d_a, d_b, n_pairs, flags, (double*)d_terms, c->d_err));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_terms, d_terms, (size_t)n_pairs * 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return end_call(c);
}

// ------------------------------------------------------------------ clustering internals
namespace {
struct AhcBuffers {
    double* ex;
    double* pk;          // packed working copies (the pair passes load these)
    double* ld;
    double* aux;
    double* mat;
    int64_t* seg_off;
    int64_t* mat_off;
    unsigned long long* smax;
    unsigned long long* smin;
};

unsigned long long host_dkey(double v) {                      // dkey() of spkd_cluster.hpp, on the host
    unsigned long long b;
    std::memcpy(&b, &v, sizeof b);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// What the initial matrix is: computed here in full (the default), only the rows
// [row_begin, row_end) of a single problem (spkd_distance_rows: a rank's block of a matrix
// tiled over several GPUs), or given by the caller (spkd_ahc_matrix: the gathered blocks).
struct MatrixPlan {
    const double* d_init = nullptr;
    double init_max = NAN, init_min = NAN;            // variant 1: max / min over the distances behind d_init
    int64_t row_begin = -1, row_end = -1;
};

spkd_status ahc_prepare(spkd_ctx* c, const double* d_stats, const int64_t* h_seg_off, int64_t n_prob,
                        int variant, int kind, double lambdac, AhcBuffers& B, int64_t& n_total,
                        std::vector<int64_t>& offs, std::vector<int32_t>& prob_of,      // both must outlive the stream work
                        const MatrixPlan& plan = MatrixPlan()) {
    n_total = h_seg_off[n_prob];
    offs.clear();                                    // seg_off | mat_off
    offs.insert(offs.end(), h_seg_off, h_seg_off + n_prob + 1);
    int64_t cells = 0;
    for (int64_t p = 0; p < n_prob; ++p) {
        const int64_t n = h_seg_off[p + 1] - h_seg_off[p];
        if (n < 0) return fail(c, SPKD_EINVAL, "seg_off must be non-decreasing");
        offs.push_back(cells);
        cells += n * n;
    }
    offs.push_back(cells);
    prob_of.assign((size_t)n_total, 0);              // record -> its problem
    for (int64_t p = 0; p < n_prob; ++p)
        for (int64_t r = h_seg_off[p]; r < h_seg_off[p + 1]; ++r) prob_of[(size_t)r] = (int32_t)p;
    // k_matrix's launch order, appended to the same upload: block b computes the matrix row
    // of record sched[b] (-1: nothing).  Workgroups go to the eight XCDs round-robin by block
    // index and every XCD has an L2 of its own, so the rows of one problem -- they all read
    // that problem's records as partners -- are given to ONE XCD: its working set is then a
    // problem or two (2.5 MB each at N = 390) instead of a slice of all of them.  Whole
    // problems are dealt to the least-loaded XCD, largest first; with fewer problems than
    // XCDs the rows are dealt round-robin instead.
    int64_t grid_rows = 0;
    {
        const int X = 8;
        std::vector<std::vector<int32_t>> lists(X);
        if (plan.row_begin >= 0) {
            // a block of rows of the one problem: dealt round-robin (every XCD holds the records)
            for (int64_t r = plan.row_begin; r < plan.row_end; ++r) lists[(size_t)((r - plan.row_begin) % X)].push_back((int32_t)r);
        } else if (n_prob >= X) {
            std::vector<int64_t> order((size_t)n_prob);
            for (int64_t p = 0; p < n_prob; ++p) order[(size_t)p] = p;
            std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
                return h_seg_off[a + 1] - h_seg_off[a] > h_seg_off[b + 1] - h_seg_off[b];
            });
            double load[X] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int64_t p : order) {
                int x = 0;
                for (int i = 1; i < X; ++i) if (load[i] < load[x]) x = i;
                const double n = (double)(h_seg_off[p + 1] - h_seg_off[p]);
                load[x] += n * n;
                for (int64_t r = h_seg_off[p]; r < h_seg_off[p + 1]; ++r) lists[x].push_back((int32_t)r);
            }
        } else {
            for (int64_t r = 0; r < n_total; ++r) lists[(size_t)(r % X)].push_back((int32_t)r);
        }
        size_t longest = 0;
        for (auto& l : lists) longest = std::max(longest, l.size());
        grid_rows = (int64_t)longest * X;
        prob_of.resize((size_t)(n_total + grid_rows), -1);
        for (int x = 0; x < X; ++x)
            for (size_t i = 0; i < lists[x].size(); ++i) prob_of[(size_t)n_total + i * X + x] = lists[x][i];
    }
    int64_t* d_offs = nullptr;
    int32_t* d_prob = nullptr;
    spkd_status st;
    if ((st = upload(c, S_AHC_OFF, offs, &d_offs)) != SPKD_OK) return st;
    if ((st = upload(c, S_AHC_PROB, prob_of, &d_prob)) != SPKD_OK) return st;
    B.seg_off = d_offs;
    B.mat_off = d_offs + n_prob + 1;
    void* p = nullptr;
    if ((st = scratch(c, S_AHC_LD, (size_t)n_total * sizeof(double), &p)) != SPKD_OK) return st;
    B.ld = (double*)p;
    if ((st = scratch(c, S_AHC_AUX, (size_t)n_total * AUX * sizeof(double), &p)) != SPKD_OK) return st;
    B.aux = (double*)p;
    if ((st = scratch(c, S_AHC_MAT, (size_t)cells * sizeof(double), &p)) != SPKD_OK) return st;
    B.mat = (double*)p;
    if ((st = scratch(c, S_AHC_MISC, (size_t)n_prob * 2 * sizeof(unsigned long long), &p)) != SPKD_OK) return st;
    B.smax = (unsigned long long*)p;
    B.smin = B.smax + n_prob;
    HIPCHK(c, hipMemsetAsync(B.smax, 0x00, (size_t)n_prob * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(B.smin, 0xff, (size_t)n_prob * sizeof(unsigned long long), c->stream));
    if (plan.d_init && n_prob == 1) {
        // (the keys live in the context: the copies are asynchronous)
        c->init_keys[0] = plan.init_max == plan.init_max ? host_dkey(plan.init_max) : 0ull;
        c->init_keys[1] = plan.init_min == plan.init_min ? host_dkey(plan.init_min) : ~0ull;
        HIPCHK(c, hipMemcpyAsync(B.smax, &c->init_keys[0], sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(B.smin, &c->init_keys[1], sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    }
    // a private working copy of the records (clusters are merged in place), expanded to the
    // quad layout the clustering kernels load from
    if ((st = scratch(c, S_AHC_STATS, (size_t)n_total * QREC * sizeof(double), &p)) != SPKD_OK) return st;
    B.ex = (double*)p;
    if ((st = scratch(c, S_AHC_PACKED, (size_t)n_total * REC * sizeof(double), &p)) != SPKD_OK) return st;
    B.pk = (double*)p;
    if (n_total > 0) {
        hipLaunchKernelGGL(k_to_quadrec, dim3((unsigned)n_total), dim3(256), 0, c->stream, d_stats, n_total, B.ex);
        HIPCHK(c, hipMemcpyAsync(B.pk, d_stats, (size_t)n_total * REC * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    if (n_total > 0) {
        // KL2: one wave per record; BIC / GLR: four records per wave
        const int64_t per_block = kind == SPKD_KL2 ? PT_WAVES : 4 * PT_WAVES;
        const unsigned blocks = (unsigned)((n_total + per_block - 1) / per_block);
        TIMED(c, SPKD_T_CLUSTER_PREP,
              hipLaunchKernelGGL(k_cluster_prep, dim3(blocks), dim3(PT_WAVES * WAVE), 0, c->stream,
                                 (const double*)B.ex, n_total, kind, B.ld, B.aux, c->d_err));
        auto kmat = kind == SPKD_GLR ? k_matrix<true> : k_matrix<false>;   // GLR has a second rank-one term
        if (plan.d_init) {
            HIPCHK(c, hipMemcpyAsync(B.mat, plan.d_init, (size_t)cells * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        } else if (grid_rows > 0) {
            TIMED(c, SPKD_T_MATRIX,
                  hipLaunchKernelGGL(kmat, dim3((unsigned)grid_rows), dim3(MX_WAVES * WAVE), 0, c->stream,
                                     (const double*)B.ex, (const double*)B.pk, (const int64_t*)B.seg_off, (const int32_t*)d_prob,
                                     (const int32_t*)d_prob + n_total, variant, kind, lambdac,
                                     (const double*)B.ld, (const double*)B.aux, B.mat, (const int64_t*)B.mat_off,
                                     B.smax, B.smin, c->d_err));
        }
    }
    HIPCHK(c, hipGetLastError());
    return SPKD_OK;
}

double key_to_double(unsigned long long k, bool is_max) {
    if (is_max ? (k == 0ull) : (k == ~0ull)) return std::nan("");
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double v;
    std::memcpy(&v, &b, sizeof v);
    return v;
}
}  // namespace

spkd_status spkd_distance_matrix(spkd_ctx* c, int kind, double lambdac, const double* d_stats,
                                 int64_t n, double* d_matrix) {
    if (!c || n < 0 || kind < 0 || kind > 2) return SPKD_EINVAL;
    if (n == 0) return SPKD_OK;
    if (!d_stats || !d_matrix) return fail(c, SPKD_EINVAL, "null argument");
    spkd_status st = begin_call(c);
    if (st != SPKD_OK) return st;
    const int64_t seg_off[2] = {0, n};
    AhcBuffers B;
    int64_t n_total = 0;
    std::vector<int64_t> offs;
    std::vector<int32_t> prob_of;
    if ((st = ahc_prepare(c, d_stats, seg_off, 1, 1, kind, lambdac, B, n_total, offs, prob_of)) != SPKD_OK) return st;
    HIPCHK(c, hipMemcpyAsync(d_matrix, B.mat, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return end_call(c);
}

spkd_status spkd_cluster_in(spkd_ctx* c, const double* d_stats, int64_t n, int kind, double lambdac, double threshold,
                            int32_t* h_label, double* h_dist, int64_t dist_cap, int64_t* h_dist_off,
                            int64_t* h_n_done, int64_t* h_n_clusters) {
    if (!c || n < 0 || dist_cap < 0 || kind < 0 || kind > 2) return SPKD_EINVAL;
    if (h_n_done) *h_n_done = 0;
    if (h_n_clusters) *h_n_clusters = 0;
    if (n == 0) return SPKD_OK;
    if (!d_stats || !h_label || !h_dist_off || !h_n_done || !h_n_clusters || (dist_cap > 0 && !h_dist))
        return fail(c, SPKD_EINVAL, "null argument");
    spkd_status st = begin_call(c);
    if (st != SPKD_OK) return st;
    void *p_ex = nullptr, *p_ld = nullptr, *p_aux = nullptr, *p_cex = nullptr, *p_cpk = nullptr, *p_misc = nullptr, *p_dist = nullptr;
    const size_t nn = (size_t)n;
    if ((st = scratch(c, S_AHC_STATS, nn * QREC * sizeof(double), &p_ex)) != SPKD_OK) return st;
    if ((st = scratch(c, S_AHC_LD, nn * sizeof(double), &p_ld)) != SPKD_OK) return st;
    if ((st = scratch(c, S_AHC_AUX, nn * AUX * sizeof(double), &p_aux)) != SPKD_OK) return st;
    if ((st = scratch(c, S_STEP_EXM, nn * QREC * sizeof(double), &p_cex)) != SPKD_OK) return st;
    if ((st = scratch(c, S_STEP_PKM, nn * REC * sizeof(double), &p_cpk)) != SPKD_OK) return st;
    // misc: cluster log dets | determinants of a step | cluster KL2 vectors | dist_off | done | labels
    const size_t misc_bytes = (2 + AUX) * nn * sizeof(double) + (nn + 1 + 2) * sizeof(long long) + nn * sizeof(int32_t) + 64;
    if ((st = scratch(c, S_STEP_MISC, misc_bytes, &p_misc)) != SPKD_OK) return st;
    if ((st = scratch(c, S_AHC_MAT, (size_t)std::max<int64_t>(dist_cap, 1) * sizeof(double), &p_dist)) != SPKD_OK) return st;
    double* clu_ld = (double*)p_misc;
    double* tmp = clu_ld + nn;
    double* clu_aux = tmp + nn;
    long long* d_off = (long long*)(clu_aux + nn * AUX);
    long long* d_done = d_off + nn + 1;
    int32_t* d_label = (int32_t*)(d_done + 2);
    HIPCHK(c, hipMemsetAsync(d_done, 0, 2 * sizeof(long long), c->stream));
    // the records in the quad layout, their own log dets (four records per wave)
    hipLaunchKernelGGL(k_to_quadrec, dim3((unsigned)n), dim3(256), 0, c->stream, d_stats, n, (double*)p_ex);
    {
        const int64_t per_block = 4 * PT_WAVES;
        const unsigned blocks = (unsigned)((n + per_block - 1) / per_block);
        // (the log dets also for KL2: they are what flags a covariance with infs or NaNs, which the
        // reference's pinv refuses like its det)
        TIMED(c, SPKD_T_CLUSTER_PREP,
              hipLaunchKernelGGL(k_cluster_prep, dim3(blocks), dim3(PT_WAVES * WAVE), 0, c->stream,
                                 (const double*)p_ex, n, kind == SPKD_KL2 ? (int)SPKD_BIC : kind, (double*)p_ld, (double*)p_aux, c->d_err));
        if (kind == SPKD_KL2) {
            void* p_ld2 = nullptr;                       // (the KL2 pass of the same kernel rewrites ld with zeros)
            if ((st = scratch(c, S_AHC_OUT, nn * sizeof(double), &p_ld2)) != SPKD_OK) return st;
            const unsigned blocks1 = (unsigned)((n + PT_WAVES - 1) / PT_WAVES);
            hipLaunchKernelGGL(k_cluster_prep, dim3(blocks1), dim3(PT_WAVES * WAVE), 0, c->stream,
                               (const double*)p_ex, n, (int)SPKD_KL2, (double*)p_ld2, (double*)p_aux, c->d_err);
        }
    }
    auto kin = kind == SPKD_GLR ? k_cluster_in<true> : k_cluster_in<false>;
    (void)hipEventRecord(c->ka[SPKD_T_AHC], c->stream);
    hipLaunchKernelGGL(kin, dim3(1), dim3(CIN_TPB), 0, c->stream,
                       (const double*)p_ex, d_stats, (const double*)p_ld, (const double*)p_aux, (long long)n, kind, lambdac, threshold,
                       (double*)p_cex, (double*)p_cpk, clu_ld, clu_aux, tmp, d_label, (double*)p_dist, (long long)dist_cap,
                       d_off, d_done, c->d_err);
    (void)hipEventRecord(c->kb[SPKD_T_AHC], c->stream);
    c->kused[SPKD_T_AHC] = true;
    HIPCHK(c, hipGetLastError());
    long long done2[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(done2, d_done, sizeof done2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_label, d_label, nn * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_dist_off, d_off, (nn + 1) * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    if (dist_cap > 0)
        HIPCHK(c, hipMemcpyAsync(h_dist, p_dist, (size_t)dist_cap * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    st = end_call(c);
    *h_n_done = done2[0];
    *h_n_clusters = done2[1];
    return st;
}

spkd_status spkd_distance_rows(spkd_ctx* c, int variant, int kind, double lambdac, const double* d_stats,
                               int64_t n, int64_t row_begin, int64_t row_end, double* d_rows,
                               double* h_stat_max, double* h_stat_min) {
    if (!c || n < 0 || kind < 0 || kind > 2 || (variant != 1 && variant != 2)) return SPKD_EINVAL;
    if (row_begin < 0 || row_end < row_begin || row_end > n) return fail(c, SPKD_EINVAL, "distance_rows: bad row block");
    if (h_stat_max) *h_stat_max = std::nan("");
    if (h_stat_min) *h_stat_min = std::nan("");
    if (n == 0 || row_end == row_begin) return SPKD_OK;
    if (!d_stats || !d_rows) return fail(c, SPKD_EINVAL, "null argument");
    spkd_status st = begin_call(c);
    if (st != SPKD_OK) return st;
    const int64_t seg_off[2] = {0, n};
    AhcBuffers B;
    int64_t n_total = 0;
    std::vector<int64_t> offs;
    std::vector<int32_t> prob_of;
    MatrixPlan plan;
    plan.row_begin = row_begin;
    plan.row_end = row_end;
    if ((st = ahc_prepare(c, d_stats, seg_off, 1, variant, kind, lambdac, B, n_total, offs, prob_of, plan)) != SPKD_OK) return st;
    HIPCHK(c, hipMemcpyAsync(d_rows, B.mat + row_begin * n, (size_t)(row_end - row_begin) * n * sizeof(double),
                             hipMemcpyDeviceToDevice, c->stream));
    unsigned long long keys[2] = {0ull, ~0ull};
    HIPCHK(c, hipMemcpyAsync(&keys[0], B.smax, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&keys[1], B.smin, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    st = end_call(c);
    if (h_stat_max) *h_stat_max = key_to_double(keys[0], true);
    if (h_stat_min) *h_stat_min = key_to_double(keys[1], false);
    return st;
}

// ------------------------------------------------------------------ (4) AHC
namespace {
spkd_status ahc_impl(spkd_ctx* c, const double* d_stats, const int64_t* h_seg_off, int64_t n_prob,
                     const spkd_ahc_params* P, const MatrixPlan& plan, int32_t* h_n_merges, int32_t* h_merge_a,
                     int32_t* h_merge_b, double* h_merge_d, double* h_stat_max, double* h_stat_min) {
    if (!c || !P || n_prob < 0) return SPKD_EINVAL;
    if (n_prob == 0) return SPKD_OK;
    if (!d_stats || !h_seg_off || !h_n_merges || !h_merge_a || !h_merge_b || !h_merge_d ||
        !h_stat_max || !h_stat_min)
        return fail(c, SPKD_EINVAL, "null argument");
    if ((P->variant != 1 && P->variant != 2) || P->kind < 0 || P->kind > 2)
        return fail(c, SPKD_EINVAL, "bad variant / kind");
    for (int64_t p = 0; p < n_prob; ++p)
        if (h_seg_off[p + 1] - h_seg_off[p] < 1) return fail(c, SPKD_EINVAL, "empty clustering problem");
    spkd_status st = begin_call(c);
    if (st != SPKD_OK) return st;
    AhcBuffers B;
    int64_t n_total = 0;
    // ahc_prepare expands the records into a private working copy (merged in place)
    std::vector<int64_t> offs;
    std::vector<int32_t> prob_of;
    if ((st = ahc_prepare(c, d_stats, h_seg_off, n_prob, P->variant, P->kind, P->lambdac, B, n_total, offs, prob_of, plan)) != SPKD_OK) return st;
    int64_t n_max = 0;
    for (int64_t p = 0; p < n_prob; ++p) n_max = std::max<int64_t>(n_max, h_seg_off[p + 1] - h_seg_off[p]);
    if (n_max > AHC_MAX_N) return fail(c, SPKD_EINVAL, "clustering problem larger than 65536 records");
    // outputs + per-slot scratch
    void* op = nullptr;
    const size_t out_bytes = (size_t)n_total * (7 * sizeof(int32_t) + 3 * sizeof(double)) +
                             (size_t)n_prob * (sizeof(int32_t) + 2 * sizeof(double) + sizeof(AhcState)) + 64;
    if ((st = scratch(c, S_AHC_OUT, out_bytes, &op)) != SPKD_OK) return st;
    double* d_merge_d = (double*)op;
    double* d_tmp = d_merge_d + n_total;
    double* d_rmin = d_tmp + n_total;
    double* d_fmax = d_rmin + n_total;
    double* d_fmin = d_fmax + n_prob;
    AhcState* d_state = (AhcState*)(d_fmin + n_prob);
    int32_t* d_a = (int32_t*)(d_state + n_prob);
    int32_t* d_b = d_a + n_total;
    int32_t* d_alive = d_b + n_total;
    int32_t* d_rcache = d_alive + n_total;           // 3 ints per record: arg col, NaN col, dirty
    int32_t* d_ids = d_rcache + 3 * n_total;         // wide form: partner list per problem
    int32_t* d_n = d_ids + n_total;
    // one workgroup per problem fills the chip only when there are many problems;
    // with few, the merge loop runs as a chain of launches over all CUs instead
    int path = P->path;
    if (path != SPKD_AHC_MONO && path != SPKD_AHC_WIDE) path = n_prob <= 64 ? SPKD_AHC_WIDE : SPKD_AHC_MONO;
    const size_t lds = (size_t)(n_max + 4) * sizeof(int32_t);
    if (path == SPKD_AHC_MONO && lds > 150 * 1024) path = SPKD_AHC_WIDE;
    if (path == SPKD_AHC_MONO) {
        auto kahc = P->kind == SPKD_GLR ? k_ahc<true> : k_ahc<false>;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)kahc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        TIMED(c, SPKD_T_AHC,
              hipLaunchKernelGGL(kahc, dim3((unsigned)n_prob), dim3(AHC_TPB), lds, c->stream,
                                 B.ex, B.pk, (const int64_t*)B.seg_off, P->variant, P->kind, P->max_spk, P->lambdac,
                                 P->threshold, B.ld, B.aux, B.mat, (const int64_t*)B.mat_off, d_alive, d_tmp,
                                 d_rmin, d_rcache, d_n, d_a, d_b, d_merge_d, B.smax, B.smin, d_fmax, d_fmin, c->d_err));
    } else if (c->ahc_chain != 1 && n_max <= STEP_MAX_N) {
        // the step chain: one launch per merge, every workgroup selects for itself (spkd_cluster.hpp)
        StepArrays Q;
        void *pm = nullptr, *pe = nullptr, *pp = nullptr;
        const size_t nt = (size_t)n_total;
        const size_t misc_bytes = nt * 2 * sizeof(StepSel) + nt * sizeof(double) + nt * sizeof(unsigned long long) +
                                  (size_t)2 * n_prob * sizeof(StepState) + nt * sizeof(int32_t) + 64;
        if ((st = scratch(c, S_STEP_MISC, misc_bytes, &pm)) != SPKD_OK) return st;
        if ((st = scratch(c, S_STEP_EXM, nt * QREC * sizeof(double), &pe)) != SPKD_OK) return st;
        if ((st = scratch(c, S_STEP_PKM, nt * REC * sizeof(double), &pp)) != SPKD_OK) return st;
        Q.ex = B.ex; Q.pk = B.pk; Q.exm = (double*)pe; Q.pkm = (double*)pp;
        Q.sel2 = (StepSel*)pm;
        Q.cnt = (double*)(Q.sel2 + 2 * nt);
        Q.sw = (unsigned long long*)(Q.cnt + nt);
        Q.state2 = (StepState*)(Q.sw + nt);
        Q.death = (int32_t*)(Q.state2 + 2 * n_prob);
        Q.n_total = n_total;
        Q.n_prob = (int32_t)n_prob;
        // eight waves per workgroup once a thread of four would meet more than ~4 clusters in the
        // selection (SPKD_STEP_WAVES = 4 | 8 overrides; results do not depend on it)
        const int step_waves = c->step_waves ? c->step_waves : (n_max > STEP_WIDE_FROM ? 8 : 4);
        // partners per workgroup: seven (two of the four waves pass) while that leaves every workgroup a CU
        // of its own, else fifteen (SPKD_STEP_PARTNERS = 7 | 15 overrides; four-wave workgroups only)
        // partners of a merge per workgroup -- 3, 7 or 15: one, two or four of a workgroup's waves
        // eliminate -- chosen PER ROUND from the partners that are left, so that every workgroup
        // has a CU to itself while the chip has CUs to spare (a workgroup's record loads go through
        // one CU's address unit); SPKD_STEP_PARTNERS = 3 | 7 | 15 pins it
        const bool glr_k = P->kind == SPKD_GLR;
        using StepKernel = decltype(&k_ahc_step<false, 4, 3>);
        StepKernel kvar[3];                              // [0] 3, [1] 7, [2] 15 partners
        if (step_waves == 8) {
            kvar[0] = glr_k ? k_ahc_step<true, 8, 3> : k_ahc_step<false, 8, 3>;
            kvar[1] = glr_k ? k_ahc_step<true, 8, 7> : k_ahc_step<false, 8, 7>;
            kvar[2] = glr_k ? k_ahc_step<true, 8, STEP_PARTNERS> : k_ahc_step<false, 8, STEP_PARTNERS>;
        } else {
            kvar[0] = glr_k ? k_ahc_step<true, 4, 3> : k_ahc_step<false, 4, 3>;
            kvar[1] = glr_k ? k_ahc_step<true, 4, 7> : k_ahc_step<false, 4, 7>;
            kvar[2] = glr_k ? k_ahc_step<true, 4, STEP_PARTNERS> : k_ahc_step<false, 4, STEP_PARTNERS>;
        }
        auto sp_of_round = [&](int64_t partners) -> int {
            if (c->step_partners) return c->step_partners;
            const int64_t free_cus = 255;                // (the bookkeeper takes one)
            if (partners * n_prob <= 3 * free_cus) return 3;
            if (partners * n_prob <= 7 * free_cus) return 7;
            return STEP_PARTNERS;
        };
        const size_t nch_max = (size_t)((n_max + WAVE - 1) / WAVE);
        const size_t step_lds = ((size_t)2 * n_max + nch_max + 2) * sizeof(int32_t) + nch_max * sizeof(unsigned long long);
        if (step_lds + 20 * 1024 > 48 * 1024)
            for (int v = 0; v < 3; ++v)
                (void)hipFuncSetAttribute((const void*)kvar[v], hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds);
        (void)hipEventRecord(c->ka[SPKD_T_AHC], c->stream);
        const unsigned row_blocks = (unsigned)((n_max + AHC_WAVES - 1) / AHC_WAVES);
        hipLaunchKernelGGL(k_step_init, dim3(row_blocks, (unsigned)n_prob), dim3(AHC_TPB), 0, c->stream,
                           (const int64_t*)B.seg_off, (const double*)B.mat, (const int64_t*)B.mat_off, Q);
        for (int64_t it = 1; it < n_max; ++it) {
            const int64_t partners = n_max - it - 1;
            // (+ 1: the bookkeeper workgroup of every problem)
            const int step_sp = sp_of_round(partners);
            StepKernel kstep = kvar[step_sp == 3 ? 0 : (step_sp == 7 ? 1 : 2)];
            const unsigned blocks = (unsigned)std::max<int64_t>(1, (partners + step_sp - 1) / step_sp) + 1;
            hipLaunchKernelGGL(kstep, dim3(blocks, (unsigned)n_prob), dim3(step_waves * WAVE), step_lds, c->stream,
                               (int)it, (const int64_t*)B.seg_off, P->variant, P->kind, P->max_spk, P->lambdac,
                               P->threshold, B.ld, B.aux, B.mat, (const int64_t*)B.mat_off, Q, d_a, d_b, d_merge_d,
                               B.smax, B.smin, c->d_err);
        }
        hipLaunchKernelGGL(k_step_final, dim3((unsigned)n_prob), dim3(AHC_TPB), 0, c->stream,
                           (int)(n_max - 1), (const int64_t*)B.seg_off, (const double*)B.mat, (const int64_t*)B.mat_off,
                           Q, d_n, d_fmax, d_fmin);
        (void)hipEventRecord(c->kb[SPKD_T_AHC], c->stream);
        c->kused[SPKD_T_AHC] = true;
    } else {
        auto kround = P->kind == SPKD_GLR ? k_ahc_round<true> : k_ahc_round<false>;
        (void)hipEventRecord(c->ka[SPKD_T_AHC], c->stream);
        const unsigned row_blocks = (unsigned)((n_max + AHC_WAVES - 1) / AHC_WAVES);
        hipLaunchKernelGGL(k_ahc_init_rows, dim3(row_blocks, (unsigned)n_prob), dim3(AHC_TPB), 0, c->stream,
                           (const int64_t*)B.seg_off, (const double*)B.mat, (const int64_t*)B.mat_off, d_alive,
                           d_rmin, d_rcache);
        hipLaunchKernelGGL(k_ahc_select0, dim3((unsigned)n_prob), dim3(AHC_TPB), 0, c->stream,
                           B.ex, B.pk, (const int64_t*)B.seg_off, P->variant, P->kind, P->max_spk, P->threshold, B.aux,
                           (const double*)B.mat, (const int64_t*)B.mat_off, d_alive, d_rmin, d_rcache, d_ids,
                           d_state, d_a, d_b, d_merge_d, B.smax, B.smin, c->d_err);
        // round `it` finishes merge `it` (its distances, row caches) and selects merge it + 1;
        // after merge `it` a problem of n records has n - it clusters, i.e. n - it - 1 partners
        for (int64_t it = 1; it < n_max; ++it) {
            const int64_t partners = n_max - it - 1;
            const unsigned blocks = (unsigned)std::max<int64_t>(1, (partners + RND_PARTNERS - 1) / RND_PARTNERS);
            hipLaunchKernelGGL(kround, dim3(blocks, (unsigned)n_prob), dim3(AHC_TPB), 0, c->stream,
                               (int)it, B.ex, B.pk, (const int64_t*)B.seg_off, P->variant, P->kind, P->max_spk,
                               P->lambdac, P->threshold, B.ld, B.aux, B.mat, (const int64_t*)B.mat_off, d_alive,
                               d_rmin, d_rcache, d_ids, d_state, d_a, d_b, d_merge_d, B.smax, B.smin, c->d_err);
        }
        hipLaunchKernelGGL(k_ahc_final, dim3((unsigned)n_prob), dim3(AHC_TPB), 0, c->stream,
                           (const int64_t*)B.seg_off, (const double*)B.mat, (const int64_t*)B.mat_off,
                           (const int32_t*)d_alive, (const AhcState*)d_state, d_n, d_fmax, d_fmin);
        (void)hipEventRecord(c->kb[SPKD_T_AHC], c->stream);
        c->kused[SPKD_T_AHC] = true;
    }
    HIPCHK(c, hipGetLastError());
    std::vector<unsigned long long> kmax((size_t)n_prob), kmin((size_t)n_prob);
    std::vector<double> fmax((size_t)n_prob), fmin((size_t)n_prob);
    HIPCHK(c, hipMemcpyAsync(h_n_merges, d_n, (size_t)n_prob * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_merge_a, d_a, (size_t)n_total * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_merge_b, d_b, (size_t)n_total * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_merge_d, d_merge_d, (size_t)n_total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(kmax.data(), B.smax, (size_t)n_prob * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(kmin.data(), B.smin, (size_t)n_prob * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(fmax.data(), d_fmax, (size_t)n_prob * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(fmin.data(), d_fmin, (size_t)n_prob * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    st = end_call(c);
    for (int64_t p = 0; p < n_prob; ++p) {
        if (P->variant == 1) {
            h_stat_max[p] = key_to_double(kmax[(size_t)p], true);
            h_stat_min[p] = key_to_double(kmin[(size_t)p], false);
        } else {
            h_stat_max[p] = fmax[(size_t)p];
            h_stat_min[p] = fmin[(size_t)p];
        }
    }
    return st;
}
}  // namespace

spkd_status spkd_ahc(spkd_ctx* c, const double* d_stats, const int64_t* h_seg_off, int64_t n_prob,
                     const spkd_ahc_params* P, int32_t* h_n_merges, int32_t* h_merge_a,
                     int32_t* h_merge_b, double* h_merge_d, double* h_stat_max, double* h_stat_min) {
    return ahc_impl(c, d_stats, h_seg_off, n_prob, P, MatrixPlan(), h_n_merges, h_merge_a, h_merge_b, h_merge_d,
                    h_stat_max, h_stat_min);
}

spkd_status spkd_ahc_matrix(spkd_ctx* c, const double* d_stats, int64_t n, const spkd_ahc_params* P,
                            const double* d_matrix, double stat_max_in, double stat_min_in,
                            int32_t* h_n_merges, int32_t* h_merge_a, int32_t* h_merge_b, double* h_merge_d,
                            double* h_stat_max, double* h_stat_min) {
    if (!c || n < 1) return SPKD_EINVAL;
    if (!d_matrix) return fail(c, SPKD_EINVAL, "ahc_matrix: null matrix");
    const int64_t seg_off[2] = {0, n};
    MatrixPlan plan;
    plan.d_init = d_matrix;
    plan.init_max = stat_max_in;
    plan.init_min = stat_min_in;
    return ahc_impl(c, d_stats, seg_off, 1, P, plan, h_n_merges, h_merge_a, h_merge_b, h_merge_d, h_stat_max, h_stat_min);
}

// ------------------------------------------------------------------ (3) change detection
// A first guess, not a bound (include/spkd.h): scans of a window end that only moves forward.
int64_t spkd_gw_event_capacity(int64_t turn_len, double rate) {
    if (turn_len < 0 || !(rate >= 10.0)) return -1;
    // an outer iteration that detects advances `start` by >= 0.4*rate, one that does not grows
    // `end` by >= 0.5*rate (while winstep >= 0.2*rate; spkd_gw_event_capacity_p covers every winstep)
    return (int64_t)((double)turn_len / (0.2 * rate)) + 8;
}

int64_t spkd_gw_event_capacity_p(int64_t turn_len, const spkd_cd_params* P) {
    if (turn_len < 0 || !P || !(P->rate >= 10.0) || !(P->winstep >= 1.0)) return -1;
    // an outer iteration that detects advances `start` by maxi >= minfeas - istep = 0.4*rate;
    // one that does not grows `end` by ws, and ws is clamped to winstep from the second
    // growth of an epoch on (CD:273-284): at least min(0.5*rate, winstep) frames.  What this
    // leaves out: a detection resets `end` to start + 2*winsize (CD:264-266), the next epoch
    // regrows over frames the last one had already covered -- turns that re-scan a lot need
    // more, are told SPKD_EOVERFLOW by the kernel, and are repeated with more by the caller.
    const double step = std::min(0.2 * P->rate, std::min(0.5 * P->rate, P->winstep));
    return (int64_t)((double)turn_len / step) + 8;
}

int64_t spkd_sw_window_count(int64_t turn_len, double winsize, double winstep) {
    if (!(winstep >= 1.0) || !(winsize >= 1.0)) return -1;
    int64_t w = 0;
    for (double s = 0; s + 2 * winsize <= (double)turn_len; s += winstep) ++w;
    return w;
}

namespace {
spkd_status build_turns(spkd_ctx* c, int64_t n_frames, const int64_t* hb, const int64_t* he, int64_t n_turns,
                        const spkd_cd_params* P, const int64_t* h_off, bool gw, std::vector<TurnDesc>& turns,
                        int64_t& n_cand) {
    n_cand = 0;
    turns.resize((size_t)n_turns);
    for (int64_t t = 0; t < n_turns; ++t) {
        if (hb[t] < 0 || he[t] < hb[t] || he[t] > n_frames) return fail(c, SPKD_EINVAL, "bad turn range");
        TurnDesc& T = turns[(size_t)t];
        T.begin = hb[t];
        T.len = he[t] - hb[t];
        T.cand_off = n_cand;
        T.cand_cap = gw ? (int64_t)((double)T.len / (P->rate / 10)) + (int64_t)(2 * (P->rate / 10)) + 16 : 0;
        n_cand += T.cand_cap;
        T.ev_off = h_off[t];
        T.ev_cap = h_off[t + 1] - h_off[t];
        T.id = t;
        if (T.ev_cap < 0) return fail(c, SPKD_EINVAL, "offsets must be non-decreasing");
    }
    // longest turns first: a turn is a serial chain on one workgroup, so this is the
    // classic LPT order that keeps the tail of the launch short
    if (gw) std::stable_sort(turns.begin(), turns.end(), [](const TurnDesc& a, const TurnDesc& b) { return a.len > b.len; });
    return SPKD_OK;
}
}  // namespace

namespace {
// from this many turns on, a wave per turn (2 048 wave slots on the chip at two waves per SIMD)
constexpr int64_t GW_WAVE_PER_TURN_FROM = 4096;
constexpr int64_t GW_EIGHT_WAVES_UP_TO = 256;           // a workgroup per CU: eight waves per turn
spkd_status gw_impl(spkd_ctx* c, const float* d_frames, int64_t n_frames, const int64_t* hb,
                    const int64_t* he, int64_t n_turns, const spkd_cd_params* P, const int64_t* h_ev_off,
                    int check_capacity, int32_t* h_n_win, double* h_win_maxd, int32_t* h_win_det,
                    double* h_det_start, double* h_det_maxi, double* h_det_d, double* h_final_start,
                    double* d_seg_stats, spkd_cand_log* h_log, int64_t log_cap, int64_t* h_log_count) {
    if (!c || !P || n_turns < 0) return SPKD_EINVAL;
    if (h_log_count) *h_log_count = 0;
    if (n_turns == 0) return SPKD_OK;
    if (!d_frames || !hb || !he || !h_ev_off || !h_n_win || !h_win_maxd || !h_win_det || !h_det_start ||
        !h_det_maxi || !h_det_d || !h_final_start)
        return fail(c, SPKD_EINVAL, "null argument");
    if (P->kind < 0 || P->kind > 2) return fail(c, SPKD_EINVAL, "gw: bad kind");
    if (!(P->rate >= 10.0) || !(P->winsize >= 1.0) || !(P->winstep >= 1.0))
        return fail(c, SPKD_EINVAL, "gw: rate >= 10, winsize >= 1 frame and winstep >= 1 frame required");
    if (log_cap < 0 || (log_cap > 0 && !h_log)) return fail(c, SPKD_EINVAL, "gw: log capacity without a log buffer");
    if (!c->gw_lds_ok) return fail(c, SPKD_EHIP, "gw: the kernel's dynamic LDS size was not admitted on this device");
    std::vector<TurnDesc> turns;
    int64_t n_cand;
    spkd_status st = build_turns(c, n_frames, hb, he, n_turns, P, h_ev_off, true, turns, n_cand);
    if (st != SPKD_OK) return st;
    for (int64_t t = 0; check_capacity && t < n_turns; ++t)
        if (turns[(size_t)t].ev_cap < spkd_gw_event_capacity_p(turns[(size_t)t].len, P))
            return fail(c, SPKD_EINVAL, "gw: event capacity too small, see spkd_gw_event_capacity_p");
    if ((st = begin_call(c)) != SPKD_OK) return st;
    const int64_t n_ev = h_ev_off[n_turns];
    TurnDesc* d_turns = nullptr;
    void *d_snap = nullptr, *d_cand = nullptr, *d_i32a = nullptr, *d_i32b = nullptr, *d_d0 = nullptr, *d_d1 = nullptr,
         *d_d2 = nullptr, *d_d3 = nullptr, *d_d4 = nullptr, *d_log = nullptr;
    if ((st = upload(c, S_TURNS, turns, &d_turns)) != SPKD_OK) return st;
    // one packed record (running moment sums at the split point) per candidate slot
    if ((st = scratch(c, S_SNAP, (size_t)n_cand * REC * sizeof(double), &d_snap)) != SPKD_OK) return st;
    if ((st = scratch(c, S_CAND, (size_t)n_cand * 4 * sizeof(double), &d_cand)) != SPKD_OK) return st;
    if ((st = scratch(c, S_EV_I32A, (size_t)n_turns * sizeof(int32_t), &d_i32a)) != SPKD_OK) return st;
    if ((st = scratch(c, S_EV_I32B, (size_t)n_ev * sizeof(int32_t), &d_i32b)) != SPKD_OK) return st;
    if ((st = scratch(c, S_EV_D0, (size_t)n_ev * sizeof(double), &d_d0)) != SPKD_OK) return st;
    if ((st = scratch(c, S_EV_D1, (size_t)n_ev * sizeof(double), &d_d1)) != SPKD_OK) return st;
    if ((st = scratch(c, S_EV_D2, (size_t)n_ev * sizeof(double), &d_d2)) != SPKD_OK) return st;
    if ((st = scratch(c, S_EV_D3, (size_t)n_ev * sizeof(double), &d_d3)) != SPKD_OK) return st;
    if ((st = scratch(c, S_EV_D4, (size_t)n_turns * sizeof(double), &d_d4)) != SPKD_OK) return st;
    if ((st = scratch(c, S_LOG, (size_t)std::max<int64_t>(log_cap, 1) * sizeof(spkd_cand_log), &d_log)) != SPKD_OK) return st;
    // every pointer the kernel dereferences (a null one would be a GPU memory fault, not a status)
    if (!d_turns || !d_snap || !d_cand || !d_i32a || !d_i32b || !d_d0 || !d_d1 || !d_d2 || !d_d3 || !d_d4 ||
        !d_log || !c->d_counter || !c->d_err)
        return fail(c, SPKD_EHIP, "gw: a device scratch buffer is missing");
    HIPCHK(c, hipMemsetAsync(c->d_counter, 0, 2 * sizeof(unsigned long long), c->stream));   // [0] log entries, [1] determinants
    // A turn is a serial chain of scans.  With few turns a workgroup of four waves shares a
    // turn's matrices (latency); with thousands, ONE WAVE per turn keeps every wave of the
    // chip busy with its own chain (throughput): no wave waits at a barrier for the serial
    // phases of its turn, the SIMD's other wave belongs to another turn.  The variants give
    // bit-identical results (same sums, same eliminations, same decisions).
    int nw = c->gw_waves;
    if (nw != 1 && nw != 2 && nw != 4 && nw != 8)
        nw = n_turns >= GW_WAVE_PER_TURN_FROM ? 1 : (n_turns <= GW_EIGHT_WAVES_UP_TO ? 8 : 4);
#define SPKD_GW_LAUNCH(NW_)                                                                                     \
    hipLaunchKernelGGL(k_gw<NW_>, dim3((unsigned)n_turns), dim3(Gw<NW_>::TPB), Gw<NW_>::LDS_BYTES, c->stream, \
                       d_frames, (const TurnDesc*)d_turns, *P, (double*)d_snap, (double*)d_cand,               \
                       (int32_t*)d_i32a, (double*)d_d0, (int32_t*)d_i32b, (double*)d_d1, (double*)d_d2,        \
                       (double*)d_d3, (double*)d_d4, d_seg_stats, (spkd_cand_log*)d_log, (long long)log_cap,   \
                       c->d_counter, c->d_err)
    TIMED(c, SPKD_T_GW, if (nw == 1) SPKD_GW_LAUNCH(1); else if (nw == 2) SPKD_GW_LAUNCH(2);
                        else if (nw == 8) SPKD_GW_LAUNCH(8); else SPKD_GW_LAUNCH(4));
#undef SPKD_GW_LAUNCH
    HIPCHK(c, hipGetLastError());
    unsigned long long cnt2[2] = {0ull, 0ull};
    unsigned long long& cnt = cnt2[0];
    HIPCHK(c, hipMemcpyAsync(h_n_win, d_i32a, (size_t)n_turns * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_win_maxd, d_d0, (size_t)n_ev * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_win_det, d_i32b, (size_t)n_ev * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_det_start, d_d1, (size_t)n_ev * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_det_maxi, d_d2, (size_t)n_ev * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_det_d, d_d3, (size_t)n_ev * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_final_start, d_d4, (size_t)n_turns * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cnt2, c->d_counter, sizeof cnt2, hipMemcpyDeviceToHost, c->stream));
    st = end_call(c);
    c->last_gw_items = (int64_t)cnt2[1];
    if (h_log_count) *h_log_count = (int64_t)cnt;
    if (h_log && log_cap > 0 && cnt > 0) {
        const size_t ncopy = (size_t)std::min<int64_t>((int64_t)cnt, log_cap);
        HIPCHK(c, hipMemcpy(h_log, d_log, ncopy * sizeof(spkd_cand_log), hipMemcpyDeviceToHost));
    }
    if (st == SPKD_OK && (int64_t)cnt > log_cap && h_log)
        return fail(c, SPKD_EOVERFLOW, "candidate log too small");
    return st;
}
}  // namespace

spkd_status spkd_gw(spkd_ctx* c, const float* d_frames, int64_t n_frames, const int64_t* hb,
                    const int64_t* he, int64_t n_turns, const spkd_cd_params* P, const int64_t* h_ev_off,
                    int32_t* h_n_win, double* h_win_maxd, int32_t* h_win_det, double* h_det_start,
                    double* h_det_maxi, double* h_det_d, double* h_final_start, spkd_cand_log* h_log,
                    int64_t log_cap, int64_t* h_log_count) {
    return gw_impl(c, d_frames, n_frames, hb, he, n_turns, P, h_ev_off, 1, h_n_win, h_win_maxd, h_win_det,
                   h_det_start, h_det_maxi, h_det_d, h_final_start, nullptr, h_log, log_cap, h_log_count);
}

spkd_status spkd_gw_ex(spkd_ctx* c, const float* d_frames, int64_t n_frames, const int64_t* hb,
                       const int64_t* he, int64_t n_turns, const spkd_cd_params* P, const int64_t* h_ev_off,
                       int check_capacity, int32_t* h_n_win, double* h_win_maxd, int32_t* h_win_det,
                       double* h_det_start, double* h_det_maxi, double* h_det_d, double* h_final_start,
                       spkd_cand_log* h_log, int64_t log_cap, int64_t* h_log_count) {
    return gw_impl(c, d_frames, n_frames, hb, he, n_turns, P, h_ev_off, check_capacity, h_n_win, h_win_maxd,
                   h_win_det, h_det_start, h_det_maxi, h_det_d, h_final_start, nullptr, h_log, log_cap,
                   h_log_count);
}

spkd_status spkd_gw_fused(spkd_ctx* c, const float* d_frames, int64_t n_frames, const int64_t* hb,
                          const int64_t* he, int64_t n_turns, const spkd_cd_params* P, const int64_t* h_ev_off,
                          int check_capacity, int32_t* h_n_win, double* h_win_maxd, int32_t* h_win_det,
                          double* h_det_start, double* h_det_maxi, double* h_det_d, double* h_final_start,
                          double* d_seg_stats, spkd_cand_log* h_log, int64_t log_cap, int64_t* h_log_count) {
    if (c && !d_seg_stats) return fail(c, SPKD_EINVAL, "gw_fused: null statistics buffer");
    return gw_impl(c, d_frames, n_frames, hb, he, n_turns, P, h_ev_off, check_capacity, h_n_win, h_win_maxd,
                   h_win_det, h_det_start, h_det_maxi, h_det_d, h_final_start, d_seg_stats, h_log, log_cap,
                   h_log_count);
}

// records d_dst[dst[i]] = d_src[src[i]] (dst = NULL: i)
namespace {
__global__ __launch_bounds__(256) void k_gather_records(const double* __restrict__ src, const int64_t* __restrict__ si,
                                                        const int64_t* __restrict__ di, int64_t n,
                                                        double* __restrict__ dst) {
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    const double2* s = reinterpret_cast<const double2*>(src + si[i] * REC);
    double2* d = reinterpret_cast<double2*>(dst + (di ? di[i] : i) * REC);
    for (int e = threadIdx.x; e < REC / 2; e += 256) d[e] = s[e];
}
}  // namespace

spkd_status spkd_gather_stats(spkd_ctx* c, const double* d_src, int64_t n_src, const int64_t* h_src_index,
                              const int64_t* h_dst_index, int64_t n, int64_t n_dst, double* d_dst) {
    if (!c || n < 0) return SPKD_EINVAL;
    if (n == 0) return SPKD_OK;
    if (!d_src || !h_src_index || !d_dst) return fail(c, SPKD_EINVAL, "null argument");
    for (int64_t i = 0; i < n; ++i) {
        if (h_src_index[i] < 0 || h_src_index[i] >= n_src) return fail(c, SPKD_EINVAL, "gather: source index out of range");
        const int64_t d = h_dst_index ? h_dst_index[i] : i;
        if (d < 0 || d >= n_dst) return fail(c, SPKD_EINVAL, "gather: destination index out of range");
    }
    spkd_status st = begin_call(c);
    if (st != SPKD_OK) return st;
    std::vector<int64_t> idx(h_src_index, h_src_index + n);
    if (h_dst_index) idx.insert(idx.end(), h_dst_index, h_dst_index + n);
    int64_t* d_idx = nullptr;
    if ((st = upload(c, S_IDXA, idx, &d_idx)) != SPKD_OK) return st;
    hipLaunchKernelGGL(k_gather_records, dim3((unsigned)n), dim3(256), 0, c->stream, d_src,
                       (const int64_t*)d_idx, h_dst_index ? (const int64_t*)(d_idx + n) : (const int64_t*)nullptr,
                       n, d_dst);
    HIPCHK(c, hipGetLastError());
    return end_call(c);       // (idx must outlive the copy: end_call waits for the stream)
}

namespace {
// D[0][1] of every 2-record problem of a batch of matrices (4 doubles each) -> out[p]
__global__ __launch_bounds__(256) void k_take_pair_distance(const double* __restrict__ mat, int64_t n, double* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < n) out[p] = mat[4 * p + 1];
}
}  // namespace

// Sliding window (dist_sw, spk-change-detection.py:304-312): window w of a turn compares the
// frames [a, a + size) with [a + size, a + 2 size), a = int(w * step).  Every window is a pair
// distance between two frame sets, which is what the clustering kernels compute: the two
// statistics records of every window come from the frames (k_chunk_stats / k_reduce_sets: a
// window re-reads 2 size / step times the frames of its step, from L2 -- 9 M frame reads for
// a 1 h turn at the default 5 s / 0.5 s, under a millisecond), their log dets / KL2 vectors
// from k_cluster_prep (four records per wave) and the pooled (BIC) or within (GLR) term from
// k_matrix, every window a two-record "problem": the quad elimination and its pivoting
// fallback serve this mode too, no sliding-window kernel of its own.
spkd_status spkd_sw(spkd_ctx* c, const float* d_frames, int64_t n_frames, const int64_t* hb,
                    const int64_t* he, int64_t n_turns, const spkd_cd_params* P, const int64_t* h_d_off,
                    double* h_d) {
    if (!c || !P || n_turns < 0) return SPKD_EINVAL;
    if (n_turns == 0) return SPKD_OK;
    if (!d_frames || !hb || !he || !h_d_off || !h_d) return fail(c, SPKD_EINVAL, "null argument");
    if (P->kind < 0 || P->kind > 2) return fail(c, SPKD_EINVAL, "bad kind");
    if (!(P->winsize >= 1.0) || !(P->winstep >= 1.0)) return fail(c, SPKD_EINVAL, "sw: window and step must be >= 1 frame");
    const int64_t n_d = h_d_off[n_turns];
    std::vector<int64_t> rb, re;
    std::vector<int32_t> rs;
    rb.reserve((size_t)(2 * n_d)); re.reserve((size_t)(2 * n_d)); rs.reserve((size_t)(2 * n_d));
    const int64_t wsz = (int64_t)P->winsize;
    for (int64_t t = 0; t < n_turns; ++t) {
        if (hb[t] < 0 || he[t] < hb[t] || he[t] > n_frames) return fail(c, SPKD_EINVAL, "bad turn range");
        const int64_t len = he[t] - hb[t];
        const int64_t W = spkd_sw_window_count(len, P->winsize, P->winstep);
        if (h_d_off[t + 1] - h_d_off[t] != W) return fail(c, SPKD_EINVAL, "sw: offsets do not match spkd_sw_window_count");
        for (int64_t w = 0; w < W; ++w) {
            const int64_t a = hb[t] + (int64_t)((double)w * P->winstep);
            const int32_t s0 = (int32_t)(2 * (h_d_off[t] + w));
            rb.push_back(a); re.push_back(a + wsz); rs.push_back(s0);
            rb.push_back(a + wsz); re.push_back(a + 2 * wsz); rs.push_back(s0 + 1);
        }
    }
    if (n_d == 0) return SPKD_OK;
    if (2 * n_d > 0x7fffffffLL) return fail(c, SPKD_EINVAL, "sw: too many windows in one call");
    spkd_status st = begin_call(c);
    if (st != SPKD_OK) return st;
    void *d_rec = nullptr, *d_out = nullptr;
    if ((st = scratch(c, S_SNAP, (size_t)(2 * n_d) * REC * sizeof(double), &d_rec)) != SPKD_OK) return st;
    if ((st = scratch(c, S_EV_D0, (size_t)n_d * sizeof(double), &d_out)) != SPKD_OK) return st;
    std::vector<Chunk> chunks;
    std::vector<int64_t> set_off;
    (void)hipEventRecord(c->ka[SPKD_T_SW], c->stream);
    if ((st = set_stats_launch(c, d_frames, n_frames, rb.data(), re.data(), rs.data(), 2 * n_d, 2 * n_d, (double*)d_rec,
                               chunks, set_off)) != SPKD_OK) {
        (void)hipStreamSynchronize(c->stream);
        return st;
    }
    std::vector<int64_t> seg_off((size_t)n_d + 1);
    for (int64_t w = 0; w <= n_d; ++w) seg_off[(size_t)w] = 2 * w;
    AhcBuffers B;
    int64_t n_total = 0;
    std::vector<int64_t> offs;
    std::vector<int32_t> prob_of;
    if ((st = ahc_prepare(c, (const double*)d_rec, seg_off.data(), n_d, 1, P->kind, P->lambdac, B, n_total, offs, prob_of)) != SPKD_OK) {
        (void)hipStreamSynchronize(c->stream);
        return st;
    }
    hipLaunchKernelGGL(k_take_pair_distance, dim3((unsigned)((n_d + 255) / 256)), dim3(256), 0, c->stream,
                       (const double*)B.mat, n_d, (double*)d_out);
    (void)hipEventRecord(c->kb[SPKD_T_SW], c->stream);
    c->kused[SPKD_T_SW] = true;
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_d, d_out, (size_t)n_d * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return end_call(c);
}

// ------------------------------------------------------------------ (6) front-end
spkd_status spkd_mfcc(spkd_ctx* c, const int16_t* d_pcm, int64_t n_samples, const spkd_mfcc_params* P,
                      const float* h_melfb, const float* h_dct, const float* h_mean, const float* h_scale,
                      const float* h_transform, float* d_features, int64_t* h_n_frames) {
    if (!c || !P || !h_n_frames) return SPKD_EINVAL;
    *h_n_frames = 0;
    if (n_samples < 0 || !h_melfb || !h_dct || !h_mean || !h_scale || !h_transform)
        return fail(c, SPKD_EINVAL, "mfcc: null argument");
    if (P->window_width != MF_WIN || P->n_fft != MF_NFFT || P->n_mel != MF_MEL || P->n_cep != MF_CEP ||
        P->frame_rate <= 0 || P->sample_rate <= 0 || P->sample_rate % P->frame_rate != 0)
        return fail(c, SPKD_EINVAL, "mfcc: this build does 400-sample windows, a 512-point transform, 21 mel bins, 12 cepstra");
    if (P->cms_left < 0 || P->cms_right < 0 || P->cms_left + P->cms_right > 1024 || P->delta_width[0] < 1 ||
        P->delta_width[0] > 2 || P->delta_width[1] < 1 || P->delta_width[1] > 2 || !(P->delta_norm[0] > 0.f) ||
        !(P->delta_norm[1] > 0.f))
        return fail(c, SPKD_EINVAL, "mfcc: unsupported mean-subtraction window or delta parameters");
    const int hop = P->sample_rate / P->frame_rate;
    if (hop < 1) return fail(c, SPKD_EINVAL, "mfcc: frame rate above the sample rate");
    // everything that can be refused is refused BEFORE the first launch: an early return behind
    // one would leave the stream with work that references host memory on its way out
    const int span = MP_FR + 2 * MP_HALO;
    const size_t lds = (size_t)((span + P->cms_left + P->cms_right) * MF_STATIC + 2 * span * MF_STATIC +
                                MP_FR * MF_DIM + MF_DIM * MF_DIM) * sizeof(float);
    if (lds > 60 * 1024) return fail(c, SPKD_EINVAL, "mfcc: mean-subtraction window too wide for the LDS tile");
    const int64_t T = n_samples / hop;
    *h_n_frames = T;
    if (T == 0) return SPKD_OK;
    if (!d_pcm || !d_features) return fail(c, SPKD_EINVAL, "mfcc: null device buffer");
    spkd_status st = begin_call(c);
    if (st != SPKD_OK) return st;
    // tables: mel filterbank | dct | mean | scale | transform
    std::vector<float> tab;
    tab.insert(tab.end(), h_melfb, h_melfb + MF_MEL * MF_BINS);
    tab.insert(tab.end(), h_dct, h_dct + MF_CEP * MF_MEL);
    tab.insert(tab.end(), h_mean, h_mean + MF_DIM);
    tab.insert(tab.end(), h_scale, h_scale + MF_DIM);
    tab.insert(tab.end(), h_transform, h_transform + MF_DIM * MF_DIM);
    float* d_tab = nullptr;
    void* d_static = nullptr;
    if ((st = upload(c, S_MFCC_TAB, tab, &d_tab)) != SPKD_OK ||
        (st = scratch(c, S_MFCC_STATIC, (size_t)T * MF_STATIC * sizeof(float), &d_static)) != SPKD_OK) {
        (void)hipStreamSynchronize(c->stream);       // (the upload of `tab` may be in flight)
        return st;
    }
    const float* d_fb = d_tab;
    const float* d_dct = d_fb + MF_MEL * MF_BINS;
    const float* d_mean = d_dct + MF_CEP * MF_MEL;
    const float* d_scale = d_mean + MF_DIM;
    const float* d_tr = d_scale + MF_DIM;
    hipLaunchKernelGGL(k_mfcc_static, dim3((unsigned)((T + MF_FR - 1) / MF_FR)), dim3(MF_TPB), 0, c->stream,
                       d_pcm, (long long)n_samples, (long long)T, hop, P->pre_emph, d_fb, d_dct, (float*)d_static);
    hipLaunchKernelGGL(k_mfcc_post, dim3((unsigned)((T + MP_FR - 1) / MP_FR)), dim3(MF_TPB), lds, c->stream,
                       (const float*)d_static, (long long)T, P->cms_left, P->cms_right, P->delta_width[0],
                       P->delta_norm[0], P->delta_width[1], P->delta_norm[1], d_mean, d_scale, d_tr, d_features);
    if (hipGetLastError() != hipSuccess) {
        (void)hipStreamSynchronize(c->stream);
        return fail(c, SPKD_EHIP, "mfcc: kernel launch failed");
    }
    return end_call(c);        // (tab must outlive the upload: end_call waits for the stream)
}

// ------------------------------------------------------------------ (5) host helpers
// float('%.12g' % x) without printf for the common range: x = m 2^k exactly, so
// x 10^s (s = 11 - floor(log10 x)) is formed exactly in 128-bit integers, rounded half
// to even to a 12-digit integer q, and q / 10^s (both exact doubles) is one correctly
// rounded IEEE division -- the same value a correctly rounded strtod gives.
static const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                  1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
static const uint64_t kPow10u[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull,
                                     10000000ull, 100000000ull, 1000000000ull, 10000000000ull,
                                     100000000000ull, 1000000000000ull, 10000000000000ull,
                                     100000000000000ull, 1000000000000000ull, 10000000000000000ull,
                                     100000000000000000ull, 1000000000000000000ull,
                                     10000000000000000000ull};

static bool roundtrip_fast(double x, double* out) {
    if (!(x >= 1e-3 && x < 1e11)) return false;
    int e10 = (int)std::floor(std::log10(x));
    if (e10 < -3) e10 = -3;
    if (e10 >= 0) { if (x < kPow10[e10]) --e10; else if (e10 + 1 <= 22 && x >= kPow10[e10 + 1]) ++e10; }
    else { if (x * kPow10[-e10] < 1.0) --e10; else if (x * kPow10[-e10 - 1] >= 1.0) ++e10; }
    if (e10 < -3 || e10 > 10) return false;
    int s = 11 - e10;                              // 1 .. 14
    int k;
    const double fr = std::frexp(x, &k);            // x = fr 2^k, 0.5 <= fr < 1
    const uint64_t m = (uint64_t)std::ldexp(fr, 53);   // 53-bit integer mantissa
    k -= 53;                                        // x = m 2^k
    unsigned __int128 prod = (unsigned __int128)m * kPow10u[s];
    uint64_t q;
    if (k >= 0) {
        if (k > 20) return false;
        q = (uint64_t)(prod << k);
    } else {
        const int sh = -k;
        if (sh >= 120) return false;
        const unsigned __int128 one = 1;
        const unsigned __int128 fl = prod >> sh;
        const unsigned __int128 remn = prod & ((one << sh) - 1);
        const unsigned __int128 half = one << (sh - 1);
        q = (uint64_t)fl;
        if (remn > half || (remn == half && (q & 1))) ++q;
    }
    if (q >= 1000000000000ull) {                    // rounded up to 13 digits: one digit less
        // 10^12 exactly: value is 10^(e10+1)
        if (q != 1000000000000ull) return false;
        q = 100000000000ull;
        s -= 1;
    } else if (q < 100000000000ull) {
        return false;                               // exponent estimate off: let printf decide
    }
    *out = s >= 0 ? (double)q / kPow10[s] : (double)q * kPow10[-s];
    return true;
}

static void roundtrip_range(double* v, int64_t lo, int64_t hi) {
    char buf[64];
    for (int64_t i = lo; i < hi; ++i) {
        double r;
        if (roundtrip_fast(v[i], &r)) { v[i] = r; continue; }
        std::snprintf(buf, sizeof buf, "%.12g", v[i]);
        v[i] = std::strtod(buf, nullptr);
    }
}

void spkd_py2_roundtrip(double* v, int64_t n) {
    // correctly rounded printf / strtod, split over a few host threads for big batches
    const int64_t kMinPerThread = 8192;
    int nthreads = (int)std::min<int64_t>(8, n / kMinPerThread);
    if (nthreads <= 1) { roundtrip_range(v, 0, n); return; }
    std::vector<std::thread> pool;
    const int64_t step = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        const int64_t lo = t * step, hi = std::min<int64_t>(n, lo + step);
        if (lo < hi) pool.emplace_back(roundtrip_range, v, lo, hi);
    }
    for (auto& th : pool) th.join();
}

spkd_status spkd_labels_from_merges(int64_t n, int64_t n_merges, const int32_t* h_a, const int32_t* h_b,
                                    int32_t* h_labels) {
    if (n < 0 || n_merges < 0 || (n > 0 && !h_labels) || (n_merges > 0 && (!h_a || !h_b))) return SPKD_EINVAL;
    // ids: the shrinking list of clusters (by representative record); a merge folds
    // list entry b into entry a and closes the gap, like speakers[a].extend(speakers.pop(b))
    std::vector<int32_t> ids((size_t)n), parent((size_t)n);
    for (int64_t i = 0; i < n; ++i) ids[(size_t)i] = parent[(size_t)i] = (int32_t)i;
    int64_t m = n;
    for (int64_t k = 0; k < n_merges; ++k) {
        const int32_t a = h_a[k], b = h_b[k];
        if (a < 0 || b <= a || b >= m) return SPKD_EINVAL;
        parent[(size_t)ids[(size_t)b]] = ids[(size_t)a];
        ids.erase(ids.begin() + b);
        --m;
    }
    // final position of every surviving representative, then path-compressing finds
    std::vector<int32_t> where((size_t)n, -1);
    for (int64_t p = 0; p < m; ++p) where[(size_t)ids[(size_t)p]] = (int32_t)p;
    for (int64_t i = 0; i < n; ++i) {
        int32_t r = (int32_t)i;
        while (parent[(size_t)r] != r) r = parent[(size_t)r];
        int32_t c = (int32_t)i;
        while (parent[(size_t)c] != r) { const int32_t nx = parent[(size_t)c]; parent[(size_t)c] = r; c = nx; }
        h_labels[i] = where[(size_t)r] + 1;
    }
    return SPKD_OK;
}

spkd_status spkd_count_flags(const int32_t* h_flags, const int64_t* h_off, const int32_t* h_n, int64_t n_groups,
                             int32_t* h_out) {
    if (n_groups < 0 || (n_groups > 0 && (!h_flags || !h_off || !h_n || !h_out))) return SPKD_EINVAL;
    for (int64_t g = 0; g < n_groups; ++g) {
        if (h_n[g] < 0 || h_off[g] < 0) return SPKD_EINVAL;
        const int32_t* f = h_flags + h_off[g];
        int32_t c = 0;
        for (int32_t i = 0; i < h_n[g]; ++i) c += f[i] != 0;
        h_out[g] = c;
    }
    return SPKD_OK;
}

#pragma clang fp contract(off)
spkd_status spkd_gw_lines(int64_t n_turns, const int64_t* h_off, const int32_t* h_n_det, const double* h_det_start,
                          const double* h_det_maxi, const double* h_final_start, const double* h_turn_start_s,
                          const double* h_turn_end_s, const int64_t* h_turn_begin, const int64_t* h_turn_end,
                          double rate, int text_contract, int64_t n_lines, double* h_times, int64_t* h_frame_b,
                          int64_t* h_frame_e, int64_t* h_index, int32_t* h_line_turn) {
    if (n_turns < 0 || n_lines < 0) return SPKD_EINVAL;
    if (n_turns > 0 && (!h_off || !h_n_det || !h_det_start || !h_det_maxi || !h_final_start || !h_turn_start_s ||
                        !h_turn_end_s || !h_turn_begin || !h_turn_end || !h_times))
        return SPKD_EINVAL;
    int64_t i = 0;
    for (int64_t t = 0; t < n_turns; ++t) {
        const int64_t nd = h_n_det[t];
        if (nd < 0 || i + nd + 1 > n_lines) return SPKD_EINVAL;
        const double ls = h_turn_start_s[t], le = h_turn_end_s[t];
        for (int64_t j = 0; j <= nd; ++j, ++i) {
            const bool tail = j == nd;
            double fs, fe;                       // the line's frame positions inside the turn
            if (tail) {
                fs = h_final_start[t];
                h_times[2 * i] = fs / rate + ls;
                h_times[2 * i + 1] = ((le - ls) * rate) / rate + ls;
                fe = 0.0;
            } else {
                fs = h_det_start[h_off[t] + j];
                fe = fs + h_det_maxi[h_off[t] + j];
                h_times[2 * i] = fs / rate + ls;
                h_times[2 * i + 1] = fe / rate + ls;
            }
            if (h_frame_b) h_frame_b[i] = h_turn_begin[t] + (int64_t)fs;
            if (h_frame_e) h_frame_e[i] = tail ? h_turn_end[t] : h_turn_begin[t] + (int64_t)fe;
            if (h_index) h_index[i] = h_off[t] + j;
            if (h_line_turn) h_line_turn[i] = (int32_t)t;
        }
    }
    if (i != n_lines) return SPKD_EINVAL;
    if (text_contract) spkd_py2_roundtrip(h_times, 2 * n_lines);
    return SPKD_OK;
}

spkd_status spkd_labels_from_merges_batch(int64_t n_problems, const int64_t* h_seg_off, const int32_t* h_n_merges,
                                          const int32_t* h_a, const int32_t* h_b, int32_t* h_labels) {
    if (n_problems < 0 || (n_problems > 0 && (!h_seg_off || !h_n_merges || !h_a || !h_b || !h_labels)))
        return SPKD_EINVAL;
    std::atomic<int> bad{0};
    auto work = [&](int64_t lo, int64_t hi) {
        for (int64_t p = lo; p < hi; ++p) {
            const int64_t o = h_seg_off[p];
            if (spkd_labels_from_merges(h_seg_off[p + 1] - o, h_n_merges[p], h_a + o, h_b + o, h_labels + o) != SPKD_OK)
                bad.store(1);
        }
    };
    const int nthreads = (int)std::min<int64_t>(8, n_problems / 16);
    if (nthreads <= 1) {
        work(0, n_problems);
    } else {
        std::vector<std::thread> pool;
        const int64_t step = (n_problems + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; ++t) {
            const int64_t lo = t * step, hi = std::min<int64_t>(n_problems, lo + step);
            if (lo < hi) pool.emplace_back(work, lo, hi);
        }
        for (auto& th : pool) th.join();
    }
    return bad.load() ? SPKD_EINVAL : SPKD_OK;
}

}  // extern "C"

#ifdef SPKD_PROFILE
// profiling builds only: phase clocks of k_gw since the last call (cycles summed over
// workgroups: prefix build, scan set-up, log-det jobs, finish + arg-max; scans; turns)
extern "C" int spkd_debug_ahc_prof(unsigned long long* out4) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(spkd::g_ahc_prof), sizeof z) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(spkd::g_ahc_prof), z, sizeof z) != hipSuccess) return 1;
    return 0;
}

extern "C" int spkd_debug_step_prof(unsigned long long* out8) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(spkd::g_step_prof), sizeof z) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(spkd::g_step_prof), z, sizeof z) != hipSuccess) return 1;
    return 0;
}

extern "C" int spkd_debug_pass_prof(unsigned long long* out4) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(spkd::g_pass_prof), sizeof z) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(spkd::g_pass_prof), z, sizeof z) != hipSuccess) return 1;
    return 0;
}

extern "C" int spkd_debug_gw_prof(unsigned long long* out12) {
    unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(spkd::g_gw_prof), sizeof z) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(spkd::g_gw_prof), z, sizeof z) != hipSuccess) return 1;
    return 0;
}
#endif
