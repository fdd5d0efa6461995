/*
 * spkd.h — C ABI of libspkd_hip.so: BIC / GLR / KL2 speaker-change detection and
 * agglomerative clustering on Gaussian sufficient statistics, hand-written HIP
 * for gfx950 (MI355X).
 *
 * The reference (JianCao92/speaker-diarization) has no FFI: its hot path is
 * Python calling numpy.cov / scipy.linalg.det per distance.  Each entry point
 * below names the reference function(s) whose arithmetic it replaces, so a
 * maintainer can bind it with ctypes from the scripts of the same name (see
 * INTEGRATION.md).  Conventions:
 *   - plain C, caller-owned buffers, no global state, one context per
 *     (device, stream); a context is not re-entrant, different contexts are
 *     independent;
 *   - pointer arguments are prefixed d_ (device memory) or h_ (host memory);
 *   - every call returns an spkd_status; spkd_last_error() gives the text;
 *   - all scores are IEEE binary64; frames are binary32 exactly as feacat wrote
 *     them (spk-change-detection.py:37-41).
 *
 * Statistics record ("stats"): SPKD_REC = 820 doubles, the packed upper triangle
 * (row-major, r <= c) of the augmented second-moment matrix sum([x;1][x;1]^T)
 * of a frame set, d = 39: entry (r, c) at r*40 - r*(r-1)/2 + (c - r);
 * (r, 39) = sum x_r, (39, 39) = frame count.  Records add component-wise under
 * set union, which is what replaces the reference's np.concatenate + np.cov.
 */
#ifndef SPKD_H
#define SPKD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPKD_ABI_VERSION 2
#define SPKD_DIM 39
#define SPKD_REC 820

typedef enum {
    SPKD_OK = 0,
    SPKD_EINVAL = 1,      /* bad argument / unsupported parameter combination */
    SPKD_EHIP = 2,        /* a HIP runtime call failed */
    SPKD_ENONFINITE = 3,  /* a covariance was NaN/inf: the reference raises ValueError there */
    SPKD_EOVERFLOW = 4,   /* an output / log buffer was too small; *needed is set */
    SPKD_ENOMEM = 5
} spkd_status;

typedef enum { SPKD_BIC = 0, SPKD_GLR = 1, SPKD_KL2 = 2 } spkd_kind;

typedef struct spkd_ctx spkd_ctx;

int spkd_abi_version(void);

/* device = HIP device ordinal; stream = a hipStream_t to launch on, or NULL for a
 * stream owned by the context.  The owned stream is a blocking one: it is ordered
 * with work on the legacy default stream (handle 0, which is also what torch's
 * default stream is), not with other non-blocking streams.
 * spkd_create_on_stream always launches on the given handle, NULL included (NULL =
 * the legacy default stream): this is the call for "the caller's current torch
 * stream", whose handle is 0 unless the caller made a stream of its own. */
spkd_status spkd_create(int device, void *stream, spkd_ctx **out);
spkd_status spkd_create_on_stream(int device, void *stream, spkd_ctx **out);
void spkd_destroy(spkd_ctx *ctx);
const char *spkd_last_error(const spkd_ctx *ctx);
spkd_status spkd_sync(spkd_ctx *ctx);

/* Device memory helpers so a ctypes-only host can run without torch. */
spkd_status spkd_malloc(spkd_ctx *ctx, size_t bytes, void **d_ptr);
spkd_status spkd_free(spkd_ctx *ctx, void *d_ptr);
spkd_status spkd_memcpy_h2d(spkd_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
spkd_status spkd_memcpy_d2h(spkd_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
spkd_status spkd_memcpy_d2d(spkd_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);

/* Timing of the most recent call, measured with HIP events recorded on the
 * context's stream around each kernel launch (milliseconds).
 * which: SPKD_T_CALL = whole call; SPKD_T_<kernel> = that kernel's launch inside
 * the most recent call that used it. */
enum {
    SPKD_T_CALL = 0, SPKD_T_CHUNK_STATS, SPKD_T_REDUCE_SETS, SPKD_T_PAIR_TERMS,
    SPKD_T_CLUSTER_PREP, SPKD_T_MATRIX, SPKD_T_AHC, SPKD_T_GW, SPKD_T_SW, SPKD_N_TIMERS
};
spkd_status spkd_last_kernel_ms(spkd_ctx *ctx, int which, float *ms);

/* Work done by the most recent growing-window call (spkd_gw / spkd_gw_ex / spkd_gw_fused):
 * the number of 39x39 determinants it evaluated (one per covariance the reference's bic /
 * glr form at spk-change-detection.py:87-98, 107-115 -- left, right, pooled, within).
 * For the fp64 figure beside the HBM fraction. */
spkd_status spkd_last_gw_items(spkd_ctx *ctx, int64_t *items);

/* ---------------------------------------------------------------------------
 * (1) Sufficient statistics of frame sets.
 * Replaces get_spk_features + np.cov inputs (spk-clustering.py:46-52,88-94).
 * A set is the concatenation of one or more frame ranges [begin, end); ranges
 * of one set are contiguous in the arrays and set ids are non-decreasing.
 * d_stats receives n_sets records of SPKD_REC doubles.
 */
spkd_status spkd_set_stats(spkd_ctx *ctx, const float *d_frames, int64_t n_frames,
                           const int64_t *h_range_begin, const int64_t *h_range_end,
                           const int32_t *h_range_set, int64_t n_ranges,
                           int64_t n_sets, double *d_stats);

/* ---------------------------------------------------------------------------
 * (2) Distance terms of set pairs.
 * Replaces the bodies of bic / glr / kl2 (spk-clustering.py:81-133,
 * spk-change-detection.py:72-133) for callers that keep the decision on the
 * host (merge_rec, spk_cluster_in).  For pair p the 8 doubles at h_terms[8*p]:
 *   [0] n1  [1] n2  [2] log det S1  [3] log det S2  [4] log det S(union)
 *   [5] log det((n1 S1 + n2 S2)/N)   (only with SPKD_WANT_GLR, else NaN)
 *   [6] KL2 as coded in the reference (only with SPKD_WANT_KL2, else NaN)
 *   [7] reserved
 * S = np.cov(rowvar=0) semantics (unbiased); log det = log of the LU
 * determinant: 0 -> -inf, negative -> NaN.
 */
#define SPKD_WANT_GLR 1
#define SPKD_WANT_KL2 2
spkd_status spkd_pair_terms(spkd_ctx *ctx, const double *d_stats,
                            const int32_t *h_idx_a, const int32_t *h_idx_b,
                            int64_t n_pairs, int flags, double *h_terms);

/* Full symmetric n x n distance matrix of one kind from n records
 * (the initial double loop of spk_cluster_hi, spk-clustering.py:188-200);
 * diagonal = 2^63 (sys.maxint as a float).  d_matrix: n*n doubles. */
spkd_status spkd_distance_matrix(spkd_ctx *ctx, int kind, double lambdac,
                                 const double *d_stats, int64_t n, double *d_matrix);

/* spk_cluster_in (spk-clustering.py:136-175, spk-clustering2.py:135-170) over n statistics
 * records in recipe order, as one device-resident chain: record 0 founds cluster 0; every
 * later record is compared with every cluster so far (the cluster is the distance's first
 * argument, the record its second; a cluster's record is the sum of its members') and joins
 * the first arg-min over the finite distances if that is <= threshold, else founds a cluster.
 * kind: SPKD_BIC, SPKD_GLR or SPKD_KL2.  h_label[n]: 0-based cluster of every record.  h_dist
 * [dist_cap]: all distances in evaluation order, those of record s at h_dist_off[s] ..
 * h_dist_off[s + 1] (h_dist_off[n + 1]) -- the caller replays the reference's prints and
 * statistics from them.  *h_n_done: records processed; < n with SPKD_ENONFINITE (a
 * covariance with infs or NaNs at record *h_n_done: the reference raises there, after the
 * lines before it were written) or SPKD_EOVERFLOW (dist_cap too small: call again with more;
 * n (n - 1) / 2 always fits).  *h_n_clusters: clusters founded. */
spkd_status spkd_cluster_in(spkd_ctx *ctx, const double *d_stats, int64_t n, int kind,
                            double lambdac, double threshold, int32_t *h_label,
                            double *h_dist, int64_t dist_cap, int64_t *h_dist_off,
                            int64_t *h_n_done, int64_t *h_n_clusters);

/* Rows [row_begin, row_end) of that matrix, as spk_cluster_hi's variant `variant` fills them
 * (a block of the outer loop of spk-clustering.py:188-200 / spk-clustering2.py:178-184):
 * d_rows[(a - row_begin) * n + c] for c > a is the distance, c == a the diagonal value
 * (variant 1: 2^63, variant 2: +inf), c < a: +inf for variant 2, UNSPECIFIED for variant 1
 * (the caller mirrors the upper triangle).  One long file tiled over several GPUs
 * (SURVEY.md 8(e) row 2): every rank holds all n records, computes its block of rows, the
 * blocks are gathered, spkd_ahc_matrix runs the merge loop on the assembled matrix.
 * h_stat_max / h_stat_min (may be NULL): max / min over the finite distances of the block
 * (NaN: none) -- variant 1's running statistics start from the max / min over all blocks. */
spkd_status spkd_distance_rows(spkd_ctx *ctx, int variant, int kind, double lambdac,
                               const double *d_stats, int64_t n, int64_t row_begin, int64_t row_end,
                               double *d_rows, double *h_stat_max, double *h_stat_min);

/* ---------------------------------------------------------------------------
 * (3) Change detection.
 */
typedef struct {
    int32_t kind;        /* spkd_kind */
    int32_t trace;       /* 1: log every coarse candidate (-tt); 0: only infinite ones */
    double lambdac;      /* BIC penalty weight (-l) */
    double threshold;    /* -t */
    double winsize;      /* floor(w * rate), frames */
    double winstep;      /* floor(st * rate), frames */
    double deltaws;      /* floor(rate * dws), frames */
    double rate;         /* frames per second (-f) */
} spkd_cd_params;

typedef struct {         /* one logged candidate evaluation */
    int32_t turn;
    int32_t coarse;      /* 1 = coarse scan, 0 = fine-tune scan */
    int64_t seq;         /* (scan index << 32) | candidate index; bit 31 set = fine-tune scan */
    double start, i, d;
    int64_t n1, n2;
} spkd_cand_log;

/* Growing-window detector, dist_gw (spk-change-detection.py:180-288), all turns
 * of one feature array in one launch, one workgroup (few turns) or one wave (thousands of
 * turns) per turn -- the results are bit-identical.
 * Per turn t the events land at [h_ev_off[t], h_ev_off[t+1]) of the h_win_ and
 * h_det_ arrays; spkd_gw checks capacity per turn >= spkd_gw_event_capacity_p(len, params)
 * (a first guess, see below: SPKD_EOVERFLOW asks for more).
 * winstep < 1 frame is rejected (the reference's loop does not terminate there).
 *   h_n_win[t]           number of coarse scans (outer iterations)
 *   h_win_maxd[...]      best coarse distance of each scan (NaN: none accepted)
 *   h_win_det[...]       1 if that scan ended in a detection
 *   h_det_start/maxi/d   one triple per detection, in order
 *   h_final_start[t]     `start` when the loop ended (the tail line starts here)
 * h_log may be NULL (log_cap 0); *h_log_count receives the number of records the
 * run wanted to write (SPKD_EOVERFLOW if > log_cap).
 * Device scratch held by the context: one SPKD_REC record (6 560 bytes) of running
 * moment sums per candidate slot, about turn_len / (rate / 10) slots per turn (0.52 KB
 * per frame).
 */
/* Event capacity per turn: a FIRST GUESS, not a bound.  Both count the scans of a window
 * end that only moves forward (turn_len / step + 8; spkd_gw_event_capacity assumes winstep >=
 * 0.2 * rate, i.e. -st >= 0.2 s, spkd_gw_event_capacity_p covers every winstep: below that
 * the window grows by winstep frames per negative scan).  After every detection the
 * reference resets the window end to start + 2 * winsize (spk-change-detection.py:264-266)
 * and the window regrows over frames already scanned, so a change point found early in a
 * long-grown window can make a turn need MORE scans.  Such a call returns SPKD_EOVERFLOW
 * (nothing is written out of bounds, no result is valid); the caller repeats it with larger
 * capacities -- any capacity >= the guess is accepted (the Python host doubles until it
 * fits).  -1 for parameters spkd_gw rejects. */
int64_t spkd_gw_event_capacity(int64_t turn_len, double rate);
int64_t spkd_gw_event_capacity_p(int64_t turn_len, const spkd_cd_params *params);
spkd_status spkd_gw(spkd_ctx *ctx, const float *d_frames, int64_t n_frames,
                    const int64_t *h_turn_begin, const int64_t *h_turn_end, int64_t n_turns,
                    const spkd_cd_params *params, const int64_t *h_ev_off,
                    int32_t *h_n_win, double *h_win_maxd, int32_t *h_win_det,
                    double *h_det_start, double *h_det_maxi, double *h_det_d,
                    double *h_final_start,
                    spkd_cand_log *h_log, int64_t log_cap, int64_t *h_log_count);

/* Same as spkd_gw; with check_capacity = 0 the per-turn capacity implied by
 * h_ev_off may also be smaller than spkd_gw_event_capacity_p() -- a turn that needs more
 * makes the call return SPKD_EOVERFLOW (nothing is written out of bounds) and the
 * caller repeats it with more. */
spkd_status spkd_gw_ex(spkd_ctx *ctx, const float *d_frames, int64_t n_frames,
                       const int64_t *h_turn_begin, const int64_t *h_turn_end, int64_t n_turns,
                       const spkd_cd_params *params, const int64_t *h_ev_off, int check_capacity,
                       int32_t *h_n_win, double *h_win_maxd, int32_t *h_win_det,
                       double *h_det_start, double *h_det_maxi, double *h_det_d,
                       double *h_final_start,
                       spkd_cand_log *h_log, int64_t log_cap, int64_t *h_log_count);

/* Fused mode: spkd_gw_ex that also leaves the packed statistics
 * record of every segment it emits, so that the clustering stage does not have to read
 * the frames again.  Turn t with n_det detections writes n_det + 1 records (the tail
 * segment [final_start, turn end) last) at d_seg_stats[(h_ev_off[t] + j) * SPKD_REC];
 * d_seg_stats must hold h_ev_off[n_turns] records (only the written ones are touched),
 * and a turn needs capacity >= n_det + 1 (else SPKD_EOVERFLOW).  Record j of a turn is
 * the moment sum of the turn frames [int(det_start[j]), int(det_start[j] + det_maxi[j]))
 * -- the caller decides per segment whether that is the range its clustering stage would
 * cut (it is, unless the 12-digit time round trip of SURVEY.md A-2 moved an edge). */
spkd_status spkd_gw_fused(spkd_ctx *ctx, const float *d_frames, int64_t n_frames,
                          const int64_t *h_turn_begin, const int64_t *h_turn_end, int64_t n_turns,
                          const spkd_cd_params *params, const int64_t *h_ev_off, int check_capacity,
                          int32_t *h_n_win, double *h_win_maxd, int32_t *h_win_det,
                          double *h_det_start, double *h_det_maxi, double *h_det_d,
                          double *h_final_start, double *d_seg_stats,
                          spkd_cand_log *h_log, int64_t log_cap, int64_t *h_log_count);

/* d_dst[h_dst_index[i]] = d_src[h_src_index[i]] for i < n, whole records
 * (h_dst_index = NULL: destination i).  Compacts the sparse output of spkd_gw_fused
 * into the contiguous per-problem layout spkd_ahc reads. */
spkd_status spkd_gather_stats(spkd_ctx *ctx, const double *d_src, int64_t n_src,
                              const int64_t *h_src_index, const int64_t *h_dst_index,
                              int64_t n, int64_t n_dst, double *d_dst);

/* Sliding-window distances, the per-window part of dist_sw
 * (spk-change-detection.py:304-312): window w of turn t compares
 * [int(w*step), int(w*step+size)) with [int(w*step+size), int(w*step+2*size)).
 * h_d receives the distances of turn t at [h_d_off[t], h_d_off[t+1]);
 * the count per turn is spkd_sw_window_count(len, size, step).
 * kind = SPKD_BIC evaluates a correct two-window BIC (the reference's own
 * sliding-window BIC path crashes). */
int64_t spkd_sw_window_count(int64_t turn_len, double winsize, double winstep);
spkd_status spkd_sw(spkd_ctx *ctx, const float *d_frames, int64_t n_frames,
                    const int64_t *h_turn_begin, const int64_t *h_turn_end, int64_t n_turns,
                    const spkd_cd_params *params, const int64_t *h_d_off, double *h_d);

/* ---------------------------------------------------------------------------
 * (4) Agglomerative clustering, spk_cluster_hi
 * (variant 1: spk-clustering.py:178-240, variant 2: spk-clustering2.py:173-222).
 * n_problems independent problems (files); problem p owns the records
 * [h_seg_off[p], h_seg_off[p+1]) of d_stats (read only).  Per problem:
 *   h_n_merges[p]; merge m of problem p at index h_seg_off[p] + m of
 *   h_merge_a / h_merge_b (compacted indices at the time of the merge, a < b)
 *   and h_merge_d (the minimum that triggered it);
 *   h_stat_max[p], h_stat_min[p]: variant 1 = running max / min over every
 *   finite distance evaluated (NaN when never updated from the reference's
 *   initial 0 / maxint); variant 2 = max / min of the final matrix.
 */
/* Launch shape of the merge loop.  MONO: one workgroup per problem, the whole loop in
 * one launch (many files per call).  WIDE: every merge round is a pair of launches
 * whose log-det work spreads over all CUs (one long file).  AUTO picks by problem
 * count.  Results are identical. */
enum { SPKD_AHC_AUTO = 0, SPKD_AHC_MONO = 1, SPKD_AHC_WIDE = 2 };

typedef struct {
    int32_t variant;     /* 1 or 2 */
    int32_t kind;        /* spkd_kind */
    int32_t max_spk;     /* -ms */
    int32_t path;        /* SPKD_AHC_AUTO / _MONO / _WIDE */
    double lambdac;
    double threshold;
} spkd_ahc_params;

spkd_status spkd_ahc(spkd_ctx *ctx, const double *d_stats, const int64_t *h_seg_off,
                     int64_t n_problems, const spkd_ahc_params *params,
                     int32_t *h_n_merges, int32_t *h_merge_a, int32_t *h_merge_b,
                     double *h_merge_d, double *h_stat_max, double *h_stat_min);

/* spkd_ahc for ONE problem of n records whose initial n x n matrix the caller supplies
 * (d_matrix, device, in exactly the form spkd_ahc would have computed: see
 * spkd_distance_rows): the merge loop of spk-clustering.py:201-240 alone.  stat_max_in /
 * stat_min_in: variant 1's running max / min over the distances behind d_matrix (NaN: none);
 * ignored for variant 2.  Outputs as spkd_ahc's, problem 0. */
spkd_status spkd_ahc_matrix(spkd_ctx *ctx, const double *d_stats, int64_t n, const spkd_ahc_params *params,
                            const double *d_matrix, double stat_max_in, double stat_min_in,
                            int32_t *h_n_merges, int32_t *h_merge_a, int32_t *h_merge_b,
                            double *h_merge_d, double *h_stat_max, double *h_stat_min);

/* ---------------------------------------------------------------------------
 * (6) Feature front-end: what `feacat -c fconfig.cfg -H --raw-output x.wav` computes for the
 * reference (spk-diarization2.py:98-100; external AaltoASR C++, not in the reference tree)
 * with the module chain of fconfig.cfg:1-101: pre-emphasis, 400-sample Hamming windows at
 * 125 frames/s, magnitude spectrum, mel filterbank + log, DCT (12 cepstra) and log power,
 * mean subtraction over +-75 frames, deltas and delta-deltas, normalization, 39x39
 * transform.  PARITY UNPINNED: feacat is not available, the semantics the configuration
 * file leaves open are documented choices (oracle/mfcc_numpy.py).
 * d_pcm: n_samples 16-bit mono samples in device memory; the caller passes the tables
 * (host): mel filterbank [21][257], DCT [12][21], mean[39], scale[39], transform[39][39].
 * d_features receives floor(n_samples / hop) frames of 39 floats (*h_n_frames). */
typedef struct {
    int32_t sample_rate, frame_rate, window_width, n_fft, n_mel, n_cep;
    int32_t cms_left, cms_right;
    int32_t delta_width[2];
    float pre_emph;
    float delta_norm[2];
} spkd_mfcc_params;
spkd_status spkd_mfcc(spkd_ctx *ctx, const int16_t *d_pcm, int64_t n_samples,
                      const spkd_mfcc_params *params, const float *h_melfb, const float *h_dct,
                      const float *h_mean, const float *h_scale, const float *h_transform,
                      float *d_features, int64_t *h_n_frames);

/* ---------------------------------------------------------------------------
 * (5) Host-side helpers of the boundary (no GPU work).
 *
 * spkd_py2_roundtrip: v[i] <- float(str(v[i])) with Python-2 str() = "%.12g":
 * the value the next stage reads back from a recipe time this stage writes
 * (spk-change-detection.py:59-60 -> spk-clustering.py:16-23).
 *
 * spkd_labels_from_merges: replays a merge log of spkd_ahc
 * (speakers[a].extend(speakers[b]); speakers.pop(b), spk-clustering.py:216-217)
 * and returns for each of the n initial records its final 1-based cluster index.
 */
void spkd_py2_roundtrip(double *h_values, int64_t n);
spkd_status spkd_labels_from_merges(int64_t n, int64_t n_merges, const int32_t *h_a,
                                    const int32_t *h_b, int32_t *h_labels);
/* the same for n_problems merge logs laid out like spkd_ahc's outputs (problem p:
 * records h_seg_off[p] .. h_seg_off[p+1], merges at h_seg_off[p] + m) */
spkd_status spkd_labels_from_merges_batch(int64_t n_problems, const int64_t *h_seg_off,
                                          const int32_t *h_n_merges, const int32_t *h_a,
                                          const int32_t *h_b, int32_t *h_labels);

/* Host-side: detections per turn of an spkd_gw result -- out[t] = number of non-zero flags
 * among h_flags[h_off[t] .. h_off[t] + h_n[t]) (the h_win_det entries of turn t's h_n_win[t]
 * windows; what lies behind them in a reused buffer is not looked at). */
spkd_status spkd_count_flags(const int32_t *h_flags, const int64_t *h_off, const int32_t *h_n,
                             int64_t n_groups, int32_t *h_out);

/* Host-side: the recipe lines of an spkd_gw result in recipe order -- per turn its h_n_det[t]
 * detections (detection j: [start, start + maxi) of the turn, event slot h_off[t] + j), then
 * the tail line [final start, turn end) -- as the change-detection script writes them
 * (spk-change-detection.py:374-392): h_times[2 i], h_times[2 i + 1] = start / end in seconds,
 * start_s + frames / rate in the reference's operation order, passed through
 * spkd_py2_roundtrip when text_contract is non-zero.  Optional (NULL to skip): the absolute
 * frame range [h_frame_b[i], h_frame_e[i]) whose statistics the fused detector left for the
 * line, the event slot h_index[i] of that record, the turn h_line_turn[i] of the line.
 * n_lines must be n_turns + the sum of h_n_det. */
spkd_status spkd_gw_lines(int64_t n_turns, const int64_t *h_off, const int32_t *h_n_det,
                          const double *h_det_start, const double *h_det_maxi,
                          const double *h_final_start, const double *h_turn_start_s,
                          const double *h_turn_end_s, const int64_t *h_turn_begin,
                          const int64_t *h_turn_end, double rate, int text_contract,
                          int64_t n_lines, double *h_times, int64_t *h_frame_b,
                          int64_t *h_frame_e, int64_t *h_index, int32_t *h_line_turn);

#ifdef __cplusplus
}
#endif
#endif /* SPKD_H */
