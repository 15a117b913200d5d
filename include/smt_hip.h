/*
 * smt_hip.h -- C ABI of libsmt_hip.so, the MI355X (gfx950) kernel library for the
 * VQ-VAE train-step hot path of vliu15/speech-masters-thesis.
 *
 * The reference is pure Python/PyTorch and has NO native/FFI interface
 * (SURVEY.md 8(b)); each entry point therefore cites the reference *Python
 * expression* it replaces (file:line relative to the reference root).  The
 * binding a maintainer would add is a ctypes stub: see INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - plain C types only: device pointers, sizes, a hipStream_t passed as void*;
 *   - return 0 on success, non-zero on error; smt_last_error() gives the
 *     thread-local message of the last failing call;
 *   - never allocate or free caller-visible memory: scratch is passed in, sized
 *     by the matching *_workspace_bytes() query;
 *   - asynchronous on `stream`; no host synchronisation inside (graph-capturable);
 *   - activations are channels-last: a [B, C, T] reference tensor is held as
 *     [B, T, C] ("NTC") so that one time step's channels are contiguous.
 */
#ifndef SMT_HIP_H
#define SMT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* smt_stream_t; /* hipStream_t */

/* dtype tags for activation / weight buffers of the conv stack */
enum { SMT_F32 = 0, SMT_BF16 = 1 };

const char* smt_last_error(void);
int smt_abi_version(void);

/* ------------------------------------------------------------------ VQ ---- */
/* BottleneckBlock.quantize + dequantize (models/vqvae/bottleneck.py:126-145)
 * and the per-row terms of the commit loss / fit metric (:140, :194).
 *
 *   x        [n_rows, dim]  f32, encoder output rows (NTC flattening, :92-98)
 *   codebook [k_bins, dim]  f32 (buffer `k`)
 *   row_mask [n_rows]       f32 0/1 or NULL (= all ones)
 * outputs
 *   idx      [n_rows] int64  exact argmin_j ||x - k_j||^2, lowest j on ties
 *   min_dist [n_rows] f32    ||x - k_idx||^2
 *   x_d      [n_rows, dim]   k[idx] * row_mask   (may be NULL)
 *   sums     [4] f32: {sum_all min_dist, sum_masked min_dist, sum mask,
 *                      number of rows that needed fp64 re-scoring}
 * dim in {32, 64, 128}; k_bins >= 1. */
size_t smt_vq_forward_workspace_bytes(int64_t n_rows, int k_bins, int dim);
int smt_vq_forward(const float* x, const float* codebook, const float* row_mask,
                   int64_t n_rows, int k_bins, int dim,
                   int64_t* idx, float* min_dist, float* x_d, float* sums,
                   void* workspace, size_t workspace_bytes, smt_stream_t stream);

/* Backward of the straight-through estimator + commit loss
 * (bottleneck.py:194-201):  dx = dy*mask*st_scale + g_commit * 2 (x - x_d) mask / (sum_mask * dim)
 *   dy        [n_rows, dim] grad of the (masked) quantised output, or NULL
 *   g_commit  [1] device scalar: upstream grad of the commit loss, or NULL
 *   sums      the `sums` written by smt_vq_forward (reads sums[2]) */
int smt_vq_backward(const float* x, const float* x_d_unmasked_codebook, const int64_t* idx,
                    const float* row_mask, const float* dy, const float* g_commit, const float* sums,
                    int64_t n_rows, int dim, float* dx, smt_stream_t stream);

/* Codebook EMA statistics, BottleneckBlock.update_k (bottleneck.py:64-68):
 * _k_sum = onehot @ x, _k_elem = onehot.sum(-1) over UNMASKED rows.
 *   stats [k_bins*dim + k_bins] f32: sums then counts; zeroed by this call. */
int smt_vq_ema_accumulate(const float* x, const int64_t* idx, const float* row_mask,
                          int64_t n_rows, int k_bins, int dim, float* stats, smt_stream_t stream);

/* bottleneck.py:78-90: EMA mix, dead-code revival from k_rand, metrics.
 *   stats   as above (after the cross-rank SUM, :74-75)
 *   k_rand  [k_bins, dim] revival candidates (rank 0's, :73)
 *   metrics [4] f32: {entropy, used_curr, usage, dk} */
int smt_vq_ema_apply(float* codebook, float* k_sum, float* k_elem, const float* stats,
                     const float* k_rand, float mu, float threshold, int k_bins, int dim,
                     float* metrics, smt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SMT_HIP_H */
