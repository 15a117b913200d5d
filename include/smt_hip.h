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
/* Per-codebook derived data of the nearest-code search (mean, centred bf16-pair split of the codes, -|k~|^2/2,
 * max |k~|^2), kept in a caller-owned persistent buffer so that it is rebuilt only when the codebook changes:
 * smt_vq_ema_apply refreshes it in the same call that rewrites the codebook; call smt_vq_prepare after any other
 * write to `codebook` (initialisation, checkpoint load). */
size_t smt_vq_prep_bytes(int k_bins, int dim);
int smt_vq_prepare(const float* codebook, int k_bins, int dim, void* prep, size_t prep_bytes, smt_stream_t stream);

/* BottleneckBlock.quantize + dequantize (models/vqvae/bottleneck.py:126-145)
 * and the per-row terms of the commit loss / fit metric (:140, :194).
 *
 *   x        [n_rows, dim]  f32, encoder output rows (NTC flattening, :92-98)
 *   codebook [k_bins, dim]  f32 (buffer `k`)
 *   prep     the buffer smt_vq_prepare / smt_vq_ema_apply filled for THIS codebook content, or NULL (then it is
 *            rebuilt inside the call, in the workspace).  It also carries the call's queue counter (zero between
 *            calls): one forward at a time per prep buffer.
 *   row_mask [n_rows]       f32 0/1 or NULL (= all ones)
 * outputs
 *   idx      [n_rows] int64  exact argmin_j ||x - k_j||^2, lowest j on ties
 *   min_dist [n_rows] f32    ||x - k_idx||^2
 *   x_d      [n_rows, dim]   k[idx] * row_mask   (may be NULL)
 *   sums     [4] f32: {sum_all min_dist, sum_masked min_dist, sum mask,
 *                      number of rows that needed exact (fp64) re-scoring}
 * dim in {32, 64, 128}; k_bins >= 1. */
size_t smt_vq_forward_workspace_bytes(int64_t n_rows, int k_bins, int dim);
int smt_vq_forward(const float* x, const float* codebook, void* prep, const float* row_mask,
                   int64_t n_rows, int k_bins, int dim,
                   int64_t* idx, float* min_dist, float* x_d, float* sums,
                   void* workspace, size_t workspace_bytes, smt_stream_t stream);

/* Backward of the straight-through estimator + commit loss
 * (bottleneck.py:194-201):  dx = dy*mask + g_commit * 2 (x - x_d) mask / (sum_mask * dim)
 *   x_d       [n_rows, dim] the x_d written by smt_vq_forward (= k[idx] on unmasked rows; masks are 0/1)
 *   dy        [n_rows, dim] grad of the (masked) quantised output, or NULL
 *   g_commit  [1] device scalar: upstream grad of the commit loss, or NULL
 *   sums      the `sums` written by smt_vq_forward (reads sums[2]) */
int smt_vq_backward(const float* x, const float* x_d, const float* row_mask, const float* dy,
                    const float* g_commit, const float* sums, int64_t n_rows, int dim, float* dx,
                    smt_stream_t stream);

/* Codebook EMA statistics, BottleneckBlock.update_k (bottleneck.py:64-68):
 * _k_sum = onehot @ x, _k_elem = onehot.sum(-1) over UNMASKED rows.
 *   stats [k_bins*dim + k_bins] f32: sums then counts; written by this call.
 * The rows are grouped by code first (counting sort, LDS-privatised histograms) and each wave sums a share of the sorted
 * order in registers: one 64-bit fixed-point integer atomic (units of 2^-24) per channel per (share, code) boundary, not per
 * row -- so the time does not depend on how skewed the code usage is, and the sums are bit-reproducible whatever order the
 * rows arrive in.  Exact for |x| < 2^15 on a 2^-24 grid (saturating beyond 2^39 units per element).  k_bins <= 16384. */
size_t smt_vq_ema_accumulate_workspace_bytes(int64_t n_rows, int k_bins, int dim);
int smt_vq_ema_accumulate(const float* x, const int64_t* idx, const float* row_mask,
                          int64_t n_rows, int k_bins, int dim, float* stats, void* workspace, size_t workspace_bytes,
                          smt_stream_t stream);

/* bottleneck.py:78-90: EMA mix, dead-code revival from k_rand, metrics; refreshes `prep` for the new codebook.
 *   stats   as above (after the cross-rank SUM, :74-75)
 *   k_rand  [k_bins, dim] revival candidates (rank 0's, :73)
 *   metrics [4] f32: {entropy, used_curr, usage, dk}
 *   prep    smt_vq_prep_bytes(k_bins, dim) bytes (its previous content is not read) */
int smt_vq_ema_apply(float* codebook, float* k_sum, float* k_elem, const float* stats,
                     const float* k_rand, float mu, float threshold, int k_bins, int dim,
                     float* metrics, void* prep, size_t prep_bytes, smt_stream_t stream);


/* --------------------------------------------------------------- losses ---- */
/* MultiNormReconstructionLoss (models/vqvae/losses.py:73-80) on [batch, t] fp32 signals, d = (y - yh) * [i < lens[b]]:
 * per clip the plain sums and the sum of the `topk` largest d^2 (exact k-th-largest threshold by radix select; the
 * reference materialises d^2 and calls torch.topk).  stats [batch][8] f32 =
 *   {sum d^2, sum |d|, sum of the topk largest d^2, threshold tau, #(d^2 > tau), #(d^2 == tau), 0, 0};
 * loss = l1 * sum_b stats[b][1] / (B t) + l2 * sum_b stats[b][0] / (B t) + linf * sum_b stats[b][2] / B (host side). */
int smt_recon_loss_fwd(const float* y, const float* yh, const int* lens, int batch, int t, int topk, float* stats,
                       smt_stream_t stream);
/* d loss / d yh given the stats of the forward and coef [3] (device) = upstream gradient times
 * {l1 / (B t), 2 l2 / (B t), 2 linf / B}. */
int smt_recon_loss_bwd(const float* y, const float* yh, const int* lens, const float* stats, const float* coef,
                       int batch, int t, int topk, float* dyh, smt_stream_t stream);

/* ------------------------------------------------------------------ MAS ---- */
/* GlowTTS monotonic alignment search, `maximum_path` (models/glow_tts/submodules.py:28-67), on the device instead of the
 * reference's numpy loop on the host: value / mask / path are [batch, t_x, t_y] fp32 (mask 0/1), max_neg_val = -inf in the
 * reference's calls.  path is the 0/1 alignment, bit-identical to the numpy result. */
int smt_maximum_path(const float* value, const float* mask, int batch, int t_x, int t_y, float max_neg_val, float* path,
                     smt_stream_t stream);

/* ------------------------------------------------------------ conv stack ---- */
/* Counter-based dropout ("dropout" spec).  The reference draws dropout masks from torch's global
 * RNG (models/vqvae/resnet.py:22,25), which no other device can reproduce; this build defines a
 * stateless generator so that train mode is parity-testable and backward can recompute the mask:
 *   keep(i) = ((fmix32((uint32)(i >> 1) * 0x9E3779B1 + key) >> (16 * (i & 1))) & 0xFFFF) >= thresh16
 * i = linear channels-last index ((b*T + t)*C + c) of the tensor being dropped, fmix32 = MurmurHash3's
 * finaliser, key = per (step seed, site) 32-bit key, thresh16 = round(p * 65536); kept values are
 * scaled by drop_scale = 1/(1-p).  oracle/vqvae_oracle.py restates it bit-exactly. */

/* Repack an fp32 torch-layout weight into the [taps][n_out][n_in] operand layout (dtype SMT_F32 /
 * SMT_BF16):  dst[tap][o][i] = src[o*stride_out + i*stride_in + tap_map[tap]*stride_tap]. */
int smt_pack_weight(const float* src, void* dst, int dtype, int n_out, int n_in, int taps,
                    int64_t stride_out, int64_t stride_in, int64_t stride_tap, const int* tap_map,
                    int swizzle, smt_stream_t stream);
/* swizzle = 1 (bf16, n_in % 128 == 0): LDS-DMA operand layout -- inside every group of 128 input channels the
 * 8-channel chunk c of row o is stored at chunk position c ^ (o & 15); pass w_swizzled = 1 with it. */

/* The same repack for MANY weights in one launch.  `table_dev` is an array of smt_pack_entry in DEVICE memory (built
 * once by the host: parameter storage is stable across optimiser steps); dst element index of (tap, o, i) is
 * dst_offset + tap*dst_tap_stride + o*dst_row_stride + i' (i' = i, or the swizzled position for swizzle = 1), so several
 * weights can be packed side by side into one operand (K1 of the four branches).  Block b of the launch handles elements
 * [1024*block_local_dev[b], +1024) of table row block_entry_dev[b]. */
typedef struct smt_pack_entry {
  const float* src; void* dst;
  int64_t stride_out, stride_in, stride_tap;
  int64_t dst_offset, dst_tap_stride, dst_row_stride;
  int dtype, n_out, n_in, taps, swizzle;
  int tap_map[16];
} smt_pack_entry;
int smt_pack_weights_batched(const smt_pack_entry* table_dev, const int* block_entry_dev, const int* block_local_dev,
                             int n_blocks, smt_stream_t stream);

/* One implicit-GEMM convolution over channels-last activations (forward, or a data gradient, which
 * is the same computation on repacked weights):
 *   y[b, t*out_stride + out_offset, co] =
 *       epi( bias[co] + sum_{j<taps} sum_ci x[b, t*stride + j*dilation - padding, ci] * w[j][co][ci] )
 *   rows t_in >= lens_in[b] of x read as 0 (the row mask of MaskedConv1d, conv.py:7-10);
 *   epi, in this order: if act_grad: times scale*[u != 0] with u = act_grad_src (the derivative of
 *        relu(dropout(.)) at an activated tensor u that this library produced); rows >= lens_out[b] -> 0;
 *        + res.
 *   if act_out: additionally y_act = relu(dropout(y)) (resnet.py:22-26) with the counter-based
 *        generator below; channel block [s*site_width, (s+1)*site_width) is dropout site s with key
 *        drop_keys[s] and index i = (b*t_y + t)*site_width + (co - s*site_width).  y may be NULL then.
 * Replaces F.conv1d / F.conv_transpose1d of models/vqvae/conv.py:5-18, resnet.py:24,27,205,217 and
 * their autograd data-gradients.  c_in % 16 == 0 (bf16) or % 8 (f32). */
typedef struct smt_conv_desc {
  int dtype;                       /* SMT_F32 | SMT_BF16: x, w, y, y_act, res, act_grad_src */
  int batch, t_in, t_out, t_y;     /* t_out: number of t computed; t_y: rows of y per batch item */
  int c_in, c_out;
  int taps, stride, dilation, padding, out_stride, out_offset;
  int act_out, act_grad, site_width;
  int ld_x, ld_y, ld_res, ld_act, ld_yact;  /* row pitches in elements */
  uint32_t drop_keys[8];
  uint32_t drop_thresh16;
  float drop_scale;
  int64_t bs_x, bs_y, bs_res, bs_act, bs_yact;  /* batch strides in elements */
  const void* x; const void* w; const float* bias; void* y; void* y_act;
  const void* res; const void* act_grad_src;
  const int* lens_in; const int* lens_out;
  int w_swizzled;                  /* w was packed with swizzle = 1: enables the LDS-DMA kernel */
  const void* zero_page;           /* >= 256 zero bytes in device memory (source of out-of-range rows), or NULL.  Since round 3
                                    * the fused streaming entries (smt_conv_k3gate_fwd, smt_conv_k1_bwd, smt_conv_gate_bwd,
                                    * smt_conv1x1_bwd, smt_conv4s2, smt_convt4s2, the K1 activated-output path) address their
                                    * operands through range-checked buffer descriptors and no longer read it; the argument
                                    * stays in their signatures (ABI 3) and must still be non-NULL where it was required. */
  /* Optional second 1x1 term folded into the same output (bf16 1x1 LDS-DMA path only, c_in == 128, c_in2 == 64):
   *   y += bias2[co] + sum_ci2 x2[b, t, ci2] * w2[co][ci2]
   * GatedHiFiBlock adds the branch input h1 = K1(x) + b1 to K3's output (resnet.py:226); K1 being linear, the
   * residual can be recomputed from x inside K3 instead of being written and read back as a 2w-wide tensor. */
  const void* x2; const void* w2;  /* x2 [B, t, c_in2] (pitch ld_x2, batch stride bs_x2); w2 [c_out][c_in2] packed, swizzle 0 */
  const float* bias2;
  const int* lens_in2;             /* rows >= lens_in2[b] of x2 read as 0 (lens_in applies to x only), or NULL */
  int c_in2, ld_x2;
  int64_t bs_x2;
  /* Dropout keys in DEVICE memory (or NULL): site s uses drop_keys_dev[s * drop_keys_dev_stride] instead of drop_keys[s].
   * A captured hipGraph freezes by-value arguments; a graphed train step keeps a table of per-site keys on the device and
   * refreshes it with smt_lm_make_keys (the same derivation as the host's) before the first convolution of every step. */
  const uint32_t* drop_keys_dev;
  int drop_keys_dev_stride;
} smt_conv_desc;
int smt_conv1d_ntc(const smt_conv_desc* desc, smt_stream_t stream);
/* Name of the kernel smt_conv1d_ntc dispatches this descriptor to ("conv_gemm", "conv_gemm_dma", "conv_ws",
 * "conv_ws_pipe", "conv1x1_dma", "conv1x1_fold", "conv_k1act"): for profilers and tests; no device work. */
const char* smt_conv1d_kernel_name(const smt_conv_desc* desc);

/* Weight (+ bias) gradient of the same convolution:
 *   dw[j][co][ci] = sum_{b,t} dy[b, t*out_stride + out_offset, co] * x[b, t*stride + j*dil - pad, ci]
 *   db[co]        = sum_{b,t} dy[b, ., co]                                   (if dbias != NULL)
 * written in fp32 to dweight[o*stride_out + i*stride_in + tap_map[tap]*stride_tap] (torch layout).
 * `desc` is the forward descriptor with y := dy (w, bias, res, act_*, y_act unused). */
size_t smt_conv1d_wgrad_workspace_bytes(const smt_conv_desc* desc);
int smt_conv1d_wgrad(const smt_conv_desc* desc, float* dweight, int64_t stride_out, int64_t stride_in,
                     int64_t stride_tap, const int* tap_map, float* dbias, void* workspace,
                     size_t workspace_bytes, smt_stream_t stream);
/* Every weight-gradient entry point (smt_conv1d_wgrad, smt_conv1x1_bwd, smt_conv_k1_bwd, smt_conv_gate_bwd) leaves
 * per-workgroup partial sums in its workspace and then reduces them in a fixed order (bitwise reproducible).  Between
 * smt_wgrad_reduce_defer(1, .) and smt_wgrad_reduce_defer(0, stream) the calls of this thread only QUEUE that reduction
 * (up to 16 per launch, job descriptors travel by value in the kernel arguments): each call must then be handed a workspace
 * of its own that stays untouched until the closing call, which launches the queued reductions on `stream`.  dweight / dbias
 * are valid once that launch has run.  Replaces autograd's per-layer weight gradients of the reference (the convolutions of
 * models/vqvae/resnet.py:205-241): one reduction launch per GatedHiFiBlock instead of ten. */
int smt_wgrad_reduce_defer(int on, smt_stream_t stream);
/* Name of the kernel smt_conv1d_wgrad runs for this descriptor ("conv_wgrad_shift", "conv_wgrad_dma", "conv_wgrad");
 * no device work. */
const char* smt_conv1d_wgrad_kernel_name(const smt_conv_desc* desc);

/* Fused backward of a bf16 128 -> 128 1x1 convolution whose forward input was u = relu(dropout(h)) (K3 of
 * GatedHiFiBlock, resnet.py:218-227): one pass over dy and u yields the masked data gradient AND the weight / bias
 * gradients.  `desc` is the 1x1 DATA-GRADIENT descriptor as for smt_conv1d_ntc (x = dy, y = dx, w = weights packed
 * [ci][co] with swizzle = 1, act_grad = 1 with act_grad_src = u, drop_scale, zero_page; no bias / res / act_out);
 * dweight[co*stride_out + ci*stride_in] = sum_t dy[t,co] * u[t,ci], dbias[co] = sum_t dy[t,co] (fp32, may be NULL). */
size_t smt_conv1x1_bwd_workspace_bytes(const smt_conv_desc* desc);
int smt_conv1x1_bwd(const smt_conv_desc* desc, float* dweight, int64_t stride_out, int64_t stride_in, float* dbias,
                    void* workspace, size_t workspace_bytes, smt_stream_t stream);

/* Fused backward of K1 of GatedHiFiBlock for all branches at once (bf16, 64 -> 512 1x1; resnet.py:205-216 and the
 * block residual resnet.py:241):  dx[t,ci] = keep(t) * sum_co dh[t,co] * W[co][ci] + res[t,ci]  (keep = t < lens[b]),
 * dweight[co*stride_out + ci*stride_in] = sum_t dh[t,co] * x[t,ci] (x rows >= lens[b] read as 0), dbias[co] = sum_t dh.
 * dh [B,t,512], x / res / dx [B,t,64] with explicit pitches; w_packed_bwd = smt_pack_weight(..) layout [ci][co],
 * swizzle 0; zero_page >= 256 zero bytes.  One pass over dh instead of a data-gradient and a weight-gradient pass. */
size_t smt_conv_k1_bwd_workspace_bytes(int batch, int t);
int smt_conv_k1_bwd(const void* dh, int64_t bs_dh, int ld_dh, const void* x, int64_t bs_x, int ld_x,
                    const void* w_packed_bwd, const void* res, int64_t bs_res, int ld_res, void* dx, int64_t bs_dx,
                    int ld_dx, const int* lens, int batch, int t, const void* zero_page, float* dweight,
                    int64_t stride_out, int64_t stride_in, float* dbias, void* workspace, size_t workspace_bytes,
                    smt_stream_t stream);

/* The k = 4 / stride 2 / padding 1 resampling convolutions at width 64 (bf16; conv.py:61-78,111-137) as HBM-streaming kernels:
 *   smt_convt4s2: y[b, 2m]   = bias + W1 x[b, m] + W3 x[b, m-1]        x [B, t_in, 64] -> y [B, 2 t_in, c_out], c_out in {64,128}
 *                 y[b, 2m+1] = bias + W2 x[b, m] + W0 x[b, m+1]        (MaskedConvTranspose1d forward; data gradient of conv4s2)
 *   smt_conv4s2 : y[b, t]    = bias + sum_j Wj x[b, 2t + j - 1]        x [B, t_in, c_in], c_in in {64,128} -> y [B, t_in/2, 64]
 *                                                                      (MaskedConv1d forward; data gradient of convt4s2)
 * w_packed = smt_pack_weight layout [tap][c_out][c_in], swizzle 0; bias fp32 or NULL; x rows >= lens_in[b] read as zero,
 * y rows >= lens_out[b] are written as zero (either may be NULL); zero_page >= 256 zero bytes. */
int smt_convt4s2(const void* x, int64_t bs_x, int ld_x, const void* w_packed, const float* bias, void* y, int64_t bs_y,
                 int ld_y, const int* lens_in, const int* lens_out, int batch, int t_in, int c_out, const void* zero_page,
                 smt_stream_t stream);
int smt_conv4s2(const void* x, int64_t bs_x, int ld_x, const void* w_packed, const float* bias, void* y, int64_t bs_y,
                int ld_y, const int* lens_in, const int* lens_out, int batch, int t_in, int c_in, const void* zero_page,
                smt_stream_t stream);

/* K3 of all four branches of a GatedHiFiBlock + the tanh * softmax gate in one pass (bf16, width 64; resnet.py:224-237):
 *   z[b,t,128 d + c] = b3[d][c] + sum_i W3_d[c][i] u2[b,t,128 d + i]  +  b1[d][c] + sum_j W1_d[c][j] x[b,t,j]     d = 0..3
 *   g[b,t,c]         = sum_d tanh(z[.., 128 d + c]) * softmax_d(z[.., 128 d + 64 + c])                             c < 64
 * u2 / z are [B,t,512], x / g [B,t,64] with explicit pitches; x rows >= lens[b] read as zero (lens may be NULL).
 * w3_packed = the four [128][128] weights stacked, smt_pack_weight layout with swizzle = 1; w1_packed = the four [128][64]
 * weights stacked, swizzle 0; b3 / b1 [4][128] fp32.  g is bit-identical to smt_conv1d_ntc (folded K3) + smt_gate_mix_fwd. */
int smt_conv_k3gate_fwd(const void* u2, int64_t bs_u2, int ld_u2, const void* x, int64_t bs_x, int ld_x,
                        const void* w3_packed, const void* w1_packed, const float* b3, const float* b1, void* z,
                        int64_t bs_z, int ld_z, void* g, int64_t bs_g, int ld_g, const int* lens, int batch, int t,
                        const void* zero_page, smt_stream_t stream);

/* Fused backward of the 64 -> 64 1x1 gate convolution that closes a GatedHiFiBlock (bf16; resnet.py:238-241):
 * dx[t,ci] = keep(t) * sum_co dy[t,co] * W[co][ci] (keep = t < lens[b]), dweight[co*stride_out + ci*stride_in] =
 * sum_t dy[t,co] * g[t,ci] (g rows >= lens[b] read as 0), dbias[co] = sum_t dy[t,co].  dy / g / dx are [B,t,64] with
 * explicit pitches; w_packed_bwd = smt_pack_weight(..) layout [ci][co], swizzle 0; zero_page >= 256 zero bytes. */
size_t smt_conv_gate_bwd_workspace_bytes(int batch, int t);
int smt_conv_gate_bwd(const void* dy, int64_t bs_dy, int ld_dy, const void* g, int64_t bs_g, int ld_g,
                      const void* w_packed_bwd, void* dx, int64_t bs_dx, int ld_dx, const int* lens, int batch, int t,
                      const void* zero_page, float* dweight, int64_t stride_out, int64_t stride_in, float* dbias,
                      void* workspace, size_t workspace_bytes, smt_stream_t stream);

/* sum_d tanh(t_d) * softmax_d(s_d) over `depth` branches laid side by side along the channel axis
 * (z[.., d*2w + c] = t_d, z[.., d*2w + w + c] = s_d) -- GatedHiFiBlock.forward, resnet.py:229-237. */
int smt_gate_mix_fwd(const void* z, void* g, int dtype, int64_t rows, int width, int depth, int ld_z, int ld_g,
                     smt_stream_t stream);
int smt_gate_mix_bwd(const void* z, const void* dg, void* dz, int dtype, int64_t rows, int width, int depth,
                     int ld_z, int ld_g, int ld_dz, smt_stream_t stream);

/* The encoder's first convolution (C_in = 1; conv.py:61 with input_emb_width = 1): x fp32 [B, t_in],
 * weight fp32 [c_out][taps], y [B, t_out, c_out] in `dtype`; rows >= lens[b] of x read as 0. */
int smt_conv_in_fwd(const float* x, const float* weight, const float* bias, const int* lens, void* y, int dtype,
                    int batch, int t_in, int t_out, int c_out, int taps, int stride, int padding,
                    smt_stream_t stream);
size_t smt_conv_in_wgrad_workspace_bytes(int batch, int t_out, int c_out);
int smt_conv_in_wgrad(const float* x, const void* dy, const int* lens, float* dweight, float* dbias, int dtype,
                      int batch, int t_in, int t_out, int c_out, int taps, int stride, int padding, void* workspace,
                      size_t workspace_bytes, smt_stream_t stream);

/* The decoder's final 1x1 projection to one channel on masked rows (encdec.py:61,82):
 * y[b,t] = bias + sum_c x[b,t,c]*mask*w[c], y fp32 [B, t].  bwd writes dx (dtype) and fp32 dw/db. */
int smt_conv_out_fwd(const void* x, const float* weight, const float* bias, const int* lens, float* y, int dtype,
                     int batch, int t, int c_in, smt_stream_t stream);
size_t smt_conv_out_bwd_workspace_bytes(int batch, int t, int c_in);
int smt_conv_out_bwd(const void* x, const float* weight, const int* lens, const float* dy, void* dx, float* dweight,
                     float* dbias, int dtype, int batch, int t, int c_in, void* workspace, size_t workspace_bytes,
                     smt_stream_t stream);

/* ------------------------------------------------------------- spectral ---- */
/* Windowed STFT magnitude (STFT.forward, datasets/transforms.py:108-123): reflect padding
 * (n_fft - hop)/2, frame every `hop`, window[n_fft] (Hann centre-padded), bins 0..n_fft/2.
 *   x [batch, t] f32;  window [n_fft] f32;  twiddle [n_fft/2] complex f32 = exp(-2 pi i m / n_fft)
 *   mag [batch, n_fft/2+1, frames] f32, frames = smt_stft_num_frames(t, n_fft, hop)
 * n_fft in {256, 512, 1024, 2048}. */
int smt_stft_num_frames(int t, int n_fft, int hop);
int smt_stft_magnitude(const float* x, const float* window, const float* twiddle, float* mag, int batch, int t,
                       int n_fft, int hop, smt_stream_t stream);
/* Fused log-mel (MelSpectrogram.forward, transforms.py:61-65): mel [batch, n_mels, frames] =
 * log(max(mel_basis @ |STFT|, 1e-5)); band[2m], band[2m+1] = first / one-past-last non-zero bin of row m. */
int smt_melspec(const float* x, const float* window, const float* twiddle, const float* mel_basis, const int* band,
                float* mel, int batch, int t, int n_fft, int hop, int n_mels, smt_stream_t stream);
/* One resolution of MultiResolutionSpectralLoss (models/vqvae/losses.py:39-55) without materialising
 * the spectra.  fwd: partial [batch, frames, 2] = per-frame sums of ((|Y|-|Yh|) m)^2 and
 * ((log|Y| - log|Yh|) m)^2 (clamp 1e-5), m = frame mask from lens (losses.py:33-37).
 * bwd: dyh [batch, t] = adjoint of dL/dYh with coef [batch, 2] = upstream * 1/(2 B sqrt(S)) per term (written, not
 * accumulated).  The per-frame gradient rows go through `workspace` and are summed per sample in a fixed order: the result
 * is bit-reproducible (no floating-point atomics). */
int smt_stft_loss_fwd(const float* y, const float* yh, const int* lens, const float* window, const float* twiddle,
                      float* partial, int batch, int t, int n_fft, int hop, smt_stream_t stream);
size_t smt_stft_loss_bwd_workspace_bytes(int batch, int t, int n_fft, int hop);
int smt_stft_loss_bwd(const float* y, const float* yh, const int* lens, const float* window, const float* twiddle,
                      const float* coef, float* dyh, int batch, int t, int n_fft, int hop, void* workspace,
                      size_t workspace_bytes, smt_stream_t stream);

/* STFT.inverse (datasets/transforms.py:125-156): magnitude, phase [batch, n_fft/2+1, frames] ->
 * out [batch, (frames - 1) * hop + n_fft - 2 * ((n_fft - hop) / 2)]: inverse real FFT per frame, synthesis window,
 * overlap-add, division by the window sum-square where it exceeds float tiny, pad_amount trimmed on both sides.
 * window / twiddle as for smt_stft_magnitude. */
int smt_stft_inverse(const float* magnitude, const float* phase, const float* window, const float* twiddle, float* out,
                     int batch, int n_fft, int hop, int frames, smt_stream_t stream);

/* Test hook (no reference counterpart): the one-wave in-LDS transform of the spectral kernels alone on ONE complex frame.
 * in / out [n_fft][2] f32, twiddle as for smt_stft_magnitude; inverse != 0 uses the conjugate twiddles (no 1/N). */
int smt_fft_selftest(const float* in, const float* twiddle, float* out, int n_fft, int inverse, smt_stream_t stream);

/* ------------------------------------------------------- TransformerLM ---- */
/* The kernels between the dense projections of the causal TransformerLM over VQ codes
 * (models/transformer_lm/transformer_lm.py:32-135: nn.TransformerEncoder, post-norm layers, ReLU feed-forward).
 * Activations are batch-major rows [batch, len, dim] f32 (the reference runs them [len, batch, dim]; the layout is
 * internal to the model).  Every dropout site uses the counter-based generator above on the element's linear index
 * in the tensor being dropped; thresh16 = 0 disables it (eval).  Every entry takes the site key twice: `drop_key` by value
 * and `drop_key_dev`, a DEVICE pointer to one key that overrides it when non-NULL -- a captured hipGraph freezes by-value
 * arguments, so a graphed step keeps its keys in device memory and refreshes them with smt_lm_make_keys. */

/* keys_dev[s] = fmix32(seed_dev[0] * 0x9E3779B1 + s * 0x7F4A7C15 + 1), s < n_sites: the per-site keys of the step whose
 * seed is in device memory (the same derivation the host uses for `drop_key`). */
int smt_lm_make_keys(const uint32_t* seed_dev, uint32_t* keys_dev, int n_sites, smt_stream_t stream);

/* out = (emb[tokens] * mul + pe[position]) * keep   (transformer_lm.py:114-116, PositionalEncoding :27-29);
 * tokens [batch, len] int64, emb [vocab_rows, dim], pe [>= len, dim].  bwd: demb [vocab_rows, dim] is zeroed and
 * accumulated; row padding_idx receives nothing (nn.Embedding(padding_idx=PAD), :42-46). */
int smt_lm_embed_fwd(const int64_t* tokens, const float* emb, const float* pe, float* out, int batch, int len, int dim,
                     float mul, uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, smt_stream_t stream);
int smt_lm_embed_bwd(const int64_t* tokens, const float* dout, float* demb, int batch, int len, int dim, int vocab_rows,
                     float mul, uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, int64_t padding_idx,
                     smt_stream_t stream);

/* Multi-head self-attention core of nn.MultiheadAttention as the reference calls it (:110-111,117: additive causal
 * mask triu(-inf, 1) plus the key-padding mask ~sequence_mask(lens)): head dim 32, len * 3 * heads * 32 < 2^31.
 *   qkv [batch, len, 3*heads*32] = (q | k | v) rows as in_proj produces them;  lens [batch] int32 or NULL
 *   ctx [batch, len, heads*32] = dropout(softmax(q k^T / sqrt(32) + masks)) v;   lse [batch, heads, len]: scratch between
 *   the forward and the backward call (log2 of the sum of 2^(score * log2 e): the kernels work in base 2)
 * key j is visible to query i iff (j <= i or causal == 0) and j < lens[b] -- causal = 0 is what `sample` runs
 * (mask=None, :142); dropout index ((b*heads + h)*len + i)*len + j.
 * bwd writes dqkv [batch, len, 3*heads*32] (every element); delta [batch, heads, len] is scratch it fills and reads. */
int smt_lm_attention_fwd(const float* qkv, const int* lens, float* ctx, float* lse, int batch, int len, int heads,
                         int causal, uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, smt_stream_t stream);
int smt_lm_attention_bwd(const float* qkv, const int* lens, const float* ctx, const float* lse, const float* dctx,
                         float* dqkv, float* delta, int batch, int len, int heads, int causal, uint32_t drop_key,
                         const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, smt_stream_t stream);

/* y = LayerNorm(x + dropout(h + h_bias)) * gamma + beta over the last dim (TransformerEncoderLayer, norm_first = False):
 * h is the bias-free output of out_proj / linear2 and h_bias [dim] that projection's bias (NULL = none), so that the
 * bias gradient falls out of this kernel's backward instead of a separate reduction; with h = NULL the plain final
 * LayerNorm of the encoder (:63-66).  stats [rows, 2] = (mean, rstd).  dim = 64 * {1,2,4,8,12,16,32}.
 * bwd: dx / dh may be NULL (not wanted); dparams [3, dim] = dgamma, dbeta, dh_bias (column sums of dh). */
int smt_lm_add_ln_fwd(const float* x, const float* h, const float* h_bias, const float* gamma, const float* beta, float* y,
                      float* stats, int64_t rows, int dim, float eps, uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16,
                      float drop_scale, smt_stream_t stream);
size_t smt_lm_add_ln_bwd_workspace_bytes(int64_t rows, int dim);
int smt_lm_add_ln_bwd(const float* x, const float* h, const float* h_bias, const float* dy, const float* gamma,
                      const float* stats, float* dx, float* dh, float* dparams, int64_t rows, int dim, uint32_t drop_key,
                      const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, void* workspace,
                      size_t workspace_bytes, smt_stream_t stream);

/* h <- dropout(relu(h + bias)) in place (linear1 -> activation -> dropout of the feed-forward).  bwd: dh = da * keep *
 * [a != 0] (a = the forward's result; dh may alias da), dbias [dim] = its column sums in a fixed order. */
int smt_lm_bias_relu_fwd(float* h, const float* bias, int64_t rows, int dim, uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16,
                         float drop_scale, smt_stream_t stream);
size_t smt_lm_bias_relu_bwd_workspace_bytes(int64_t rows, int dim);
int smt_lm_bias_relu_bwd(const float* a, const float* da, float* dh, float* dbias, int64_t rows, int dim, uint32_t drop_key,
                         const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, void* workspace,
                         size_t workspace_bytes, smt_stream_t stream);

/* Next-token cross entropy and accuracy (:121-128).  target [rows] int64, < 0 = row not counted (the reference's
 * loss_mask); row_out [rows, 2] = (logsumexp - logit[target] or 0, argmax == target as 0/1, lowest index on ties);
 * lse [rows].  bwd: dlogits = coef[0] * (softmax - onehot) on counted rows, 0 elsewhere; coef is a DEVICE scalar
 * (upstream gradient / number of counted rows). */
int smt_lm_ce_fwd(const float* logits, const int64_t* target, float* row_out, float* lse, int64_t rows, int vocab,
                  smt_stream_t stream);
int smt_lm_ce_bwd(const float* logits, const int64_t* target, const float* lse, const float* coef, float* dlogits,
                  int64_t rows, int vocab, smt_stream_t stream);

/* ------------------------------------------------------- GlowTTS ---- */
/* SURVEY 8(f4) / BASELINE.json configs[4]: the pieces of the reference's GlowTTS (models/glow_tts/glow_tts.py:59-130,
 * modules.py:134-236, submodules.py:88-512) that are not convolutions; fp32, channels-last rows [batch, t, channels] with
 * prefix row masks (lens[batch], NULL = no mask).  Row reductions are two-stage in a fixed order (reproducible).
 * Workspaces for the column reductions: smt_glow_reduce_workspace_bytes(rows, columns). */
size_t smt_glow_reduce_workspace_bytes(int64_t rows, int cols);
/* ActNorm (submodules.py:237-253): z = (bias + exp(logs) x) mask; reverse: (x - bias) exp(-logs) mask.  bwd: dx (may be NULL),
 * dlogs[c] = sum dz x exp(logs) mask, dbias[c] = sum dz mask; workspace columns = 2 * channels. */
int smt_glow_actnorm_fwd(const float* x, const float* logs, const float* bias, const int* lens, float* z, int batch, int t,
                         int channels, int reverse, smt_stream_t stream);
int smt_glow_actnorm_bwd(const float* x, const float* dz, const float* logs, const int* lens, float* dx, float* dlogs,
                         float* dbias, int batch, int t, int channels, void* workspace, size_t workspace_bytes,
                         smt_stream_t stream);
/* InvConvNear (submodules.py:292-323), n_split = 4: the 4 x 4 weight mixes the channel quadruples (h C/2 + 2 j + k);
 * transpose = 1 applies weight^T (the data gradient).  wgrad: dweight[s'][s] = sum dz[s'] x[s] over rows and groups
 * (workspace columns = 16). */
int smt_glow_invconv(const float* x, const float* weight, const int* lens, float* z, int batch, int t, int channels,
                     int transpose, smt_stream_t stream);
int smt_glow_invconv_wgrad(const float* x, const float* dz, const int* lens, float* dweight, int batch, int t, int channels,
                           void* workspace, size_t workspace_bytes, smt_stream_t stream);
/* WN gate (submodules.py:88-95 fused_add_tanh_sigmoid_multiply after the in_layer's dropout, :213-220):
 * acts[r, c] = tanh(d(a[r, c])) sigmoid(d(a[r, hidden + c])), d = counter dropout over the linear index of a [rows, 2 hidden]. */
int smt_glow_gate_fwd(const float* a, float* acts, int64_t rows, int hidden, uint32_t drop_key, const uint32_t* drop_key_dev,
                      uint32_t drop_thresh16, float drop_scale, smt_stream_t stream);
int smt_glow_gate_bwd(const float* a, const float* dacts, float* da, int64_t rows, int hidden, uint32_t drop_key,
                      const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, smt_stream_t stream);
/* Plain dropout over the linear element index: y[i] = x[i] keep(i) (DurationPredictor, submodules.py:629-633). */
int smt_glow_dropout(const float* x, float* y, int64_t n, uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16,
                     float drop_scale, smt_stream_t stream);
/* Affine coupling (submodules.py:383-405): out = (m | logs) from the `end` convolution, x = (x0 | x1):
 * z = (x0 | (m + exp(logs) x1) mask), logdet[b] = sum logs mask (NULL: not wanted; workspace >= batch * ceil(t / 64) floats);
 * reverse: z1 = (x1 - m) exp(-logs) mask.  bwd: dout = (dm | dlogs), dx = (dz0 | dz1 exp(logs) mask). */
int smt_glow_coupling_fwd(const float* out, const float* x, const int* lens, float* z, float* logdet, int batch, int t,
                          int channels, int sigmoid_scale, int reverse, void* workspace, size_t workspace_bytes,
                          smt_stream_t stream);
int smt_glow_coupling_bwd(const float* out, const float* x, const float* dz, const float* dlogdet, const int* lens,
                          float* dout, float* dx, int batch, int t, int channels, int sigmoid_scale, smt_stream_t stream);
/* Self-attention with relative-position keys and values (AttentionBlock.attention, submodules.py:463-512; window W, the
 * embeddings [2 W + 1, head_dim] shared by the heads): q, k, v, ctx [batch, t, heads * head_dim]; scores of padded queries /
 * keys are FILLED with -1e4 like the reference (a fully padded row is uniform); probs [batch, heads, t, t] = the softmax,
 * kept for the backward; dropout on the probabilities (index = linear index of probs). */
int smt_glow_attention_fwd(const float* q, const float* k, const float* v, const float* emb_rel_k, const float* emb_rel_v,
                           const int* lens, float* ctx, float* probs, int batch, int t, int heads, int head_dim, int window,
                           uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale,
                           smt_stream_t stream);
size_t smt_glow_attention_bwd_workspace_bytes(int batch, int t, int heads, int head_dim, int window);
int smt_glow_attention_bwd(const float* q, const float* k, const float* v, const float* emb_rel_k, const float* emb_rel_v,
                           const float* probs, const float* dctx, float* dq, float* dk, float* dv, float* demb_rel_k,
                           float* demb_rel_v, int batch, int t, int heads, int head_dim, int window, uint32_t drop_key,
                           const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, void* workspace,
                           size_t workspace_bytes, smt_stream_t stream);
/* Prior log-likelihood of every (token, frame) pair (glow_tts.py:87-95), the input of smt_maximum_path:
 * x_m, x_logs [batch, t_x, dim] (x_logs NULL = zeros), z [batch, t_y, dim] -> logp [batch, t_x, t_y]. */
int smt_glow_prior_logp(const float* x_m, const float* x_logs, const float* z, float* logp, int batch, int t_x, int t_y,
                        int dim, smt_stream_t stream);
/* Alignment path [batch, t_x, t_y] (0/1) -> idx [batch, t_y] (token of each frame, -1 = none), durations [batch, t_x];
 * gather: z[b, j, :] = x[b, idx[b, j], :] (glow_tts.py:100-101, the matmul with the path); scatter = its adjoint. */
int smt_glow_align_index(const float* path, int* idx, float* durations, int batch, int t_x, int t_y, smt_stream_t stream);
int smt_glow_align_gather(const float* x, const int* idx, float* z, int batch, int t_x, int t_y, int dim, smt_stream_t stream);
int smt_glow_align_scatter(const float* dz, const int* idx, float* dx, int batch, int t_x, int t_y, int dim, smt_stream_t stream);
/* MLE loss pieces (glow_tts.py:115-119): sums[0] = sum z_logs, sums[1] = sum exp(-2 z_logs) (z - z_m)^2 over n elements
 * (z_logs NULL = zeros); bwd with the DEVICE scalar coef: dz = coef exp(-2 zl)(z - zm), dz_m = -dz, dz_logs = coef (1 - ...). */
size_t smt_glow_mle_workspace_bytes(int64_t n);
int smt_glow_mle_sums(const float* z, const float* z_m, const float* z_logs, int64_t n, float* sums, void* workspace,
                      size_t workspace_bytes, smt_stream_t stream);
int smt_glow_mle_bwd(const float* z, const float* z_m, const float* z_logs, const float* coef, int64_t n, float* dz, float* dz_m,
                     float* dz_logs, smt_stream_t stream);
/* Duration loss (glow_tts.py:99, 120): diff[b, t] = (logw - log(1e-8 + durations)) mask, sum[0] = sum diff^2. */
int smt_glow_length_loss(const float* logw, const float* durations, const int* lens, int batch, int t_x, float* diff, float* sum,
                         smt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SMT_HIP_H */
