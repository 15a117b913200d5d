// Kernels of the causal TransformerLM over VQ codes (reference models/transformer_lm/transformer_lm.py:32-135, built there
// from torch.nn.TransformerEncoder: post-norm layers, ReLU feed-forward, dropout after the embedding, on the attention
// weights, on both sub-layer outputs and inside the feed-forward).  The five dense projections of a layer stay library
// GEMMs; everything between them is fused here so that a layer is 5 GEMMs + 4 launches instead of ~25 eager ops:
//
//   lm_embed        tokens -> emb[token] * sqrt(d) + pe[pos], dropout                          (transformer_lm.py:114-116, :27-29)
//   lm_attention    softmax(Q K^T / sqrt(dh) + causal + key-padding mask), dropout, . V        (nn.MultiheadAttention)
//   lm_add_ln       LayerNorm(x + dropout(h))                                                   (TransformerEncoderLayer, post-norm)
//   lm_bias_relu    dropout(relu(h + b))  in place                                              (linear1 -> activation -> dropout)
//   lm_ce           masked mean cross-entropy + accuracy over the next-token logits              (transformer_lm.py:121-128)
//
// Activations are [B, L, C] fp32 rows (the fp32 parity path; head dim 32, any L with L * 3 * heads * 32 < 2^31).  Dropout masks come from the
// counter-based generator of include/smt_hip.h ("dropout"): keep(i) of the element's linear index under a per-site key,
// so the backward kernels recompute them.
#include <algorithm>

#include "conv_common.h"

namespace smt {

__device__ __forceinline__ float lm_keep(unsigned long long i, unsigned key, unsigned thresh16, float scale) {
  return thresh16 == 0 ? 1.f : (drop_keep(i, key, thresh16) ? scale : 0.f);
}

// ------------------------------------------------------------------------------------------------ embedding
__global__ __launch_bounds__(256) void lm_embed_fwd_kernel(const long long* __restrict__ tok, const float* __restrict__ emb,
                                                           const float* __restrict__ pe, float* __restrict__ out, int B, int L,
                                                           int D, float mul, unsigned key, const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;                                // device-resident key (graph replay): overrides the by-value one
  const long long total = (long long)B * L * D;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e % D);
    const long long row = e / D;
    const int pos = (int)(row % L);
    const float v = emb[tok[row] * D + c] * mul + pe[(long long)pos * D + c];
    out[e] = v * lm_keep((unsigned long long)e, key, thr, dscale);
  }
}
// dW[token] += dout * mask * mul (the padding row 0 gets no gradient: nn.Embedding(padding_idx = 0))
__global__ __launch_bounds__(256) void lm_embed_bwd_kernel(const long long* __restrict__ tok, const float* __restrict__ dout,
                                                           float* __restrict__ demb, int B, int L, int D, float mul,
                                                           unsigned key, const unsigned* __restrict__ key_dev, unsigned thr, float dscale, long long pad_idx) {
  if (key_dev) key = *key_dev;                                // device-resident key (graph replay): overrides the by-value one
  const long long total = (long long)B * L * D;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long t = tok[e / D];
    if (t == pad_idx) continue;
    atomicAdd(demb + t * D + (e % D), dout[e] * lm_keep((unsigned long long)e, key, thr, dscale) * mul);
  }
}

// ------------------------------------------------------------------------------------------------ attention
// qkv [B, L, 3 d] (q | k | v, heads side by side inside each), ctx [B, L, d], lse [B, H, L] (log-sum-exp of the scaled,
// masked scores).  Key j is visible to query i iff (j <= i or not causal) and j < lens[b].  Attention-weight dropout:
// element index ((b H + h) L + i) L + j.  Head dim 32.  (Two earlier generations of these kernels ran on the VALU: K / V
// staged in LDS with four lanes per query row was bound by the 128 B/clk LDS pipe -- every lane pulling the same 256 bytes
// per key -- at 48 / 50 / 58 us for forward / dq / dkv at 8 x 16 x 258; broadcasting the rows through scalar loads instead
// was bound by the scalar cache's miss latency, 70 / 50 / 63 us.)
constexpr int LM_DH = 32;
// Longest sequence: the kernels walk the keys / queries in 32-row blocks (online softmax), so the only limit is the 32-bit
// element offset inside one batch item's qkv block, len * 3 * heads * 32 < 2^31 (the reference's max_len is 5000).
static inline bool lm_len_ok(int len, int heads) { return (long long)len * 3 * heads * LM_DH < (1ll << 31); }

// The attention kernels run on the f32-input matrix pipe (v_mfma_f32_32x32x2_f32: exact f32 products and sums at the
// VALU's FLOP rate, but one operand register per 32 FMAs instead of one LDS read per FMA).  A workgroup owns 32 rows of
// one (batch, head) -- queries in the forward and the dq kernel, keys in the dk/dv kernel -- and its four waves take
// every fourth 32-row block of the opposite index; their partial results meet in LDS at the end.  Everything is
// transposed so that the owned row is the accumulator COLUMN (= the lane): S^T = K Q^T, softmax statistics per lane,
// ctx^T += V^T P^T with P^T taken straight from the accumulator registers (its key order, rows (i & 3) + 8 (i >> 2) +
// 4 (lane >> 5), is matched by the order V's rows are fetched in).  Operands come straight from global memory
// (L2-resident: a head's K and V are 66 KB): no LDS in the loops, no barriers but the final merge; with up to four waves per
// SIMD the loads of one wave hide behind the MFMAs of the others (register double-buffering them measured slower).  Two fragment shapes:
// "row" = 16 consecutive channels [16 hh, 16 hh + 16) of row `col` (the contraction runs over channels), "column" =
// channel `col` of the 16 rows at_acc_row(t, hh) (the contraction runs over rows).  Offsets are 32-bit element counts
// from a wave-uniform base (64-bit per-lane address arithmetic cost more than the MFMAs in a first version).
typedef float f32x16v __attribute__((ext_vector_type(16)));
constexpr int AT_W = 4;
__device__ __forceinline__ int at_acc_row(int i, int hh) { return (i & 3) + 8 * (i >> 2) + 4 * hh; }
__device__ __forceinline__ void at_row_frag(float (&r)[16], const float* __restrict__ mat, unsigned pitch, int row, int last, int hh,
                                            float mul) {
  const float* src = mat + ((unsigned)min(row, last) * pitch + 16u * hh);
#pragma unroll
  for (int c = 0; c < 16; c += 4) {
    const f32x4 v = *(const f32x4*)(src + c);
    r[c] = v.x * mul; r[c + 1] = v.y * mul; r[c + 2] = v.z * mul; r[c + 3] = v.w * mul;
  }
}
__device__ __forceinline__ void at_col_frag(float (&r)[16], const float* __restrict__ mat, unsigned pitch, int row0, int last, int col,
                                            int hh) {
  const int first = row0 + 4 * hh;
  const unsigned base_off = (unsigned)first * pitch + col, last_off = (unsigned)last * pitch + col;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int k = (t & 3) + 8 * (t >> 2);                     // compile time: k * pitch is a scalar
    r[t] = mat[first + k <= last ? base_off + (unsigned)k * pitch : last_off];
  }
}
__device__ __forceinline__ f32x16v at_mfma16(const float (&a)[16], const float (&b)[16], f32x16v c) {
#pragma unroll
  for (int t = 0; t < 16; ++t) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], c, 0, 0, 0);
  return c;
}
__device__ __forceinline__ f32x16v at_zero16() {
  f32x16v z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
// Sum the four waves' transposed accumulators (each scaled by its own `scale`) through LDS red [4][16][64]; wave w then
// stores channel group w -- channels 8 w + 4 hh + {0..3} -- of the row this lane's column stands for, times mul.
__device__ __forceinline__ void at_merge_store(float* red, const f32x16v& o, float scale, int w, int lane, int hh, float mul, bool live,
                                               float* __restrict__ dst) {
#pragma unroll
  for (int i = 0; i < 16; ++i) red[(w * 16 + i) * 64 + lane] = o[i] * scale;
  __syncthreads();
  f32x4 out;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = 4 * w + k;
    out[k] = (((red[i * 64 + lane] + red[(16 + i) * 64 + lane]) + red[(32 + i) * 64 + lane]) + red[(48 + i) * 64 + lane]) * mul;
  }
  if (live) *(f32x4*)(dst + 8 * w + 4 * hh) = out;
  __syncthreads();
}

// Dropout factors of the 16 accumulator elements of a lane whose element i stands for linear index e_first + at_acc_row(i, hh)
// (consecutive in groups of four).  Two consecutive indices 2 j, 2 j + 1 share one hash (smt_hip.h "dropout"), so when
// e_first is even -- always, if the row length L is even -- one hash serves two elements: 8 hashes instead of 16.
__device__ __forceinline__ void at_keep16(float (&kf)[16], unsigned long long e_first, int hh, unsigned key, unsigned thr,
                                          float dscale) {
  if (thr == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = 1.f;
  } else if ((e_first & 1ull) == 0) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const unsigned h = fmix32((unsigned)((e_first + at_acc_row(i, hh)) >> 1) * 0x9E3779B1u + key);
      kf[i] = (h & 0xFFFFu) >= thr ? dscale : 0.f;
      kf[i + 1] = (h >> 16) >= thr ? dscale : 0.f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = drop_keep(e_first + at_acc_row(i, hh), key, thr) ? dscale : 0.f;
  }
}
constexpr float LM_LOG2E = 1.4426950408889634f;

__global__ __launch_bounds__(256) void lm_attn_fwd_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                          float* __restrict__ ctx, float* __restrict__ lse, int L, int H,
                                                          int causal, unsigned key, const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;                                // device-resident key (graph replay): overrides the by-value one
  __shared__ float red[AT_W * 16 * 64];
  __shared__ float red_m[AT_W][64], red_z[AT_W][64];
  const int qb = gridDim.x - 1 - blockIdx.x;                  // longest (last) query blocks first
  const int b = blockIdx.y / H, h = blockIdx.y % H, d = H * LM_DH;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, col = lane & 31, hh = lane >> 5;
  const unsigned pitch = 3u * d;
  const float* qm = qkv + (size_t)b * L * pitch + h * LM_DH;
  const float* km = qm + d;
  const float* vm = qm + 2 * d;
  const int len = lens ? max(0, min(lens[b], L)) : L;
  const int q0 = qb * 32, qi = q0 + col, qc = min(qi, L - 1);
  const int kend = causal ? min(len, min(L, q0 + 32)) : len;  // keys any query of this block can see
  float qf[16];
  at_row_frag(qf, qm, pitch, qi, L - 1, hh, rsqrtf((float)LM_DH) * LM_LOG2E);   // scores in base-2 units: exp2 is ONE instruction
  f32x16v o = at_zero16();
  float m = -INFINITY, z = 0.f;
  const unsigned long long e0 = (((unsigned long long)b * H + h) * L + qc) * L;
  for (int k0 = 32 * w; k0 < kend; k0 += 32 * AT_W) {
    float kf[16], vf[16];
    at_row_frag(kf, km, pitch, k0 + col, L - 1, hh, 1.f);
    at_col_frag(vf, vm, pitch, k0, L - 1, col, hh);
    f32x16v st = at_mfma16(kf, qf, at_zero16());
    float mloc = -INFINITY;
    if (k0 + 32 <= kend && (!causal || k0 + 31 <= q0)) {      // uniform: every key of the block visible to every query
#pragma unroll
      for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, st[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kj = k0 + at_acc_row(i, hh);
        st[i] = (kj < kend && (!causal || kj <= qi)) ? st[i] : -INFINITY;
        mloc = fmaxf(mloc, st[i]);
      }
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mn = fmaxf(m, mloc), ms = (mn == -INFINITY) ? 0.f : mn;
    const float corr = __builtin_amdgcn_exp2f(m - ms);        // m = -inf: 0
    z *= corr;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] *= corr;
    m = mn;
    float pk[16];
    at_keep16(pk, e0 + k0, hh, key, thr, dscale);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __builtin_amdgcn_exp2f(st[i] - ms);     // masked: exp2(-inf) = 0
      z += p;
      pk[i] *= p;
    }
    o = at_mfma16(vf, pk, o);
  }
  // merge the four waves' partial softmaxes (m is per query = the same in both lane halves; z is a per-half partial)
  red_m[w][lane] = m; red_z[w][lane] = z;
  __syncthreads();
  const float mm = fmaxf(fmaxf(red_m[0][lane], red_m[1][lane]), fmaxf(red_m[2][lane], red_m[3][lane]));
  float zz = 0.f;
#pragma unroll
  for (int k = 0; k < AT_W; ++k)
    zz += red_m[k][lane] == -INFINITY ? 0.f : red_z[k][lane] * __builtin_amdgcn_exp2f(red_m[k][lane] - mm);
  zz += __shfl_xor(zz, 32, 64);
  const bool any = mm != -INFINITY;                           // at least one visible key
  at_merge_store(red, o, (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mm), w, lane, hh, any ? 1.f / zz : 0.f, qi < L,
                 ctx + ((size_t)b * L + qc) * d + h * LM_DH);
  // lse is kept in BASE-2 units (log2 of the sum of 2^score): it is scratch between this kernel and the backward ones
  if (w == 0 && hh == 0 && qi < L) lse[((size_t)b * H + h) * L + qi] = any ? mm + __builtin_amdgcn_logf(zz) : 0.f;
}

// Backward: P_ij = exp(s_ij - lse_i); dPd_ij = dctx_i . v_j; delta_i = sum_j P_ij keep_ij dPd_ij = dctx_i . ctx_i;
// dS_ij = P_ij (keep_ij dPd_ij - delta_i); dq_i = sum_j dS_ij k_j / sqrt(dh); dk_j = sum_i dS_ij q_i / sqrt(dh);
// dv_j = sum_i P_ij keep_ij dctx_i.
// dq: 32 queries per workgroup (the accumulator column): S^T = K Q^T, dPd^T = V dctx^T, dq^T += K^T dS^T; leaves delta.
__global__ __launch_bounds__(256) void lm_attn_dq_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                         const float* __restrict__ ctx, const float* __restrict__ lse,
                                                         const float* __restrict__ dctx, float* __restrict__ dqkv,
                                                         float* __restrict__ delta, int L, int H, int causal, unsigned key,
                                                         const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;                                // device-resident key (graph replay): overrides the by-value one
  __shared__ float red[AT_W * 16 * 64];
  const int qb = gridDim.x - 1 - blockIdx.x;
  const int b = blockIdx.y / H, h = blockIdx.y % H, d = H * LM_DH;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, col = lane & 31, hh = lane >> 5;
  const unsigned pitch = 3u * d;
  const float* qm = qkv + (size_t)b * L * pitch + h * LM_DH;
  const float* km = qm + d;
  const float* vm = qm + 2 * d;
  const int len = lens ? max(0, min(lens[b], L)) : L;
  const int q0 = qb * 32, qi = q0 + col, qc = min(qi, L - 1);
  const int kend = causal ? min(len, min(L, q0 + 32)) : len;
  const float sc = rsqrtf((float)LM_DH);
  float qf[16], gf[16];
  at_row_frag(qf, qm, pitch, qi, L - 1, hh, sc * LM_LOG2E);    // base-2 scores, as in the forward (lse is base-2)
  at_row_frag(gf, dctx + (size_t)b * L * d + h * LM_DH, d, qi, L - 1, hh, 1.f);
  float dl = 0.f;
  {
    float cf[16];
    at_row_frag(cf, ctx + (size_t)b * L * d + h * LM_DH, d, qi, L - 1, hh, 1.f);
#pragma unroll
    for (int t = 0; t < 16; ++t) dl = fmaf(gf[t], cf[t], dl);
    dl += __shfl_xor(dl, 32, 64);
  }
  if (w == 0 && hh == 0 && qi < L) delta[((size_t)b * H + h) * L + qi] = dl;
  const float li = lse[((size_t)b * H + h) * L + qc];
  const unsigned long long e0 = (((unsigned long long)b * H + h) * L + qc) * L;
  f32x16v dq = at_zero16();
  for (int k0 = 32 * w; k0 < kend; k0 += 32 * AT_W) {
    float kf[16], vf[16], kc[16];
    at_row_frag(kf, km, pitch, k0 + col, L - 1, hh, 1.f);
    at_row_frag(vf, vm, pitch, k0 + col, L - 1, hh, 1.f);
    at_col_frag(kc, km, pitch, k0, L - 1, col, hh);
    const f32x16v st = at_mfma16(kf, qf, at_zero16());
    const f32x16v dp = at_mfma16(vf, gf, at_zero16());
    float ds[16];
    at_keep16(ds, e0 + k0, hh, key, thr, dscale);
    if (k0 + 32 <= kend && (!causal || k0 + 31 <= q0)) {      // uniform: no mask inside the block
#pragma unroll
      for (int i = 0; i < 16; ++i) ds[i] = __builtin_amdgcn_exp2f(st[i] - li) * (ds[i] * dp[i] - dl);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kj = k0 + at_acc_row(i, hh);
        const float p = (kj < kend && (!causal || kj <= qi)) ? __builtin_amdgcn_exp2f(st[i] - li) : 0.f;
        ds[i] = p * (ds[i] * dp[i] - dl);
      }
    }
    dq = at_mfma16(kc, ds, dq);
  }
  at_merge_store(red, dq, 1.f, w, lane, hh, sc, qi < L, dqkv + ((size_t)b * L + qc) * pitch + h * LM_DH);
}

// dk, dv: 32 keys per workgroup (the accumulator column): S = Q K^T and dPd = dctx V^T with the queries in the accumulator
// rows, then dv^T += dctx^T (P keep), dk^T += Q^T dS.  lse and delta of the 16 query rows a lane holds are loaded per block.
__global__ __launch_bounds__(256) void lm_attn_dkv_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                          const float* __restrict__ lse, const float* __restrict__ dctx,
                                                          const float* __restrict__ delta, float* __restrict__ dqkv, int L,
                                                          int H, int causal, unsigned key, const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;                                // device-resident key (graph replay): overrides the by-value one
  __shared__ float red[AT_W * 16 * 64];
  const int kb = blockIdx.x;                                  // first key blocks see the most queries: longest first
  const int b = blockIdx.y / H, h = blockIdx.y % H, d = H * LM_DH;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, col = lane & 31, hh = lane >> 5;
  const unsigned pitch = 3u * d;
  const float* qm = qkv + (size_t)b * L * pitch + h * LM_DH;
  const float* gm = dctx + (size_t)b * L * d + h * LM_DH;
  const float* lrow = lse + ((size_t)b * H + h) * L;
  const float* drow = delta + ((size_t)b * H + h) * L;
  const int len = lens ? max(0, min(lens[b], L)) : L;
  const int kj = kb * 32 + col, kc = min(kj, L - 1);
  const float sc = rsqrtf((float)LM_DH);
  float kf[16], vf[16];
  at_row_frag(kf, qm + d, pitch, kj, L - 1, hh, sc * LM_LOG2E); // 1/sqrt(dh) and the base-2 conversion folded into the key
  at_row_frag(vf, qm + 2 * d, pitch, kj, L - 1, hh, 1.f);
  f32x16v dk = at_zero16(), dv = at_zero16();
  const unsigned long long e0 = ((unsigned long long)b * H + h) * L;
  if (kb * 32 < len) {                                        // uniform: any visible key in this block at all?
    for (int q0 = (causal ? kb * 32 : 0) + 32 * w; q0 < L; q0 += 32 * AT_W) {
      float qa[16], ga[16], qc[16], gc[16];
      at_row_frag(qa, qm, pitch, q0 + col, L - 1, hh, 1.f);
      at_row_frag(ga, gm, d, q0 + col, L - 1, hh, 1.f);
      at_col_frag(qc, qm, pitch, q0, L - 1, col, hh);
      at_col_frag(gc, gm, d, q0, L - 1, col, hh);
      const f32x16v st = at_mfma16(qa, kf, at_zero16());
      const f32x16v dp = at_mfma16(ga, vf, at_zero16());
      float pk[16], ds[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qr = q0 + at_acc_row(i, hh), qrc = min(qr, L - 1);
        const float p = (qr < L && kj < len && (!causal || kj <= qr)) ? __builtin_amdgcn_exp2f(st[i] - lrow[qrc]) : 0.f;
        pk[i] = p * lm_keep((e0 + qrc) * L + kc, key, thr, dscale);
        ds[i] = pk[i] * dp[i] - p * drow[qrc];
      }
      dv = at_mfma16(gc, pk, dv);
      dk = at_mfma16(qc, ds, dk);
    }
  }
  float* dst = dqkv + ((size_t)b * L + kc) * pitch + h * LM_DH;
  at_merge_store(red, dk, 1.f, w, lane, hh, sc, kj < L, dst + d);
  at_merge_store(red, dv, 1.f, w, lane, hh, 1.f, kj < L, dst + 2 * d);
}

// ------------------------------------------------------------------------------------------------ add + LayerNorm
// y = LN(x + dropout(h + hbias)) * gamma + beta; one wave per row, PER = C / 64 elements per lane in registers;
// stats [rows][2] = (mean, rstd).
template <int PER>
__global__ __launch_bounds__(256) void lm_add_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ hh,
                                                            const float* __restrict__ hbias,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, float* __restrict__ stats, long long rows,
                                                            float eps, unsigned key, const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;                                // device-resident key (graph replay): overrides the by-value one
  constexpr int C = PER * 64;
  const int lane = threadIdx.x & 63;
  const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  float v[PER];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const long long e = row * C + lane + 64 * q;
    const float hb = hbias ? hbias[lane + 64 * q] : 0.f;
    const float a = (x ? x[e] : 0.f) + (hh ? (hh[e] + hb) * lm_keep((unsigned long long)e, key, thr, dscale) : 0.f);
    v[q] = a; s += a;
  }
  const float mean = wave_sum(s) / (float)C;
  float s2 = 0.f;
#pragma unroll
  for (int q = 0; q < PER; ++q) { const float dlt = v[q] - mean; s2 = fmaf(dlt, dlt, s2); }
  const float rstd = rsqrtf(wave_sum(s2) / (float)C + eps);
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int c = lane + 64 * q;
    y[row * C + c] = (v[q] - mean) * rstd * gamma[c] + beta[c];
  }
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
// xhat recomputed from (x, h, stats); g = dy gamma; dpre = rstd (g - mean(g) - xhat mean(g xhat)); dx = dpre (if dx),
// dh = dpre * mask (if dh); dgamma / dbeta / dhbias (= column sums of dh) partials per workgroup (LN_RPB rows) -> part [nwg][3][C]
constexpr int LN_RPB = 4;
template <int PER>
__global__ __launch_bounds__(256) void lm_add_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ hh,
                                                            const float* __restrict__ hbias,
                                                            const float* __restrict__ dy, const float* __restrict__ gamma,
                                                            const float* __restrict__ stats, float* __restrict__ dx,
                                                            float* __restrict__ dh, float* __restrict__ part, long long rows,
                                                            unsigned key, const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;                                // device-resident key (graph replay): overrides the by-value one
  constexpr int C = PER * 64;
  extern __shared__ float red[];                              // [4][3][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pg[PER], pb[PER], ph[PER], gm[PER], hb[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    pg[q] = 0.f; pb[q] = 0.f; ph[q] = 0.f; gm[q] = gamma[lane + 64 * q]; hb[q] = hbias ? hbias[lane + 64 * q] : 0.f;
  }
  for (int rr = 0; rr < LN_RPB / 4; ++rr) {
    const long long row = (long long)blockIdx.x * LN_RPB + wave * (LN_RPB / 4) + rr;
    if (row >= rows) break;                                   // wave-uniform
    float xh[PER], g[PER];
    float sg = 0.f, sgx = 0.f;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const long long e = row * C + lane + 64 * q;
      const float dyv = dy[e];
      const float pre = (x ? x[e] : 0.f) + (hh ? (hh[e] + hb[q]) * lm_keep((unsigned long long)e, key, thr, dscale) : 0.f);
      const float xv = (pre - mean) * rstd, gv = dyv * gm[q];
      xh[q] = xv; g[q] = gv; sg += gv; sgx = fmaf(gv, xv, sgx);
      pg[q] = fmaf(dyv, xv, pg[q]); pb[q] += dyv;
    }
    sg = wave_sum(sg) / (float)C; sgx = wave_sum(sgx) / (float)C;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const long long e = row * C + lane + 64 * q;
      const float dpre = rstd * (g[q] - sg - xh[q] * sgx);
      if (dx) dx[e] = dpre;
      const float dhv = dpre * lm_keep((unsigned long long)e, key, thr, dscale);
      if (dh) dh[e] = dhv;
      ph[q] += dhv;
    }
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    red[(wave * 3 + 0) * C + lane + 64 * q] = pg[q];
    red[(wave * 3 + 1) * C + lane + 64 * q] = pb[q];
    red[(wave * 3 + 2) * C + lane + 64 * q] = ph[q];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 3 * C; c += 256) {
    const int plane = c / C, cc = c % C;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) a += red[(w * 3 + plane) * C + cc];
    part[(size_t)blockIdx.x * 3 * C + c] = a;
  }
}
#define LM_LN_DISPATCH(dim, CALL)                                                       \
  switch ((dim) / 64) {                                                                 \
    case 1: CALL(1); break;  case 2: CALL(2); break;  case 3: CALL(3); break;           \
    case 4: CALL(4); break;  case 8: CALL(8); break;  case 12: CALL(12); break;         \
    case 16: CALL(16); break; case 32: CALL(32); break;                                 \
    default: SMT_CHECK_ARG(false, "add_ln: dim %d not built (64 x {1,2,3,4,8,12,16,32})", (int)(dim)); \
  }
// fixed-order column sums of part [n][width] -> out [width]: a workgroup takes 64 columns, its four waves a quarter of
// the rows each; a wave keeps eight loads in flight (row r goes to accumulator r % 8) and the partial sums are combined
// in a fixed order.
__global__ __launch_bounds__(256) void lm_colsum_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int width) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
  const int per = (n + 3) / 4, lo = wave * per, hi = min(n, lo + per);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < width) {
    int i = lo;
    for (; i + 7 < hi; i += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += part[(size_t)(i + u) * width + c];
    }
    for (int u = 0; i < hi; ++i, ++u) s[u] += part[(size_t)i * width + c];
  }
  red[wave][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (wave == 0 && c < width) out[c] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// ------------------------------------------------------------------------------------------------ bias + relu + dropout
__global__ __launch_bounds__(256) void lm_bias_relu_fwd_kernel(float* __restrict__ hbuf, const float* __restrict__ bias,
                                                               long long rows, int C, unsigned key, const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;                                // device-resident key (graph replay): overrides the by-value one
  const long long total = rows * C;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const float v = fmaxf(hbuf[e] + bias[e % C], 0.f);
    hbuf[e] = v * lm_keep((unsigned long long)e, key, thr, dscale);
  }
}
// dh = da * mask * [a != 0] (dh may be da); bias-gradient partials per workgroup row block -> part [nblk][C]
__global__ __launch_bounds__(256) void lm_bias_relu_bwd_kernel(const float* __restrict__ a, const float* da, float* dh,
                                                               float* __restrict__ part, long long rows, int C, int rows_per_blk,
                                                               unsigned key, const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;                                // device-resident key (graph replay): overrides the by-value one
  const long long r0 = (long long)blockIdx.x * rows_per_blk;
  for (int c = blockIdx.y * 256 + threadIdx.x; c < C; c += 256 * gridDim.y) {
    float s = 0.f;
    for (long long r = r0; r < min(rows, r0 + rows_per_blk); ++r) {
      const long long e = r * C + c;
      // a = relu(.) * keep: a != 0 <=> pre-activation > 0 and kept (the derivative of both at once)
      const float g = a[e] != 0.f ? da[e] * lm_keep((unsigned long long)e, key, thr, dscale) : 0.f;
      dh[e] = g; s += g;
    }
    part[(size_t)blockIdx.x * C + c] = s;
  }
}

// ------------------------------------------------------------------------------------------------ cross entropy
// One wave per row: row_loss = logsumexp(logits) - logits[target] for target >= 0; out [rows][2] = (loss or 0, correct 0/1).
__global__ __launch_bounds__(256) void lm_ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                        float* __restrict__ out, float* __restrict__ lse, long long rows, int V) {
  const int lane = threadIdx.x & 63;
  const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  const long long tg = target[row];
  float m = -INFINITY;
  int am = 0x7fffffff;
  for (int c = lane; c < V; c += 64) {
    const float v = logits[row * V + c];
    if (v > m) { m = v; am = c; }                              // increasing c per lane: first maximum
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    if (om > m || (om == m && oa < am)) { m = om; am = oa; }   // lowest index among equal maxima (torch.argmax)
  }
  float z = 0.f;
  for (int c = lane; c < V; c += 64) z += __expf(logits[row * V + c] - m);
  z = wave_sum(z);
  const float l = m + __logf(z);
  if (lane == 0) {
    lse[row] = l;
    out[2 * row] = tg >= 0 ? l - logits[row * V + tg] : 0.f;
    out[2 * row + 1] = (tg >= 0 && am == (int)tg) ? 1.f : 0.f;
  }
}
__global__ __launch_bounds__(256) void lm_ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                        const float* __restrict__ lse, const float* __restrict__ coef,
                                                        float* __restrict__ dlogits, long long rows, int V) {
  const float g = coef[0];                                     // upstream gradient / number of valid rows
  const long long total = rows * V;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long row = e / V;
    const int c = (int)(e % V);
    const long long tg = target[row];
    dlogits[e] = tg >= 0 ? g * (__expf(logits[e] - lse[row]) - (c == tg ? 1.f : 0.f)) : 0.f;
  }
}

// keys[s] = fmix32(seed * 0x9E3779B1 + s * 0x7F4A7C15 + 1): the per-site dropout keys (smt_amd.convops.dropout_key) of the step
// whose seed sits in device memory -- what a captured graph replays with a new seed every time
__global__ void lm_make_keys_kernel(const unsigned* __restrict__ seed, unsigned* __restrict__ keys, int n) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < n) keys[s] = fmix32(seed[0] * 0x9E3779B1u + (unsigned)s * 0x7F4A7C15u + 1u);
}

static unsigned lm_grid(long long total) { return (unsigned)std::min<long long>(4096, (total + 255) / 256); }

}  // namespace smt

using namespace smt;

extern "C" int smt_lm_make_keys(const uint32_t* seed_dev, uint32_t* keys_dev, int n_sites, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(seed_dev && keys_dev && n_sites > 0, "smt_lm_make_keys: bad arguments");
  lm_make_keys_kernel<<<(n_sites + 255) / 256, 256, 0, stream>>>(seed_dev, keys_dev, n_sites);
  SMT_CHECK_LAUNCH("lm_make_keys");
  return 0;
}

extern "C" int smt_lm_embed_fwd(const int64_t* tokens, const float* emb, const float* pe, float* out, int batch, int len, int dim,
                                float mul, uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0 || len <= 0) return 0;
  SMT_CHECK_ARG(tokens && emb && pe && out, "smt_lm_embed_fwd: null pointer");
  lm_embed_fwd_kernel<<<lm_grid((long long)batch * len * dim), 256, 0, stream>>>((const long long*)tokens, emb, pe, out, batch, len, dim,
                                                                             mul, drop_key, drop_key_dev, drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("lm_embed_fwd");
  return 0;
}

extern "C" int smt_lm_embed_bwd(const int64_t* tokens, const float* dout, float* demb, int batch, int len, int dim, int vocab_rows,
                                float mul, uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, int64_t padding_idx,
                                smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(demb, "smt_lm_embed_bwd: null pointer");
  (void)hipMemsetAsync(demb, 0, (size_t)vocab_rows * dim * sizeof(float), stream);
  if (batch <= 0 || len <= 0) return 0;
  SMT_CHECK_ARG(tokens && dout, "smt_lm_embed_bwd: null pointer");
  lm_embed_bwd_kernel<<<lm_grid((long long)batch * len * dim), 256, 0, stream>>>((const long long*)tokens, dout, demb, batch, len, dim,
                                                                             mul, drop_key, drop_key_dev, drop_thresh16, drop_scale, padding_idx);
  SMT_CHECK_LAUNCH("lm_embed_bwd");
  return 0;
}

extern "C" int smt_lm_attention_fwd(const float* qkv, const int* lens, float* ctx, float* lse, int batch, int len, int heads,
                                    int causal, uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0 || len <= 0) return 0;
  SMT_CHECK_ARG(qkv && ctx && lse, "smt_lm_attention_fwd: null pointer");
  SMT_CHECK_ARG(heads >= 1 && lm_len_ok(len, heads), "smt_lm_attention_fwd: len * 3 * heads * 32 must be < 2^31 (len %d, heads %d)", len, heads);
  SMT_CHECK_ARG((long long)batch * heads <= 65535, "smt_lm_attention_fwd: batch * heads must be <= 65535");
  lm_attn_fwd_kernel<<<dim3((len + 31) / 32, batch * heads), 256, 0, stream>>>(qkv, lens, ctx, lse, len, heads, causal, drop_key, drop_key_dev,
                                                                           drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("lm_attention_fwd");
  return 0;
}

extern "C" int smt_lm_attention_bwd(const float* qkv, const int* lens, const float* ctx, const float* lse, const float* dctx,
                                    float* dqkv, float* delta, int batch, int len, int heads, int causal, uint32_t drop_key,
                                    const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0 || len <= 0) return 0;
  SMT_CHECK_ARG(qkv && ctx && lse && dctx && dqkv && delta, "smt_lm_attention_bwd: null pointer");
  SMT_CHECK_ARG(heads >= 1 && lm_len_ok(len, heads), "smt_lm_attention_bwd: len * 3 * heads * 32 must be < 2^31 (len %d, heads %d)", len, heads);
  SMT_CHECK_ARG((long long)batch * heads <= 65535, "smt_lm_attention_bwd: batch * heads must be <= 65535");
  const dim3 grid((len + 31) / 32, batch * heads);
  lm_attn_dq_kernel<<<grid, 256, 0, stream>>>(qkv, lens, ctx, lse, dctx, dqkv, delta, len, heads, causal, drop_key, drop_key_dev, drop_thresh16,
                                            drop_scale);
  SMT_CHECK_LAUNCH("lm_attention_dq");
  lm_attn_dkv_kernel<<<grid, 256, 0, stream>>>(qkv, lens, lse, dctx, delta, dqkv, len, heads, causal, drop_key, drop_key_dev, drop_thresh16,
                                             drop_scale);
  SMT_CHECK_LAUNCH("lm_attention_dkv");
  return 0;
}

extern "C" int smt_lm_add_ln_fwd(const float* x, const float* h, const float* h_bias, const float* gamma, const float* beta, float* y,
                                 float* stats,
                                 int64_t rows, int dim, float eps, uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale,
                                 smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows <= 0) return 0;
  SMT_CHECK_ARG((x || h) && gamma && beta && y && stats, "smt_lm_add_ln_fwd: null pointer");
  SMT_CHECK_ARG(dim % 64 == 0 && dim <= 2048, "smt_lm_add_ln_fwd: dim must be a multiple of 64 up to 2048 (got %d)", dim);
#define LM_CALL(P) lm_add_ln_fwd_kernel<P><<<(unsigned)((rows + 3) / 4), 256, 0, stream>>>(x, h, h_bias, gamma, beta, y, stats, rows, eps, \
                                                                                     drop_key, drop_key_dev, drop_thresh16, drop_scale)
  LM_LN_DISPATCH(dim, LM_CALL)
#undef LM_CALL
  SMT_CHECK_LAUNCH("lm_add_ln_fwd");
  return 0;
}

extern "C" size_t smt_lm_add_ln_bwd_workspace_bytes(int64_t rows, int dim) {
  return (size_t)((rows + LN_RPB - 1) / LN_RPB) * 3 * dim * sizeof(float);
}

extern "C" int smt_lm_add_ln_bwd(const float* x, const float* h, const float* h_bias, const float* dy, const float* gamma,
                                 const float* stats, float* dx, float* dh, float* dparams, int64_t rows, int dim, uint32_t drop_key,
                                 const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, void* workspace, size_t workspace_bytes,
                                 smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(dparams, "smt_lm_add_ln_bwd: null pointer");
  if (rows <= 0) {
    (void)hipMemsetAsync(dparams, 0, 3 * (size_t)dim * sizeof(float), stream);
    return 0;
  }
  SMT_CHECK_ARG((x || h) && dy && gamma && stats && workspace, "smt_lm_add_ln_bwd: null pointer");
  SMT_CHECK_ARG(dim % 64 == 0 && dim <= 2048, "smt_lm_add_ln_bwd: dim must be a multiple of 64 up to 2048 (got %d)", dim);
  SMT_CHECK_ARG(workspace_bytes >= smt_lm_add_ln_bwd_workspace_bytes(rows, dim), "smt_lm_add_ln_bwd: workspace too small");
  const int nblk = (int)((rows + LN_RPB - 1) / LN_RPB);
  float* part = (float*)workspace;                           // [nblk][3][dim]: dgamma, dbeta, dh_bias partials
  const size_t lds = 12 * (size_t)dim * sizeof(float);
#define LM_CALL(P)                                                                                                        \
  (void)hipFuncSetAttribute((const void*)lm_add_ln_bwd_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
  lm_add_ln_bwd_kernel<P><<<nblk, 256, lds, stream>>>(x, h, h_bias, dy, gamma, stats, dx, dh, part, rows, drop_key,        \
                                                      drop_key_dev, drop_thresh16, drop_scale)
  LM_LN_DISPATCH(dim, LM_CALL)
#undef LM_CALL
  SMT_CHECK_LAUNCH("lm_add_ln_bwd");
  lm_colsum_kernel<<<(3 * dim + 63) / 64, 256, 0, stream>>>(part, dparams, nblk, 3 * dim);
  SMT_CHECK_LAUNCH("lm_colsum");
  return 0;
}

extern "C" int smt_lm_bias_relu_fwd(float* h, const float* bias, int64_t rows, int dim, uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16,
                                    float drop_scale, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows <= 0) return 0;
  SMT_CHECK_ARG(h && bias, "smt_lm_bias_relu_fwd: null pointer");
  lm_bias_relu_fwd_kernel<<<lm_grid(rows * dim), 256, 0, stream>>>(h, bias, rows, dim, drop_key, drop_key_dev, drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("lm_bias_relu_fwd");
  return 0;
}

extern "C" size_t smt_lm_bias_relu_bwd_workspace_bytes(int64_t rows, int dim) { return (size_t)((rows + 15) / 16) * dim * sizeof(float); }

extern "C" int smt_lm_bias_relu_bwd(const float* a, const float* da, float* dh, float* dbias, int64_t rows, int dim, uint32_t drop_key,
                                    const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, void* workspace, size_t workspace_bytes,
                                    smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(dbias, "smt_lm_bias_relu_bwd: null pointer");
  if (rows <= 0) { (void)hipMemsetAsync(dbias, 0, dim * sizeof(float), stream); return 0; }
  SMT_CHECK_ARG(a && da && dh && workspace, "smt_lm_bias_relu_bwd: null pointer");
  SMT_CHECK_ARG(workspace_bytes >= smt_lm_bias_relu_bwd_workspace_bytes(rows, dim), "smt_lm_bias_relu_bwd: workspace too small");
  const int nblk = (int)((rows + 15) / 16);
  lm_bias_relu_bwd_kernel<<<dim3(nblk, (dim + 255) / 256), 256, 0, stream>>>(a, da, dh, (float*)workspace, rows, dim, 16, drop_key, drop_key_dev, drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("lm_bias_relu_bwd");
  lm_colsum_kernel<<<(dim + 63) / 64, 256, 0, stream>>>((const float*)workspace, dbias, nblk, dim);
  SMT_CHECK_LAUNCH("lm_colsum");
  return 0;
}

extern "C" int smt_lm_ce_fwd(const float* logits, const int64_t* target, float* row_out, float* lse, int64_t rows, int vocab,
                             smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows <= 0) return 0;
  SMT_CHECK_ARG(logits && target && row_out && lse, "smt_lm_ce_fwd: null pointer");
  lm_ce_fwd_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, stream>>>(logits, (const long long*)target, row_out, lse, rows, vocab);
  SMT_CHECK_LAUNCH("lm_ce_fwd");
  return 0;
}

extern "C" int smt_lm_ce_bwd(const float* logits, const int64_t* target, const float* lse, const float* coef, float* dlogits,
                             int64_t rows, int vocab, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows <= 0) return 0;
  SMT_CHECK_ARG(logits && target && lse && coef && dlogits, "smt_lm_ce_bwd: null pointer");
  lm_ce_bwd_kernel<<<lm_grid(rows * vocab), 256, 0, stream>>>(logits, (const long long*)target, lse, coef, dlogits, rows, vocab);
  SMT_CHECK_LAUNCH("lm_ce_bwd");
  return 0;
}
