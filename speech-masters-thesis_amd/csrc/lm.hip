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
// Activations are [B, L, C] fp32 rows (the fp32 parity path; L <= 512, head dim 32).  Dropout masks come from the
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
                                                           int D, float mul, unsigned key, unsigned thr, float dscale) {
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
                                                           unsigned key, unsigned thr, float dscale, long long pad_idx) {
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
// element index ((b H + h) L + i) L + j.
//
// Work split (all three kernels): a workgroup owns 64 consecutive rows of one (batch, head) -- queries in the forward and
// the dq kernel, keys in the dk/dv kernel -- and AT_G = 4 adjacent lanes share a row, lane g walking the opposite index
// j = g, g + 4, ...; the four partial results meet in two xor-shuffles.  The opposite side (K, V or scaled Q, dctx) sits in
// LDS at a pitch of 36 floats: the four rows a lane group reads at once fall into disjoint bank groups, and the 16 groups
// of a wave read the same four rows (broadcast).  B H ceil(L / 64) workgroups (640 at the reference's 8 x 16 x 258) of four
// waves fill the chip; the causal triangle makes later tiles longer, so tiles are issued longest first.
constexpr int LM_DH = 32, LM_MAXL = 512, AT_G = 4, AT_ROWS = 64, AT_LD = 36;

// stage `n` rows of width 32 from global (row pitch `pitch` floats) into LDS rows of AT_LD floats, times `mul`
__device__ __forceinline__ void at_stage(float* dst, const float* __restrict__ src, size_t pitch, int n, float mul) {
  for (int f = threadIdx.x; f < n * 8; f += 256) {
    const int j = f >> 3, c4 = f & 7;
    f32x4 v = *(const f32x4*)(src + (size_t)j * pitch + c4 * 4);
    *(f32x4*)(dst + j * AT_LD + c4 * 4) = v * mul;
  }
}
__device__ __forceinline__ float at_dot(const float* __restrict__ row, const float (&q)[LM_DH]) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int c = 0; c < LM_DH; c += 8) {
    const f32x4 a = *(const f32x4*)(row + c), b = *(const f32x4*)(row + c + 4);
    s0 = fmaf(q[c], a.x, s0); s0 = fmaf(q[c + 1], a.y, s0); s0 = fmaf(q[c + 2], a.z, s0); s0 = fmaf(q[c + 3], a.w, s0);
    s1 = fmaf(q[c + 4], b.x, s1); s1 = fmaf(q[c + 5], b.y, s1); s1 = fmaf(q[c + 6], b.z, s1); s1 = fmaf(q[c + 7], b.w, s1);
  }
  return s0 + s1;
}
__device__ __forceinline__ void at_axpy(float (&acc)[LM_DH], float a, const float* __restrict__ row) {
#pragma unroll
  for (int c = 0; c < LM_DH; c += 4) {
    const f32x4 v = *(const f32x4*)(row + c);
    acc[c] = fmaf(a, v.x, acc[c]); acc[c + 1] = fmaf(a, v.y, acc[c + 1]);
    acc[c + 2] = fmaf(a, v.z, acc[c + 2]); acc[c + 3] = fmaf(a, v.w, acc[c + 3]);
  }
}
__device__ __forceinline__ void at_load_row(float (&r)[LM_DH], const float* __restrict__ src, float mul) {
#pragma unroll
  for (int c = 0; c < LM_DH; c += 4) {
    const f32x4 v = *(const f32x4*)(src + c);
    r[c] = v.x * mul; r[c + 1] = v.y * mul; r[c + 2] = v.z * mul; r[c + 3] = v.w * mul;
  }
}
// lane g of the group keeps channels [8 g, 8 g + 8) of the group sum of acc
__device__ __forceinline__ void at_group_sum_store(float (&acc)[LM_DH], int g, float mul, float* __restrict__ dst) {
#pragma unroll
  for (int c = 0; c < LM_DH; ++c) {
    acc[c] += __shfl_xor(acc[c], 1, 64);
    acc[c] += __shfl_xor(acc[c], 2, 64);
  }
  f32x4 o0, o1;
#pragma unroll
  for (int gg = 0; gg < AT_G; ++gg)
    if (g == gg) {
      o0 = f32x4{acc[8 * gg], acc[8 * gg + 1], acc[8 * gg + 2], acc[8 * gg + 3]} * mul;
      o1 = f32x4{acc[8 * gg + 4], acc[8 * gg + 5], acc[8 * gg + 6], acc[8 * gg + 7]} * mul;
    }
  *(f32x4*)(dst + 8 * g) = o0;
  *(f32x4*)(dst + 8 * g + 4) = o1;
}

__global__ __launch_bounds__(256) void lm_attn_fwd_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                          float* __restrict__ ctx, float* __restrict__ lse, int L, int H,
                                                          int causal, unsigned key, unsigned thr, float dscale) {
  extern __shared__ float sm[];                               // K [nk][36] | V [nk][36]
  const int tile = gridDim.x - 1 - blockIdx.x;                // longest (last) tiles first
  const int b = blockIdx.y / H, h = blockIdx.y % H, d = H * LM_DH;
  const float* base = qkv + (size_t)b * L * 3 * d + h * LM_DH;
  const int len = lens ? max(0, min(lens[b], L)) : L;
  const int nk = causal ? min(len, min(L, (tile + 1) * AT_ROWS)) : len;   // keys any row of this tile can see
  float* ks = sm;
  float* vs = sm + (size_t)nk * AT_LD;
  at_stage(ks, base + d, 3 * (size_t)d, nk, 1.f);
  at_stage(vs, base + 2 * d, 3 * (size_t)d, nk, 1.f);
  __syncthreads();
  const int g = threadIdx.x & (AT_G - 1), i = tile * AT_ROWS + (threadIdx.x >> 2);
  if (i >= L) return;                                         // whole lane groups leave together
  float q[LM_DH], acc[LM_DH];
  at_load_row(q, base + (size_t)i * 3 * d, rsqrtf((float)LM_DH));
#pragma unroll
  for (int c = 0; c < LM_DH; ++c) acc[c] = 0.f;
  float m = -INFINITY, z = 0.f;
  const int jn = causal ? min(i + 1, len) : len;
  const unsigned long long e0 = (((unsigned long long)b * H + h) * L + i) * L;
  for (int j = g; j < jn; j += AT_G) {
    const float s = at_dot(ks + j * AT_LD, q);
    if (s > m) {                                              // new running maximum: rescale what is accumulated
      const float corr = __expf(m - s);
      z *= corr;
#pragma unroll
      for (int c = 0; c < LM_DH; ++c) acc[c] *= corr;
      m = s;
    }
    const float p = __expf(s - m);
    z += p;
    at_axpy(acc, p * lm_keep(e0 + j, key, thr, dscale), vs + j * AT_LD);
  }
  // merge the four partial softmaxes of the row
  float mm = fmaxf(m, __shfl_xor(m, 1, 64));
  mm = fmaxf(mm, __shfl_xor(mm, 2, 64));
  const float sc = (m == -INFINITY) ? 0.f : __expf(m - mm);
  z *= sc;
  z += __shfl_xor(z, 1, 64);
  z += __shfl_xor(z, 2, 64);
#pragma unroll
  for (int c = 0; c < LM_DH; ++c) acc[c] *= sc;
  at_group_sum_store(acc, g, jn > 0 ? 1.f / z : 0.f, ctx + ((size_t)b * L + i) * d + h * LM_DH);
  if (g == 0) lse[((size_t)b * H + h) * L + i] = jn > 0 ? mm + __logf(z) : 0.f;
}

// Backward: P_ij = exp(s_ij - lse_i); dPd_ij = dctx_i . v_j; delta_i = sum_j P_ij keep_ij dPd_ij = dctx_i . ctx_i;
// dS_ij = P_ij (keep_ij dPd_ij - delta_i); dq_i = sum_j dS_ij k_j / sqrt(dh); dk_j = sum_i dS_ij q_i / sqrt(dh);
// dv_j = sum_i P_ij keep_ij dctx_i.  dq: query tiles against K, V in LDS.
__global__ __launch_bounds__(256) void lm_attn_dq_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                         const float* __restrict__ ctx, const float* __restrict__ lse,
                                                         const float* __restrict__ dctx, float* __restrict__ dqkv, int L, int H,
                                                         int causal, unsigned key, unsigned thr, float dscale) {
  extern __shared__ float sm[];
  const int tile = gridDim.x - 1 - blockIdx.x;
  const int b = blockIdx.y / H, h = blockIdx.y % H, d = H * LM_DH;
  const float* base = qkv + (size_t)b * L * 3 * d + h * LM_DH;
  const int len = lens ? max(0, min(lens[b], L)) : L;
  const int nk = causal ? min(len, min(L, (tile + 1) * AT_ROWS)) : len;
  float* ks = sm;
  float* vs = sm + (size_t)nk * AT_LD;
  at_stage(ks, base + d, 3 * (size_t)d, nk, 1.f);
  at_stage(vs, base + 2 * d, 3 * (size_t)d, nk, 1.f);
  __syncthreads();
  const int g = threadIdx.x & (AT_G - 1), i = tile * AT_ROWS + (threadIdx.x >> 2);
  if (i >= L) return;
  const float sc = rsqrtf((float)LM_DH);
  float q[LM_DH], go[LM_DH], dq[LM_DH];
  at_load_row(q, base + (size_t)i * 3 * d, sc);
  at_load_row(go, dctx + ((size_t)b * L + i) * d + h * LM_DH, 1.f);
  const float dl = at_dot(ctx + ((size_t)b * L + i) * d + h * LM_DH, go);
#pragma unroll
  for (int c = 0; c < LM_DH; ++c) dq[c] = 0.f;
  const float li = lse[((size_t)b * H + h) * L + i];
  const int jn = causal ? min(i + 1, len) : len;
  const unsigned long long e0 = (((unsigned long long)b * H + h) * L + i) * L;
  for (int j = g; j < jn; j += AT_G) {
    const float p = __expf(at_dot(ks + j * AT_LD, q) - li);
    const float dp = at_dot(vs + j * AT_LD, go);
    at_axpy(dq, p * (lm_keep(e0 + j, key, thr, dscale) * dp - dl), ks + j * AT_LD);
  }
  at_group_sum_store(dq, g, sc, dqkv + ((size_t)b * L + i) * 3 * d + h * LM_DH);
}

// dk, dv: key tiles against the scaled queries and dctx (plus lse, delta per query) in LDS.
__global__ __launch_bounds__(256) void lm_attn_dkv_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                          const float* __restrict__ ctx, const float* __restrict__ lse,
                                                          const float* __restrict__ dctx, float* __restrict__ dqkv, int L, int H,
                                                          int causal, unsigned key, unsigned thr, float dscale) {
  extern __shared__ float sm[];                               // Q [nq][36] | dO [nq][36] | lse [nq] | delta [nq]
  const int tile = blockIdx.x;                                // first key tiles see the most queries: already longest first
  const int b = blockIdx.y / H, h = blockIdx.y % H, d = H * LM_DH;
  const float* base = qkv + (size_t)b * L * 3 * d + h * LM_DH;
  float* dbase = dqkv + (size_t)b * L * 3 * d + h * LM_DH;
  const int len = lens ? max(0, min(lens[b], L)) : L;
  const int i0 = causal ? tile * AT_ROWS : 0, nq = L - i0;    // queries [i0, L) can see keys of this tile
  const int g = threadIdx.x & (AT_G - 1), j = tile * AT_ROWS + (threadIdx.x >> 2);
  float* qs = sm;
  float* gs = qs + (size_t)nq * AT_LD;
  float* ls = gs + (size_t)nq * AT_LD;
  float* delta = ls + nq;
  const bool tile_live = tile * AT_ROWS < len;                // uniform: any visible key in this tile at all?
  if (tile_live) {
    const float sc = rsqrtf((float)LM_DH);
    at_stage(qs, base + (size_t)i0 * 3 * d, 3 * (size_t)d, nq, sc);
    const float* gsrc = dctx + ((size_t)b * L + i0) * d + h * LM_DH;
    const float* csrc = ctx + ((size_t)b * L + i0) * d + h * LM_DH;
    for (int f = threadIdx.x; f < ((nq * 8 + 63) & ~63); f += 256) {      // whole waves: the shuffles below need all 8 lanes
      const int r = f >> 3, c4 = f & 7;
      float part = 0.f;
      if (r < nq) {
        const f32x4 gv = *(const f32x4*)(gsrc + (size_t)r * d + c4 * 4), cv = *(const f32x4*)(csrc + (size_t)r * d + c4 * 4);
        *(f32x4*)(gs + r * AT_LD + c4 * 4) = gv;
        part = gv.x * cv.x + gv.y * cv.y + gv.z * cv.z + gv.w * cv.w;
      }
      part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
      if (r < nq && c4 == 0) { delta[r] = part; ls[r] = lse[((size_t)b * H + h) * L + i0 + r]; }
    }
  }
  __syncthreads();
  if (j >= L) return;
  float dk[LM_DH], dv[LM_DH];
#pragma unroll
  for (int c = 0; c < LM_DH; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
  if (j < len) {
    float kk[LM_DH], vv[LM_DH];
    at_load_row(kk, base + (size_t)j * 3 * d + d, 1.f);
    at_load_row(vv, base + (size_t)j * 3 * d + 2 * d, 1.f);
    const int r0 = causal ? (j - i0) : 0;                     // first query row (relative to i0) that sees key j
    for (int r = (r0 & ~(AT_G - 1)) + g; r < nq; r += AT_G) {
      if (r < r0) continue;
      const float p = __expf(at_dot(qs + r * AT_LD, kk) - ls[r]);
      const float kp = lm_keep((((unsigned long long)b * H + h) * L + i0 + r) * L + j, key, thr, dscale);
      const float dp = at_dot(gs + r * AT_LD, vv);
      at_axpy(dk, p * (kp * dp - delta[r]), qs + r * AT_LD);   // q in LDS already carries 1/sqrt(dh)
      at_axpy(dv, p * kp, gs + r * AT_LD);
    }
  }
  at_group_sum_store(dk, g, 1.f, dbase + (size_t)j * 3 * d + d);
  at_group_sum_store(dv, g, 1.f, dbase + (size_t)j * 3 * d + 2 * d);
}

// ------------------------------------------------------------------------------------------------ add + LayerNorm
// y = LN(x + dropout(h)) * gamma + beta; one wave per row, PER = C / 64 elements per lane in registers;
// stats [rows][2] = (mean, rstd).
template <int PER>
__global__ __launch_bounds__(256) void lm_add_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ hh,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, float* __restrict__ stats, long long rows,
                                                            float eps, unsigned key, unsigned thr, float dscale) {
  constexpr int C = PER * 64;
  const int lane = threadIdx.x & 63;
  const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  float v[PER];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const long long e = row * C + lane + 64 * q;
    const float a = (x ? x[e] : 0.f) + (hh ? hh[e] * lm_keep((unsigned long long)e, key, thr, dscale) : 0.f);
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
// dh = dpre * mask (if dh); dgamma / dbeta partials per workgroup (LN_RPB rows, LN_RPB / 4 per wave) -> part [nwg][2][C]
constexpr int LN_RPB = 16;
template <int PER>
__global__ __launch_bounds__(256) void lm_add_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ hh,
                                                            const float* __restrict__ dy, const float* __restrict__ gamma,
                                                            const float* __restrict__ stats, float* __restrict__ dx,
                                                            float* __restrict__ dh, float* __restrict__ part, long long rows,
                                                            unsigned key, unsigned thr, float dscale) {
  constexpr int C = PER * 64;
  extern __shared__ float red[];                              // [4][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pg[PER], pb[PER], gm[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) { pg[q] = 0.f; pb[q] = 0.f; gm[q] = gamma[lane + 64 * q]; }
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
      const float pre = (x ? x[e] : 0.f) + (hh ? hh[e] * lm_keep((unsigned long long)e, key, thr, dscale) : 0.f);
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
      if (dh) dh[e] = dpre * lm_keep((unsigned long long)e, key, thr, dscale);
    }
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    red[(wave * 2 + 0) * C + lane + 64 * q] = pg[q];
    red[(wave * 2 + 1) * C + lane + 64 * q] = pb[q];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, bsum = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { a += red[(w * 2 + 0) * C + c]; bsum += red[(w * 2 + 1) * C + c]; }
    part[((size_t)blockIdx.x * 2 + 0) * C + c] = a;
    part[((size_t)blockIdx.x * 2 + 1) * C + c] = bsum;
  }
}
#define LM_LN_DISPATCH(dim, CALL)                                                       \
  switch ((dim) / 64) {                                                                 \
    case 1: CALL(1); break;  case 2: CALL(2); break;  case 4: CALL(4); break;           \
    case 8: CALL(8); break;  case 12: CALL(12); break; case 16: CALL(16); break;        \
    case 32: CALL(32); break;                                                           \
    default: SMT_CHECK_ARG(false, "add_ln: dim %d not built (64 x {1,2,4,8,12,16,32})", (int)(dim)); \
  }
// fixed-order column sums of part [n][planes * C] -> out [planes * C]: a workgroup takes 64 columns, its four waves a
// quarter of the rows each (in order), and the four partial sums are added in wave order.
__global__ __launch_bounds__(256) void lm_colsum_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int width) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
  const int per = (n + 3) / 4, lo = wave * per, hi = min(n, lo + per);
  float s0 = 0.f, s1 = 0.f;
  if (c < width) {
    int i = lo;
    for (; i + 1 < hi; i += 2) { s0 += part[(size_t)i * width + c]; s1 += part[(size_t)(i + 1) * width + c]; }
    if (i < hi) s0 += part[(size_t)i * width + c];
  }
  red[wave][lane] = s0 + s1;
  __syncthreads();
  if (wave == 0 && c < width) out[c] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// ------------------------------------------------------------------------------------------------ bias + relu + dropout
__global__ __launch_bounds__(256) void lm_bias_relu_fwd_kernel(float* __restrict__ hbuf, const float* __restrict__ bias,
                                                               long long rows, int C, unsigned key, unsigned thr, float dscale) {
  const long long total = rows * C;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const float v = fmaxf(hbuf[e] + bias[e % C], 0.f);
    hbuf[e] = v * lm_keep((unsigned long long)e, key, thr, dscale);
  }
}
// dh = da * mask * [a != 0] (dh may be da); bias-gradient partials per workgroup row block -> part [nblk][C]
__global__ __launch_bounds__(256) void lm_bias_relu_bwd_kernel(const float* __restrict__ a, const float* da, float* dh,
                                                               float* __restrict__ part, long long rows, int C, int rows_per_blk,
                                                               unsigned key, unsigned thr, float dscale) {
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

static unsigned lm_grid(long long total) { return (unsigned)std::min<long long>(4096, (total + 255) / 256); }

}  // namespace smt

using namespace smt;

extern "C" int smt_lm_embed_fwd(const int64_t* tokens, const float* emb, const float* pe, float* out, int batch, int len, int dim,
                                float mul, uint32_t drop_key, uint32_t drop_thresh16, float drop_scale, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0 || len <= 0) return 0;
  SMT_CHECK_ARG(tokens && emb && pe && out, "smt_lm_embed_fwd: null pointer");
  lm_embed_fwd_kernel<<<lm_grid((long long)batch * len * dim), 256, 0, stream>>>((const long long*)tokens, emb, pe, out, batch, len, dim,
                                                                             mul, drop_key, drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("lm_embed_fwd");
  return 0;
}

extern "C" int smt_lm_embed_bwd(const int64_t* tokens, const float* dout, float* demb, int batch, int len, int dim, int vocab_rows,
                                float mul, uint32_t drop_key, uint32_t drop_thresh16, float drop_scale, int64_t padding_idx,
                                smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(demb, "smt_lm_embed_bwd: null pointer");
  (void)hipMemsetAsync(demb, 0, (size_t)vocab_rows * dim * sizeof(float), stream);
  if (batch <= 0 || len <= 0) return 0;
  SMT_CHECK_ARG(tokens && dout, "smt_lm_embed_bwd: null pointer");
  lm_embed_bwd_kernel<<<lm_grid((long long)batch * len * dim), 256, 0, stream>>>((const long long*)tokens, dout, demb, batch, len, dim,
                                                                             mul, drop_key, drop_thresh16, drop_scale, padding_idx);
  SMT_CHECK_LAUNCH("lm_embed_bwd");
  return 0;
}

extern "C" int smt_lm_attention_fwd(const float* qkv, const int* lens, float* ctx, float* lse, int batch, int len, int heads,
                                    int causal, uint32_t drop_key, uint32_t drop_thresh16, float drop_scale, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0 || len <= 0) return 0;
  SMT_CHECK_ARG(qkv && ctx && lse, "smt_lm_attention_fwd: null pointer");
  SMT_CHECK_ARG(len <= LM_MAXL && heads >= 1, "smt_lm_attention_fwd: len must be <= %d (got %d)", LM_MAXL, len);
  SMT_CHECK_ARG((long long)batch * heads <= 65535, "smt_lm_attention_fwd: batch * heads must be <= 65535");
  const size_t lds = 2 * (size_t)len * AT_LD * sizeof(float);
  (void)hipFuncSetAttribute((const void*)lm_attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const dim3 grid((len + AT_ROWS - 1) / AT_ROWS, batch * heads);
  lm_attn_fwd_kernel<<<grid, 256, lds, stream>>>(qkv, lens, ctx, lse, len, heads, causal, drop_key, drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("lm_attention_fwd");
  return 0;
}

extern "C" int smt_lm_attention_bwd(const float* qkv, const int* lens, const float* ctx, const float* lse, const float* dctx,
                                    float* dqkv, int batch, int len, int heads, int causal, uint32_t drop_key,
                                    uint32_t drop_thresh16, float drop_scale, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0 || len <= 0) return 0;
  SMT_CHECK_ARG(qkv && ctx && lse && dctx && dqkv, "smt_lm_attention_bwd: null pointer");
  SMT_CHECK_ARG(len <= LM_MAXL && heads >= 1, "smt_lm_attention_bwd: len must be <= %d (got %d)", LM_MAXL, len);
  SMT_CHECK_ARG((long long)batch * heads <= 65535, "smt_lm_attention_bwd: batch * heads must be <= 65535");
  const dim3 grid((len + AT_ROWS - 1) / AT_ROWS, batch * heads);
  const size_t lds_q = 2 * (size_t)len * AT_LD * sizeof(float);
  (void)hipFuncSetAttribute((const void*)lm_attn_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
  lm_attn_dq_kernel<<<grid, 256, lds_q, stream>>>(qkv, lens, ctx, lse, dctx, dqkv, len, heads, causal, drop_key, drop_thresh16,
                                                 drop_scale);
  SMT_CHECK_LAUNCH("lm_attention_dq");
  const size_t lds_k = (2 * (size_t)len * AT_LD + 2 * (size_t)len) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)lm_attn_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
  lm_attn_dkv_kernel<<<grid, 256, lds_k, stream>>>(qkv, lens, ctx, lse, dctx, dqkv, len, heads, causal, drop_key, drop_thresh16,
                                                  drop_scale);
  SMT_CHECK_LAUNCH("lm_attention_dkv");
  return 0;
}

extern "C" int smt_lm_add_ln_fwd(const float* x, const float* h, const float* gamma, const float* beta, float* y, float* stats,
                                 int64_t rows, int dim, float eps, uint32_t drop_key, uint32_t drop_thresh16, float drop_scale,
                                 smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows <= 0) return 0;
  SMT_CHECK_ARG((x || h) && gamma && beta && y && stats, "smt_lm_add_ln_fwd: null pointer");
  SMT_CHECK_ARG(dim % 64 == 0 && dim <= 2048, "smt_lm_add_ln_fwd: dim must be a multiple of 64 up to 2048 (got %d)", dim);
#define LM_CALL(P) lm_add_ln_fwd_kernel<P><<<(unsigned)((rows + 3) / 4), 256, 0, stream>>>(x, h, gamma, beta, y, stats, rows, eps, \
                                                                                     drop_key, drop_thresh16, drop_scale)
  LM_LN_DISPATCH(dim, LM_CALL)
#undef LM_CALL
  SMT_CHECK_LAUNCH("lm_add_ln_fwd");
  return 0;
}

extern "C" size_t smt_lm_add_ln_bwd_workspace_bytes(int64_t rows, int dim) {
  return (size_t)((rows + LN_RPB - 1) / LN_RPB) * 2 * dim * sizeof(float);
}

extern "C" int smt_lm_add_ln_bwd(const float* x, const float* h, const float* dy, const float* gamma, const float* stats,
                                 float* dx, float* dh, float* dgamma, float* dbeta, int64_t rows, int dim, uint32_t drop_key,
                                 uint32_t drop_thresh16, float drop_scale, void* workspace, size_t workspace_bytes,
                                 smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(dgamma && dbeta, "smt_lm_add_ln_bwd: null pointer");
  if (rows <= 0) {
    (void)hipMemsetAsync(dgamma, 0, dim * sizeof(float), stream);
    (void)hipMemsetAsync(dbeta, 0, dim * sizeof(float), stream);
    return 0;
  }
  SMT_CHECK_ARG((x || h) && dy && gamma && stats && workspace, "smt_lm_add_ln_bwd: null pointer");
  SMT_CHECK_ARG(dim % 64 == 0 && dim <= 2048, "smt_lm_add_ln_bwd: dim must be a multiple of 64 up to 2048 (got %d)", dim);
  SMT_CHECK_ARG(workspace_bytes >= smt_lm_add_ln_bwd_workspace_bytes(rows, dim), "smt_lm_add_ln_bwd: workspace too small");
  const int nblk = (int)((rows + LN_RPB - 1) / LN_RPB);
  float* part = (float*)workspace;
  // part [nblk][2][dim]: plane 0 = dgamma, plane 1 = dbeta; dgamma and dbeta must be adjacent for the one reduction
  SMT_CHECK_ARG(dbeta == dgamma + dim, "smt_lm_add_ln_bwd: dbeta must follow dgamma (one [2][dim] buffer)");
#define LM_CALL(P) lm_add_ln_bwd_kernel<P><<<nblk, 256, 8 * dim * sizeof(float), stream>>>(x, h, dy, gamma, stats, dx, dh, part, rows, \
                                                                                      drop_key, drop_thresh16, drop_scale)
  LM_LN_DISPATCH(dim, LM_CALL)
#undef LM_CALL
  SMT_CHECK_LAUNCH("lm_add_ln_bwd");
  lm_colsum_kernel<<<(2 * dim + 63) / 64, 256, 0, stream>>>(part, dgamma, nblk, 2 * dim);
  SMT_CHECK_LAUNCH("lm_colsum");
  return 0;
}

extern "C" int smt_lm_bias_relu_fwd(float* h, const float* bias, int64_t rows, int dim, uint32_t drop_key, uint32_t drop_thresh16,
                                    float drop_scale, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows <= 0) return 0;
  SMT_CHECK_ARG(h && bias, "smt_lm_bias_relu_fwd: null pointer");
  lm_bias_relu_fwd_kernel<<<lm_grid(rows * dim), 256, 0, stream>>>(h, bias, rows, dim, drop_key, drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("lm_bias_relu_fwd");
  return 0;
}

extern "C" size_t smt_lm_bias_relu_bwd_workspace_bytes(int64_t rows, int dim) { return (size_t)((rows + 15) / 16) * dim * sizeof(float); }

extern "C" int smt_lm_bias_relu_bwd(const float* a, const float* da, float* dh, float* dbias, int64_t rows, int dim, uint32_t drop_key,
                                    uint32_t drop_thresh16, float drop_scale, void* workspace, size_t workspace_bytes,
                                    smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(dbias, "smt_lm_bias_relu_bwd: null pointer");
  if (rows <= 0) { (void)hipMemsetAsync(dbias, 0, dim * sizeof(float), stream); return 0; }
  SMT_CHECK_ARG(a && da && dh && workspace, "smt_lm_bias_relu_bwd: null pointer");
  SMT_CHECK_ARG(workspace_bytes >= smt_lm_bias_relu_bwd_workspace_bytes(rows, dim), "smt_lm_bias_relu_bwd: workspace too small");
  const int nblk = (int)((rows + 15) / 16);
  lm_bias_relu_bwd_kernel<<<dim3(nblk, (dim + 255) / 256), 256, 0, stream>>>(a, da, dh, (float*)workspace, rows, dim, 16, drop_key, drop_thresh16, drop_scale);
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
