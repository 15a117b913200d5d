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
// so the backward kernels recompute them.  Attention: one workgroup per (batch, head); K and V of the head sit in LDS,
// a thread owns one query row (workgroup = L rounded up to whole waves) and walks its causal prefix with an online softmax -- at L = 258, dh = 32 a head is
// 2 MFLOP, far below anything worth tiling for the matrix pipe; the layer's time is in the GEMMs and launches.
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
// masked scores).  Key j is visible to query i iff (j <= i or not causal) and j < lens[b].  Attention-weight dropout: element index
// ((b H + h) L + i) L + j.
constexpr int LM_DH = 32, LM_MAXL = 512;

__global__ __launch_bounds__(512) void lm_attn_fwd_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                          float* __restrict__ ctx, float* __restrict__ lse, int L, int H,
                                                          int causal, unsigned key, unsigned thr, float dscale) {
  extern __shared__ float sm[];                               // K [L][33] | V [L][33]
  float* ks = sm;
  float* vs = sm + (size_t)L * 33;
  const int b = blockIdx.x / H, h = blockIdx.x % H, d = H * LM_DH;
  const float* base = qkv + (size_t)b * L * 3 * d;
  for (int f = threadIdx.x; f < L * LM_DH; f += blockDim.x) {
    const int j = f / LM_DH, c = f % LM_DH;
    ks[j * 33 + c] = base[(size_t)j * 3 * d + d + h * LM_DH + c];
    vs[j * 33 + c] = base[(size_t)j * 3 * d + 2 * d + h * LM_DH + c];
  }
  __syncthreads();
  const int len = lens ? min(lens[b], L) : L;
  const float sc = rsqrtf((float)LM_DH);
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    float q[LM_DH], acc[LM_DH];
#pragma unroll
    for (int c = 0; c < LM_DH; ++c) { q[c] = base[(size_t)i * 3 * d + h * LM_DH + c] * sc; acc[c] = 0.f; }
    float m = -INFINITY, z = 0.f;
    const int jn = causal ? min(i + 1, len) : len;
    const unsigned long long e0 = (((unsigned long long)b * H + h) * L + i) * L;
    for (int j = 0; j < jn; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < LM_DH; ++c) s = fmaf(q[c], ks[j * 33 + c], s);
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn), p = __expf(s - mn);
      z = z * corr + p;
      const float pk = p * lm_keep(e0 + j, key, thr, dscale);
#pragma unroll
      for (int c = 0; c < LM_DH; ++c) acc[c] = acc[c] * corr + pk * vs[j * 33 + c];
      m = mn;
    }
    const float inv = jn > 0 ? 1.f / z : 0.f;
#pragma unroll
    for (int c = 0; c < LM_DH; ++c) ctx[((size_t)b * L + i) * d + h * LM_DH + c] = acc[c] * inv;
    lse[((size_t)b * H + h) * L + i] = jn > 0 ? m + __logf(z) : 0.f;
  }
}

// dqkv from dctx: P_ij = exp(s_ij - lse_i); dPd_ij = dctx_i . v_j; delta_i = sum_j P_ij keep_ij dPd_ij = dctx_i . ctx_i;
// dS_ij = P_ij (keep_ij dPd_ij - delta_i); dq_i = sum_j dS_ij k_j / sqrt(dh); dk_j = sum_i dS_ij q_i / sqrt(dh);
// dv_j = sum_i P_ij keep_ij dctx_i.  Pass 1: a thread per query (dq); pass 2: a thread per key (dk, dv).
__global__ __launch_bounds__(512) void lm_attn_bwd_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                          const float* __restrict__ ctx, const float* __restrict__ lse,
                                                          const float* __restrict__ dctx, float* __restrict__ dqkv, int L, int H,
                                                          int causal, unsigned key, unsigned thr, float dscale) {
  extern __shared__ float sm[];                               // A [L][33] | Bm [L][33] | delta [L] | lse [L]
  float* as = sm;
  float* bs = sm + (size_t)L * 33;
  float* delta = bs + (size_t)L * 33;
  float* ls = delta + L;
  const int b = blockIdx.x / H, h = blockIdx.x % H, d = H * LM_DH;
  const float* base = qkv + (size_t)b * L * 3 * d;
  float* dbase = dqkv + (size_t)b * L * 3 * d;
  const int len = lens ? min(lens[b], L) : L;
  const float sc = rsqrtf((float)LM_DH);
  // pass 1: K, V in LDS; thread per query
  for (int f = threadIdx.x; f < L * LM_DH; f += blockDim.x) {
    const int j = f / LM_DH, c = f % LM_DH;
    as[j * 33 + c] = base[(size_t)j * 3 * d + d + h * LM_DH + c];
    bs[j * 33 + c] = base[(size_t)j * 3 * d + 2 * d + h * LM_DH + c];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    float q[LM_DH], go[LM_DH], dq[LM_DH];
    float dl = 0.f;
#pragma unroll
    for (int c = 0; c < LM_DH; ++c) {
      q[c] = base[(size_t)i * 3 * d + h * LM_DH + c] * sc;
      go[c] = dctx[((size_t)b * L + i) * d + h * LM_DH + c];
      dl = fmaf(go[c], ctx[((size_t)b * L + i) * d + h * LM_DH + c], dl);
      dq[c] = 0.f;
    }
    const float li = lse[((size_t)b * H + h) * L + i];
    delta[i] = dl; ls[i] = li;
    const int jn = causal ? min(i + 1, len) : len;
    const unsigned long long e0 = (((unsigned long long)b * H + h) * L + i) * L;
    for (int j = 0; j < jn; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < LM_DH; ++c) { s = fmaf(q[c], as[j * 33 + c], s); dp = fmaf(go[c], bs[j * 33 + c], dp); }
      const float p = __expf(s - li);
      const float ds = p * (lm_keep(e0 + j, key, thr, dscale) * dp - dl);
#pragma unroll
      for (int c = 0; c < LM_DH; ++c) dq[c] = fmaf(ds, as[j * 33 + c], dq[c]);
    }
#pragma unroll
    for (int c = 0; c < LM_DH; ++c) dbase[(size_t)i * 3 * d + h * LM_DH + c] = dq[c] * sc;
  }
  __syncthreads();
  // pass 2: Q (scaled), dctx in LDS; thread per key
  for (int f = threadIdx.x; f < L * LM_DH; f += blockDim.x) {
    const int i = f / LM_DH, c = f % LM_DH;
    as[i * 33 + c] = base[(size_t)i * 3 * d + h * LM_DH + c] * sc;
    bs[i * 33 + c] = dctx[((size_t)b * L + i) * d + h * LM_DH + c];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < L; j += blockDim.x) {
    float kk[LM_DH], vv[LM_DH], dk[LM_DH], dv[LM_DH];
#pragma unroll
    for (int c = 0; c < LM_DH; ++c) {
      kk[c] = base[(size_t)j * 3 * d + d + h * LM_DH + c];
      vv[c] = base[(size_t)j * 3 * d + 2 * d + h * LM_DH + c];
      dk[c] = 0.f; dv[c] = 0.f;
    }
    if (j < len) {
      for (int i = causal ? j : 0; i < L; ++i) {              // queries that see key j
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < LM_DH; ++c) { s = fmaf(as[i * 33 + c], kk[c], s); dp = fmaf(bs[i * 33 + c], vv[c], dp); }
        const float p = __expf(s - ls[i]);
        const float kp = lm_keep((((unsigned long long)b * H + h) * L + i) * L + j, key, thr, dscale);
        const float ds = p * (kp * dp - delta[i]);
        const float pv = p * kp;
#pragma unroll
        for (int c = 0; c < LM_DH; ++c) { dk[c] = fmaf(ds, as[i * 33 + c], dk[c]); dv[c] = fmaf(pv, bs[i * 33 + c], dv[c]); }
      }
    }
#pragma unroll
    for (int c = 0; c < LM_DH; ++c) {
      dbase[(size_t)j * 3 * d + d + h * LM_DH + c] = dk[c];    // q in LDS already carries 1/sqrt(dh)
      dbase[(size_t)j * 3 * d + 2 * d + h * LM_DH + c] = dv[c];
    }
  }
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
// dh = dpre * mask (if dh); dgamma / dbeta partials per workgroup (4 rows) -> part [nwg][2][C]
template <int PER>
__global__ __launch_bounds__(256) void lm_add_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ hh,
                                                            const float* __restrict__ dy, const float* __restrict__ gamma,
                                                            const float* __restrict__ stats, float* __restrict__ dx,
                                                            float* __restrict__ dh, float* __restrict__ part, long long rows,
                                                            unsigned key, unsigned thr, float dscale) {
  constexpr int C = PER * 64;
  extern __shared__ float red[];                              // [4][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + wave;
  const bool live = row < rows;
  float xh[PER], g[PER];
  float sg = 0.f, sgx = 0.f;
  const float mean = live ? stats[2 * row] : 0.f, rstd = live ? stats[2 * row + 1] : 0.f;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int c = lane + 64 * q;
    float xv = 0.f, gv = 0.f, dyv = 0.f;
    if (live) {
      const long long e = row * C + c;
      dyv = dy[e];
      const float pre = (x ? x[e] : 0.f) + (hh ? hh[e] * lm_keep((unsigned long long)e, key, thr, dscale) : 0.f);
      xv = (pre - mean) * rstd;
      gv = dyv * gamma[c];
    }
    xh[q] = xv; g[q] = gv; sg += gv; sgx = fmaf(gv, xv, sgx);
    red[(wave * 2 + 0) * C + c] = dyv * xv;
    red[(wave * 2 + 1) * C + c] = dyv;
  }
  sg = wave_sum(sg) / (float)C; sgx = wave_sum(sgx) / (float)C;
  if (live) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const long long e = row * C + lane + 64 * q;
      const float dpre = rstd * (g[q] - sg - xh[q] * sgx);
      if (dx) dx[e] = dpre;
      if (dh) dh[e] = dpre * lm_keep((unsigned long long)e, key, thr, dscale);
    }
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
// fixed-order column sums of part [n][2][C] (or [n][C] with planes = 1) -> out [planes][C]
__global__ __launch_bounds__(256) void lm_colsum_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int planes,
                                                        int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= planes * C) return;
  float s = 0.f;
#pragma unroll 8
  for (int i = 0; i < n; ++i) s += part[(size_t)i * planes * C + c];
  out[c] = s;
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
  for (int c = threadIdx.x; c < C; c += 256) {
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
  const size_t lds = 2 * (size_t)len * 33 * sizeof(float);
  (void)hipFuncSetAttribute((const void*)lm_attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  lm_attn_fwd_kernel<<<batch * heads, std::min(512, (len + 63) / 64 * 64), lds, stream>>>(qkv, lens, ctx, lse, len, heads, causal, drop_key, drop_thresh16, drop_scale);
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
  const size_t lds = (2 * (size_t)len * 33 + 2 * (size_t)len) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)lm_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  lm_attn_bwd_kernel<<<batch * heads, std::min(512, (len + 63) / 64 * 64), lds, stream>>>(qkv, lens, ctx, lse, dctx, dqkv, len, heads, causal, drop_key,
                                                          drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("lm_attention_bwd");
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

extern "C" size_t smt_lm_add_ln_bwd_workspace_bytes(int64_t rows, int dim) { return (size_t)((rows + 3) / 4) * 2 * dim * sizeof(float); }

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
  const int nblk = (int)((rows + 3) / 4);
  float* part = (float*)workspace;
  // part [nblk][2][dim]: plane 0 = dgamma, plane 1 = dbeta; dgamma and dbeta must be adjacent for the one reduction
  SMT_CHECK_ARG(dbeta == dgamma + dim, "smt_lm_add_ln_bwd: dbeta must follow dgamma (one [2][dim] buffer)");
#define LM_CALL(P) lm_add_ln_bwd_kernel<P><<<nblk, 256, 8 * dim * sizeof(float), stream>>>(x, h, dy, gamma, stats, dx, dh, part, rows, \
                                                                                      drop_key, drop_thresh16, drop_scale)
  LM_LN_DISPATCH(dim, LM_CALL)
#undef LM_CALL
  SMT_CHECK_LAUNCH("lm_add_ln_bwd");
  lm_colsum_kernel<<<(2 * dim + 255) / 256, 256, 0, stream>>>(part, dgamma, nblk, 2, dim);
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
  lm_bias_relu_bwd_kernel<<<nblk, 256, 0, stream>>>(a, da, dh, (float*)workspace, rows, dim, 16, drop_key, drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("lm_bias_relu_bwd");
  lm_colsum_kernel<<<(dim + 255) / 256, 256, 0, stream>>>((const float*)workspace, dbias, nblk, 1, dim);
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
