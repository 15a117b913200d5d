// GlowTTS (BASELINE.json configs[4]: "flow decoder + monotonic alignment search"): the pieces of the reference's
// models/glow_tts/{glow_tts,modules,submodules}.py that are not convolutions.  The convolutions (WN stack, 1x1 projections,
// feed-forward, prenet) run on the MFMA implicit-GEMM kernels of conv.hip; LayerNorm / ReLU+dropout on the kernels of lm.hip;
// the alignment search on mas.hip.  Everything here is fp32 on channels-last rows [B, T, C] with prefix row masks (int32 lens).
//
//   actnorm      z = (bias + exp(logs) x) mask                               submodules.py:237-253
//   invconv      4 x 4 mixing of channel quadruples (h C/2 + 2 j + k)         submodules.py:292-323
//   gate         tanh(a[:H]) sigmoid(a[H:]) of dropout(a)                     submodules.py:88-95, 213-220
//   coupling     z1 = (m + exp(logs) x1) mask, logdet = sum logs mask         submodules.py:383-405
//   attention    softmax(q k^T / sqrt(d) + relative keys, -1e4 fill), dropout, p v + relative values   submodules.py:463-512
//   prior_logp   log N(z_j; m_i, exp(logs_i)) for every (token i, frame j)    glow_tts.py:87-95
//   align        z_m = x_m[path], z_logs = x_logs[path], durations            glow_tts.py:99-101
//   loss         MLE + duration losses                                        glow_tts.py:115-121
// Reductions over rows are two-stage with a fixed order (bitwise reproducible); the only atomics are integer.
#include <math.h>

#include "conv_common.h"

namespace smt {

constexpr int GL_NT = 256;
// dropout factor of linear element index i (counter-based generator of include/smt_hip.h, "dropout")
__device__ __forceinline__ float gl_keep(unsigned long long i, unsigned key, unsigned thr, float scale) {
  return thr == 0 ? 1.f : (drop_keep(i, key, thr) ? scale : 0.f);
}
static inline unsigned gl_grid(long long n, int per = GL_NT) { return (unsigned)std::min<long long>(4096, std::max<long long>(1, (n + per - 1) / per)); }

// ------------------------------------------------------------------------------------------------ column reductions
// part[chunk][stride] -> out[col] = sum over chunks in index order, for col < ncol
__global__ __launch_bounds__(GL_NT) void gl_colsum_kernel(const float* __restrict__ part, int nchunks, int stride, int ncol,
                                                          float* __restrict__ out) {
  const int c = blockIdx.x * GL_NT + threadIdx.x;
  if (c >= ncol) return;
  float s = 0.f;
  for (int k = 0; k < nchunks; ++k) s += part[(size_t)k * stride + c];
  out[c] = s;
}

// the same in two stages for many chunks: slice s of GL_SLICES sums its contiguous range of chunks, then the slices are summed
constexpr int GL_SLICES = 64;
__global__ __launch_bounds__(GL_NT) void gl_colsum_slice_kernel(const float* __restrict__ part, int nchunks, int stride, int ncol,
                                                                float* __restrict__ scratch) {
  const int c = blockIdx.x * GL_NT + threadIdx.x, sl = blockIdx.y;
  if (c >= ncol) return;
  const int per = (nchunks + GL_SLICES - 1) / GL_SLICES, k0 = sl * per, k1 = min(nchunks, k0 + per);
  float s = 0.f;
  for (int k = k0; k < k1; ++k) s += part[(size_t)k * stride + c];
  scratch[(size_t)sl * ncol + c] = s;
}
static int gl_colsum_big(const float* part, int nchunks, int stride, int ncol, float* out, float* scratch, hipStream_t stream) {
  gl_colsum_slice_kernel<<<dim3((ncol + GL_NT - 1) / GL_NT, GL_SLICES), GL_NT, 0, stream>>>(part, nchunks, stride, ncol, scratch);
  SMT_CHECK_LAUNCH("glow_colsum_slice");
  gl_colsum_kernel<<<(ncol + GL_NT - 1) / GL_NT, GL_NT, 0, stream>>>(scratch, GL_SLICES, ncol, ncol, out);
  SMT_CHECK_LAUNCH("glow_colsum");
  return 0;
}

// ------------------------------------------------------------------------------------------------ ActNorm
__global__ __launch_bounds__(GL_NT) void gl_actnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ logs,
                                                               const float* __restrict__ bias, const int* __restrict__ lens,
                                                               float* __restrict__ z, int B, int T, int C, int reverse) {
  const long long total = (long long)B * T * C;
  for (long long e = (long long)blockIdx.x * GL_NT + threadIdx.x; e < total; e += (long long)gridDim.x * GL_NT) {
    const int c = (int)(e % C);
    const long long row = e / C;
    const int t = (int)(row % T), b = (int)(row / T);
    const float m = (lens && t >= lens[b]) ? 0.f : 1.f;
    z[e] = reverse ? (x[e] - bias[c]) * expf(-logs[c]) * m : (bias[c] + expf(logs[c]) * x[e]) * m;
  }
}

constexpr int GL_ROWS = 64;      // rows per workgroup of the column-reducing kernels
// dx = dz exp(logs) mask; partial dlogs[c] = sum dz x exp(logs) mask, dbias[c] = sum dz mask over this workgroup's rows
__global__ __launch_bounds__(GL_NT) void gl_actnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                               const float* __restrict__ logs, const int* __restrict__ lens,
                                                               float* __restrict__ dx, float* __restrict__ part, int B, int T, int C) {
  const long long rows = (long long)B * T, r0 = (long long)blockIdx.x * GL_ROWS;
  for (int c = threadIdx.x; c < C; c += GL_NT) {
    const float el = expf(logs[c]);
    float sl = 0.f, sb = 0.f;
    for (long long r = r0; r < min(rows, r0 + GL_ROWS); ++r) {
      const int t = (int)(r % T), b = (int)(r / T);
      const float m = (lens && t >= lens[b]) ? 0.f : 1.f;
      const float g = dz[r * C + c] * m;
      if (dx) dx[r * C + c] = g * el;
      sl += g * el * x[r * C + c];
      sb += g;
    }
    part[(size_t)blockIdx.x * 2 * C + c] = sl;
    part[(size_t)blockIdx.x * 2 * C + C + c] = sb;
  }
}

// ------------------------------------------------------------------------------------------------ InvConvNear
// channel of (half h, group j, k) = h C/2 + 2 j + k (n_split = 4: s = 2 h + k); z[s'] = sum_s w[s'][s] x[s]
__global__ __launch_bounds__(GL_NT) void gl_invconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const int* __restrict__ lens, float* __restrict__ z, int B, int T, int C,
                                                           int transpose) {
  __shared__ float ws[16];
  if (threadIdx.x < 16) ws[threadIdx.x] = transpose ? w[(threadIdx.x & 3) * 4 + (threadIdx.x >> 2)] : w[threadIdx.x];
  __syncthreads();
  const int G = C / 4;
  const long long total = (long long)B * T * G;
  for (long long e = (long long)blockIdx.x * GL_NT + threadIdx.x; e < total; e += (long long)gridDim.x * GL_NT) {
    const int j = (int)(e % G);
    const long long row = e / G;
    const int t = (int)(row % T), b = (int)(row / T);
    const float m = (lens && t >= lens[b]) ? 0.f : 1.f;
    const float* xr = x + row * C;
    float v[4] = {xr[2 * j], xr[2 * j + 1], xr[C / 2 + 2 * j], xr[C / 2 + 2 * j + 1]};
    float o[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) o[s] = (((ws[4 * s] * v[0] + ws[4 * s + 1] * v[1]) + ws[4 * s + 2] * v[2]) + ws[4 * s + 3] * v[3]) * m;
    float* zr = z + row * C;
    zr[2 * j] = o[0]; zr[2 * j + 1] = o[1]; zr[C / 2 + 2 * j] = o[2]; zr[C / 2 + 2 * j + 1] = o[3];
  }
}
// partial dW[s'][s] = sum over this workgroup's rows and groups of dz[s'] mask x[s]
__global__ __launch_bounds__(GL_NT) void gl_invconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                 const int* __restrict__ lens, float* __restrict__ part, int B, int T, int C) {
  __shared__ float red[GL_NT / 64][16];
  const int G = C / 4;
  const long long rows = (long long)B * T, r0 = (long long)blockIdx.x * GL_ROWS;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const long long n = min(rows, r0 + GL_ROWS) - r0;
  for (long long e = threadIdx.x; e < n * G; e += GL_NT) {
    const long long row = r0 + e / G;
    const int j = (int)(e % G);
    const int t = (int)(row % T), b = (int)(row / T);
    if (lens && t >= lens[b]) continue;
    const float* xr = x + row * C;
    const float* gr = dz + row * C;
    const float v[4] = {xr[2 * j], xr[2 * j + 1], xr[C / 2 + 2 * j], xr[C / 2 + 2 * j + 1]};
    const float g[4] = {gr[2 * j], gr[2 * j + 1], gr[C / 2 + 2 * j], gr[C / 2 + 2 * j + 1]};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[4 * a + s] += g[a] * v[s];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float s = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    float s = red[0][threadIdx.x];
    for (int wv = 1; wv < GL_NT / 64; ++wv) s += red[wv][threadIdx.x];
    part[(size_t)blockIdx.x * 16 + threadIdx.x] = s;
  }
}

// ------------------------------------------------------------------------------------------------ WN gate
__device__ __forceinline__ float gl_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }
// a [rows, 2H] (the in_layer's output, before its dropout) -> acts [rows, H]; dropout index = the element's linear index
__global__ __launch_bounds__(GL_NT) void gl_gate_fwd_kernel(const float* __restrict__ a, float* __restrict__ acts, long long rows, int H,
                                                            unsigned key, const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;
  const long long total = rows * H;
  for (long long e = (long long)blockIdx.x * GL_NT + threadIdx.x; e < total; e += (long long)gridDim.x * GL_NT) {
    const long long r = e / H;
    const int c = (int)(e % H);
    const unsigned long long it = (unsigned long long)r * 2 * H + c, is = it + H;
    const float tv = a[it] * gl_keep(it, key, thr, dscale), sv = a[is] * gl_keep(is, key, thr, dscale);
    acts[e] = tanhf(tv) * gl_sigmoid(sv);
  }
}
__global__ __launch_bounds__(GL_NT) void gl_gate_bwd_kernel(const float* __restrict__ a, const float* __restrict__ dacts,
                                                            float* __restrict__ da, long long rows, int H, unsigned key,
                                                            const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;
  const long long total = rows * H;
  for (long long e = (long long)blockIdx.x * GL_NT + threadIdx.x; e < total; e += (long long)gridDim.x * GL_NT) {
    const long long r = e / H;
    const int c = (int)(e % H);
    const unsigned long long it = (unsigned long long)r * 2 * H + c, is = it + H;
    const float kt = gl_keep(it, key, thr, dscale), ks = gl_keep(is, key, thr, dscale);
    const float th = tanhf(a[it] * kt), sg = gl_sigmoid(a[is] * ks), g = dacts[e];
    da[it] = g * sg * (1.f - th * th) * kt;
    da[is] = g * th * sg * (1.f - sg) * ks;
  }
}

__global__ __launch_bounds__(GL_NT) void gl_dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, unsigned key,
                                                           const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;
  for (long long e = (long long)blockIdx.x * GL_NT + threadIdx.x; e < n; e += (long long)gridDim.x * GL_NT)
    y[e] = x[e] * gl_keep((unsigned long long)e, key, thr, dscale);
}

// ------------------------------------------------------------------------------------------------ affine coupling
// out [rows, C] = (m | logs) of the `end` convolution, x [rows, C] = (x0 | x1) -> z = (x0 | (m + exp(logs) x1) mask);
// ldpart[b][chunk] = sum over this workgroup's rows of item b of logs mask (one workgroup never spans two items)
__global__ __launch_bounds__(GL_NT) void gl_coupling_fwd_kernel(const float* __restrict__ out, const float* __restrict__ x,
                                                                const int* __restrict__ lens, float* __restrict__ z,
                                                                float* __restrict__ ldpart, int T, int C, int chunks, int sigmoid_scale,
                                                                int reverse) {
  __shared__ float red[GL_NT / 64];
  const int b = blockIdx.x / chunks, ch = blockIdx.x % chunks, h = C / 2;
  const int t0 = ch * GL_ROWS, t1 = min(T, t0 + GL_ROWS);
  const int len = lens ? lens[b] : T;
  float s = 0.f;
  for (int e = threadIdx.x; e < (t1 - t0) * h; e += GL_NT) {
    const int t = t0 + e / h, c = e % h;
    const size_t row = ((size_t)b * T + t) * C;
    const float m = t < len ? 1.f : 0.f;
    float lg = out[row + h + c];
    if (sigmoid_scale) lg = logf(1e-6f + gl_sigmoid(lg + 2.f));
    z[row + c] = x[row + c];
    z[row + h + c] = reverse ? (x[row + h + c] - out[row + c]) * expf(-lg) * m : (out[row + c] + expf(lg) * x[row + h + c]) * m;
    s += lg * m;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && ldpart) ldpart[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
// dz [rows, C], dlogdet [B] -> dout = (dm | dlogs), dx = (dz0 | dz1 exp(logs) mask)
__global__ __launch_bounds__(GL_NT) void gl_coupling_bwd_kernel(const float* __restrict__ out, const float* __restrict__ x,
                                                                const float* __restrict__ dz, const float* __restrict__ dlogdet,
                                                                const int* __restrict__ lens, float* __restrict__ dout,
                                                                float* __restrict__ dx, int B, int T, int C, int sigmoid_scale) {
  const int h = C / 2;
  const long long total = (long long)B * T * h;
  for (long long e = (long long)blockIdx.x * GL_NT + threadIdx.x; e < total; e += (long long)gridDim.x * GL_NT) {
    const int c = (int)(e % h);
    const long long r = e / h;
    const int t = (int)(r % T), b = (int)(r / T);
    const size_t row = (size_t)r * C;
    const float m = (lens && t >= lens[b]) ? 0.f : 1.f;
    const float raw = out[row + h + c];
    float lg = raw, dl_draw = 1.f;
    if (sigmoid_scale) {
      const float sg = gl_sigmoid(raw + 2.f);
      lg = logf(1e-6f + sg);
      dl_draw = sg * (1.f - sg) / (1e-6f + sg);
    }
    const float el = expf(lg), g1 = dz[row + h + c] * m;
    dout[row + c] = g1;
    dout[row + h + c] = (g1 * el * x[row + h + c] + (dlogdet ? dlogdet[b] : 0.f) * m) * dl_draw;
    dx[row + c] = dz[row + c];
    dx[row + h + c] = g1 * el;
  }
}

// ------------------------------------------------------------------------------------------------ relative-position attention
// q, k, v [B, T, heads * D] (heads side by side), ek / ev [2 W + 1, D] (shared by the heads), lens [B].  One workgroup per
// (batch, head, query): scores over all T keys, masked_fill(-1e4) where query or key is padding (the reference fills, it does
// not exclude: a fully masked row is uniform over all T keys), softmax, dropout, context + relative values.
// pa [B, heads, T, T] keeps the softmax probabilities (before dropout) for the backward.
__global__ __launch_bounds__(GL_NT) void gl_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ ek,
                                                            const float* __restrict__ ev, const int* __restrict__ lens,
                                                            float* __restrict__ ctx, float* __restrict__ pa, int T, int heads, int D, int W,
                                                            unsigned key, const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;
  extern __shared__ float sm[];                // q row [D] | scores [T]
  float* qs = sm;
  float* sc = sm + D;
  __shared__ float red[GL_NT / 64];
  const int i = blockIdx.x, h = blockIdx.y % heads, b = blockIdx.y / heads;
  const int C = heads * D, len = lens ? lens[b] : T;
  const float inv = rsqrtf((float)D);
  const float* qrow = q + ((size_t)b * T + i) * C + h * D;
  for (int d = threadIdx.x; d < D; d += GL_NT) qs[d] = qrow[d];
  __syncthreads();
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < T; j += GL_NT) {
    const float* krow = k + ((size_t)b * T + j) * C + h * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(qs[d], krow[d], s);
    s *= inv;
    const int rel = j - i;
    if (rel >= -W && rel <= W) {
      const float* er = ek + (size_t)(rel + W) * D;
      float sr = 0.f;
      for (int d = 0; d < D; ++d) sr = fmaf(qs[d], er[d], sr);
      s += sr * inv;
    }
    if (i >= len || j >= len) s = -1e4f;
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float z = 0.f;
  for (int j = threadIdx.x; j < T; j += GL_NT) { const float e = expf(sc[j] - mx); sc[j] = e; z += e; }
  z = wave_sum(z);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = z;
  __syncthreads();
  z = ((red[0] + red[1]) + red[2]) + red[3];
  const float rz = 1.f / z;
  float* prow = pa + (((size_t)b * heads + h) * T + i) * T;
  for (int j = threadIdx.x; j < T; j += GL_NT) {
    const float p = sc[j] * rz;
    prow[j] = p;                               // the softmax itself (the backward recomputes the dropout factors)
    sc[j] = p * gl_keep((((unsigned long long)b * heads + h) * T + i) * T + j, key, thr, dscale);
  }
  __syncthreads();
  float* crow = ctx + ((size_t)b * T + i) * C + h * D;
  for (int d = threadIdx.x; d < D; d += GL_NT) {
    float o = 0.f;
    for (int j = 0; j < T; ++j) o = fmaf(sc[j], v[((size_t)b * T + j) * C + h * D + d], o);
    for (int rel = max(-W, -i); rel <= min(W, T - 1 - i); ++rel) o = fmaf(sc[i + rel], ev[(size_t)(rel + W) * D + d], o);
    crow[d] = o;
  }
}

// dP_ij = dctx_i . (v_j + ev[j-i]); dS = keep_scale * P (dP - sum_j P dP) with P the softmax (pa / keep); the gradient of the
// scores: dq_i += dS_ij (k_j + ek[j-i]) / sqrt(D).  One workgroup per (batch, head, query); writes dq and the score
// gradients ds [B, heads, T, T] (consumed by the key-side kernel), and per-query partials of dek.
__global__ __launch_bounds__(GL_NT) void gl_attn_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const float* __restrict__ ek,
                                                              const float* __restrict__ ev, const float* __restrict__ pa,
                                                              const float* __restrict__ dctx, float* __restrict__ dq,
                                                              float* __restrict__ ds, float* __restrict__ dek_part,
                                                              float* __restrict__ dev_part, int T, int heads, int D, int W,
                                                              unsigned key, const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;
  extern __shared__ float sm[];                // dctx row [D] | q row [D] | dS [T]
  float* gs = sm;
  float* qs = sm + D;
  float* dsr = sm + 2 * D;
  __shared__ float red[GL_NT / 64];
  const int i = blockIdx.x, h = blockIdx.y % heads, b = blockIdx.y / heads;
  const int C = heads * D;
  const float inv = rsqrtf((float)D);
  for (int d = threadIdx.x; d < D; d += GL_NT) {
    gs[d] = dctx[((size_t)b * T + i) * C + h * D + d];
    qs[d] = q[((size_t)b * T + i) * C + h * D + d];
  }
  __syncthreads();
  const float* prow = pa + (((size_t)b * heads + h) * T + i) * T;
  const unsigned long long e0 = (((unsigned long long)b * heads + h) * T + i) * T;
  float dot = 0.f;
  for (int j = threadIdx.x; j < T; j += GL_NT) {
    const float* vrow = v + ((size_t)b * T + j) * C + h * D;
    float dp = 0.f;
    for (int d = 0; d < D; ++d) dp = fmaf(gs[d], vrow[d], dp);
    const int rel = j - i;
    if (rel >= -W && rel <= W) {
      const float* er = ev + (size_t)(rel + W) * D;
      for (int d = 0; d < D; ++d) dp = fmaf(gs[d], er[d], dp);
    }
    dp *= gl_keep(e0 + j, key, thr, dscale);   // d (P keep) / dP
    dsr[j] = dp;
    dot += prow[j] * dp;
  }
  dot = wave_sum(dot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  dot = ((red[0] + red[1]) + red[2]) + red[3];
  float* dsrow = ds + (((size_t)b * heads + h) * T + i) * T;
  for (int j = threadIdx.x; j < T; j += GL_NT) {
    const float g = prow[j] * (dsr[j] - dot);  // softmax backward; masked_fill entries have P = 0 up to exp(-1e4 - max) = 0
    dsr[j] = g;
    dsrow[j] = g;
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += GL_NT) {
    float o = 0.f;
    for (int j = 0; j < T; ++j) o = fmaf(dsr[j], k[((size_t)b * T + j) * C + h * D + d], o);
    for (int rel = max(-W, -i); rel <= min(W, T - 1 - i); ++rel) o = fmaf(dsr[i + rel], ek[(size_t)(rel + W) * D + d], o);
    dq[((size_t)b * T + i) * C + h * D + d] = o * inv;
  }
  // per-query partials of the relative-embedding gradients: dek[r] += dS_{i,i+r} q_i / sqrt(D), dev[r] += P keep_{i,i+r} dctx_i
  float* dkp = dek_part + (((size_t)b * heads + h) * T + i) * (size_t)(2 * W + 1) * D;
  float* dvp = dev_part + (((size_t)b * heads + h) * T + i) * (size_t)(2 * W + 1) * D;
  for (int e = threadIdx.x; e < (2 * W + 1) * D; e += GL_NT) {
    const int r = e / D - W, d = e % D, j = i + r;
    const bool ok = j >= 0 && j < T;
    dkp[e] = ok ? dsr[j] * qs[d] * inv : 0.f;
    dvp[e] = ok ? prow[j] * gl_keep(e0 + j, key, thr, dscale) * gs[d] : 0.f;
  }
}
// key side: dk_j = sum_i dS_ij q_i / sqrt(D), dv_j = sum_i (P keep)_ij dctx_i.  One workgroup per (batch, head, key).
__global__ __launch_bounds__(GL_NT) void gl_attn_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ pa,
                                                               const float* __restrict__ ds, const float* __restrict__ dctx,
                                                               float* __restrict__ dk, float* __restrict__ dv, int T, int heads, int D,
                                                               unsigned key, const unsigned* __restrict__ key_dev, unsigned thr, float dscale) {
  if (key_dev) key = *key_dev;
  const int j = blockIdx.x, h = blockIdx.y % heads, b = blockIdx.y / heads;
  const int C = heads * D;
  const float inv = rsqrtf((float)D);
  const float* pcol = pa + (((size_t)b * heads + h) * T) * T + j;
  const float* dcol = ds + (((size_t)b * heads + h) * T) * T + j;
  for (int d = threadIdx.x; d < D; d += GL_NT) {
    float ok_ = 0.f, ov = 0.f;
    for (int i = 0; i < T; ++i) {
      ok_ = fmaf(dcol[(size_t)i * T], q[((size_t)b * T + i) * C + h * D + d], ok_);
      ov = fmaf(pcol[(size_t)i * T] * gl_keep((((unsigned long long)b * heads + h) * T + i) * T + j, key, thr, dscale),
                dctx[((size_t)b * T + i) * C + h * D + d], ov);
    }
    dk[((size_t)b * T + j) * C + h * D + d] = ok_ * inv;
    dv[((size_t)b * T + j) * C + h * D + d] = ov;
  }
}

// ------------------------------------------------------------------------------------------------ prior log-likelihood
// logp[b, i, j] = sum_d ( -0.5 log 2 pi - logs_id - 0.5 (z_jd - m_id)^2 exp(-2 logs_id) ), expanded as the reference does
// (glow_tts.py:90-95): four terms, so that the fp32 round-off is the reference's.
__global__ __launch_bounds__(GL_NT) void gl_prior_logp_kernel(const float* __restrict__ xm, const float* __restrict__ xlogs,
                                                              const float* __restrict__ z, float* __restrict__ logp, int Tx, int Ty, int D) {
  extern __shared__ float sm[];                // s_r [D] | m s_r [D]
  float* sr = sm;
  float* ms = sm + D;
  __shared__ float c14;
  const int i = blockIdx.x, b = blockIdx.y;
  const float* mrow = xm + ((size_t)b * Tx + i) * D;
  const float* lrow = xlogs ? xlogs + ((size_t)b * Tx + i) * D : nullptr;
  for (int d = threadIdx.x; d < D; d += GL_NT) {
    const float lg = lrow ? lrow[d] : 0.f;
    const float s = expf(-2.f * lg);
    sr[d] = s; ms[d] = mrow[d] * s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float p1 = 0.f, p4 = 0.f;
    for (int d = 0; d < D; ++d) {
      p1 += -0.5f * 1.8378770664093453f - (lrow ? lrow[d] : 0.f);
      p4 += -0.5f * (mrow[d] * mrow[d]) * sr[d];
    }
    c14 = p1 + p4;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < Ty; j += GL_NT) {
    const float* zr = z + ((size_t)b * Ty + j) * D;
    float p2 = 0.f, p3 = 0.f;
    for (int d = 0; d < D; ++d) {
      p2 = fmaf(sr[d], -0.5f * (zr[d] * zr[d]), p2);
      p3 = fmaf(ms[d], zr[d], p3);
    }
    logp[((size_t)b * Tx + i) * Ty + j] = (c14 + p2) + p3;
  }
}

// ------------------------------------------------------------------------------------------------ alignment -> frames
// path [B, Tx, Ty] (0/1, at most one token per frame): idx[b, j] = that token or -1; dur[b, i] = frames of token i
__global__ __launch_bounds__(GL_NT) void gl_align_index_kernel(const float* __restrict__ path, int* __restrict__ idx,
                                                               float* __restrict__ dur, int Tx, int Ty) {
  const int b = blockIdx.x;
  const float* pb = path + (size_t)b * Tx * Ty;
  for (int j = threadIdx.x; j < Ty; j += GL_NT) {
    int tok = -1;
    for (int i = 0; i < Tx; ++i) if (pb[(size_t)i * Ty + j] != 0.f) tok = i;
    idx[(size_t)b * Ty + j] = tok;
  }
  for (int i = threadIdx.x; i < Tx; i += GL_NT) {
    float s = 0.f;
    for (int j = 0; j < Ty; ++j) s += pb[(size_t)i * Ty + j];
    dur[(size_t)b * Tx + i] = s;
  }
}
// zf[b, j, :] = xf[b, idx[b, j], :] (zero where idx < 0)
__global__ __launch_bounds__(GL_NT) void gl_align_gather_kernel(const float* __restrict__ xf, const int* __restrict__ idx,
                                                                float* __restrict__ zf, int B, int Tx, int Ty, int D) {
  const long long total = (long long)B * Ty * D;
  for (long long e = (long long)blockIdx.x * GL_NT + threadIdx.x; e < total; e += (long long)gridDim.x * GL_NT) {
    const int d = (int)(e % D);
    const long long r = e / D;
    const int b = (int)(r / Ty);
    const int tok = idx[r];
    zf[e] = tok >= 0 ? xf[((size_t)b * Tx + tok) * D + d] : 0.f;
  }
}
// dxf[b, i, :] = sum over the frames j with idx[b, j] == i of dzf[b, j, :], in frame order
__global__ __launch_bounds__(GL_NT) void gl_align_scatter_kernel(const float* __restrict__ dzf, const int* __restrict__ idx,
                                                                 float* __restrict__ dxf, int B, int Tx, int Ty, int D) {
  const long long total = (long long)B * Tx * D;
  for (long long e = (long long)blockIdx.x * GL_NT + threadIdx.x; e < total; e += (long long)gridDim.x * GL_NT) {
    const int d = (int)(e % D);
    const long long r = e / D;
    const int i = (int)(r % Tx), b = (int)(r / Tx);
    float s = 0.f;
    for (int j = 0; j < Ty; ++j)
      if (idx[(size_t)b * Ty + j] == i) s += dzf[((size_t)b * Ty + j) * D + d];
    dxf[e] = s;
  }
}

// ------------------------------------------------------------------------------------------------ losses
// per-workgroup partials of sum z_logs, sum exp(-2 z_logs) (z - z_m)^2 over all (b, j, d); and of sum_{t < len} (logw - logw_dec)^2
__global__ __launch_bounds__(GL_NT) void gl_mle_part_kernel(const float* __restrict__ z, const float* __restrict__ zm,
                                                            const float* __restrict__ zl, long long n, float* __restrict__ part) {
  __shared__ float red[2][GL_NT / 64];
  float a = 0.f, q = 0.f;
  for (long long e = (long long)blockIdx.x * GL_NT + threadIdx.x; e < n; e += (long long)gridDim.x * GL_NT) {
    const float lg = zl ? zl[e] : 0.f, df = z[e] - zm[e];
    a += lg;
    q += expf(-2.f * lg) * (df * df);
  }
  a = wave_sum(a); q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    part[2 * blockIdx.x + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  }
}
// dz = coef exp(-2 zl) (z - zm), dzm = -dz, dzl = coef (1 - exp(-2 zl) (z - zm)^2); coef = g / N read from device memory
__global__ __launch_bounds__(GL_NT) void gl_mle_bwd_kernel(const float* __restrict__ z, const float* __restrict__ zm,
                                                           const float* __restrict__ zl, const float* __restrict__ coef, long long n,
                                                           float* __restrict__ dz, float* __restrict__ dzm, float* __restrict__ dzl) {
  const float c = *coef;
  for (long long e = (long long)blockIdx.x * GL_NT + threadIdx.x; e < n; e += (long long)gridDim.x * GL_NT) {
    const float lg = zl ? zl[e] : 0.f, df = z[e] - zm[e], w = expf(-2.f * lg);
    const float g = c * w * df;
    dz[e] = g;
    dzm[e] = -g;
    if (dzl) dzl[e] = c * (1.f - w * df * df);
  }
}
// logw_dec = log(1e-8 + dur) mask; one workgroup: out[0] = sum_{t < len} (logw - logw_dec)^2, diff[b, t] = masked difference
__global__ __launch_bounds__(1024) void gl_length_loss_kernel(const float* __restrict__ logw, const float* __restrict__ dur,
                                                              const int* __restrict__ lens, int B, int Tx, float* __restrict__ diff,
                                                              float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int e = threadIdx.x; e < B * Tx; e += 1024) {
    const int b = e / Tx, t = e % Tx;
    const float m = (lens && t >= lens[b]) ? 0.f : 1.f;
    const float df = (logw[e] - logf(1e-8f + dur[e])) * m;
    diff[e] = df;
    s += df * df;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += red[k];
    out[0] = t;
  }
}

// part[item][chunks] -> out[item] = sum over the chunks in index order
__global__ __launch_bounds__(GL_NT) void gl_item_sum_kernel(const float* __restrict__ part, int items, int chunks, float* __restrict__ out) {
  const int b = blockIdx.x * GL_NT + threadIdx.x;
  if (b >= items) return;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += part[(size_t)b * chunks + k];
  out[b] = s;
}
// part[n][2] -> out[0..1] (single thread, index order)
__global__ void gl_pair_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float a = 0.f, q = 0.f;
    for (int k = 0; k < n; ++k) { a += part[2 * k]; q += part[2 * k + 1]; }
    out[0] = a; out[1] = q;
  }
}

}  // namespace smt

using namespace smt;

extern "C" size_t smt_glow_reduce_workspace_bytes(int64_t rows, int cols) {
  return (size_t)((rows + GL_ROWS - 1) / GL_ROWS) * (size_t)cols * sizeof(float);
}

extern "C" int smt_glow_actnorm_fwd(const float* x, const float* logs, const float* bias, const int* lens, float* z, int batch, int t,
                                    int channels, int reverse, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((long long)batch * t * channels <= 0) return 0;
  SMT_CHECK_ARG(x && logs && bias && z, "smt_glow_actnorm_fwd: null pointer");
  gl_actnorm_fwd_kernel<<<gl_grid((long long)batch * t * channels), GL_NT, 0, stream>>>(x, logs, bias, lens, z, batch, t, channels, reverse);
  SMT_CHECK_LAUNCH("glow_actnorm_fwd");
  return 0;
}

extern "C" int smt_glow_actnorm_bwd(const float* x, const float* dz, const float* logs, const int* lens, float* dx, float* dlogs,
                                    float* dbias, int batch, int t, int channels, void* workspace, size_t workspace_bytes,
                                    smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const long long rows = (long long)batch * t;
  SMT_CHECK_ARG(x && dz && logs && dlogs && dbias && workspace, "smt_glow_actnorm_bwd: null pointer");
  SMT_CHECK_ARG(workspace_bytes >= smt_glow_reduce_workspace_bytes(rows, 2 * channels), "smt_glow_actnorm_bwd: workspace too small");
  const int chunks = (int)((rows + GL_ROWS - 1) / GL_ROWS);
  float* part = (float*)workspace;
  if (chunks > 0) {
    gl_actnorm_bwd_kernel<<<chunks, GL_NT, 0, stream>>>(x, dz, logs, lens, dx, part, batch, t, channels);
    SMT_CHECK_LAUNCH("glow_actnorm_bwd");
  }
  gl_colsum_kernel<<<(channels + GL_NT - 1) / GL_NT, GL_NT, 0, stream>>>(part, chunks, 2 * channels, channels, dlogs);   // first C columns
  SMT_CHECK_LAUNCH("glow_colsum");
  gl_colsum_kernel<<<(channels + GL_NT - 1) / GL_NT, GL_NT, 0, stream>>>(part + channels, chunks, 2 * channels, channels, dbias);
  SMT_CHECK_LAUNCH("glow_colsum");
  return 0;
}

extern "C" int smt_glow_invconv(const float* x, const float* weight, const int* lens, float* z, int batch, int t, int channels,
                                int transpose, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((long long)batch * t * channels <= 0) return 0;
  SMT_CHECK_ARG(x && weight && z && channels % 4 == 0, "smt_glow_invconv: null pointer / channels not a multiple of 4 (n_split = 4)");
  gl_invconv_kernel<<<gl_grid((long long)batch * t * channels / 4), GL_NT, 0, stream>>>(x, weight, lens, z, batch, t, channels, transpose);
  SMT_CHECK_LAUNCH("glow_invconv");
  return 0;
}

extern "C" int smt_glow_invconv_wgrad(const float* x, const float* dz, const int* lens, float* dweight, int batch, int t, int channels,
                                      void* workspace, size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const long long rows = (long long)batch * t;
  SMT_CHECK_ARG(x && dz && dweight && workspace && channels % 4 == 0, "smt_glow_invconv_wgrad: null pointer / bad channels");
  SMT_CHECK_ARG(workspace_bytes >= smt_glow_reduce_workspace_bytes(rows, 16), "smt_glow_invconv_wgrad: workspace too small");
  const int chunks = (int)((rows + GL_ROWS - 1) / GL_ROWS);
  if (chunks > 0) {
    gl_invconv_wgrad_kernel<<<chunks, GL_NT, 0, stream>>>(x, dz, lens, (float*)workspace, batch, t, channels);
    SMT_CHECK_LAUNCH("glow_invconv_wgrad");
  }
  gl_colsum_kernel<<<1, GL_NT, 0, stream>>>((const float*)workspace, chunks, 16, 16, dweight);
  SMT_CHECK_LAUNCH("glow_colsum");
  return 0;
}

extern "C" int smt_glow_gate_fwd(const float* a, float* acts, int64_t rows, int hidden, uint32_t drop_key, const uint32_t* drop_key_dev,
                                 uint32_t drop_thresh16, float drop_scale, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows * hidden <= 0) return 0;
  SMT_CHECK_ARG(a && acts, "smt_glow_gate_fwd: null pointer");
  gl_gate_fwd_kernel<<<gl_grid(rows * hidden), GL_NT, 0, stream>>>(a, acts, rows, hidden, drop_key, drop_key_dev, drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("glow_gate_fwd");
  return 0;
}

extern "C" int smt_glow_gate_bwd(const float* a, const float* dacts, float* da, int64_t rows, int hidden, uint32_t drop_key,
                                 const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows * hidden <= 0) return 0;
  SMT_CHECK_ARG(a && dacts && da, "smt_glow_gate_bwd: null pointer");
  gl_gate_bwd_kernel<<<gl_grid(rows * hidden), GL_NT, 0, stream>>>(a, dacts, da, rows, hidden, drop_key, drop_key_dev, drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("glow_gate_bwd");
  return 0;
}

extern "C" int smt_glow_coupling_fwd(const float* out, const float* x, const int* lens, float* z, float* logdet, int batch, int t,
                                     int channels, int sigmoid_scale, int reverse, void* workspace, size_t workspace_bytes,
                                     smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0 || t <= 0) return 0;
  SMT_CHECK_ARG(out && x && z && channels % 2 == 0, "smt_glow_coupling_fwd: null pointer / odd channel count");
  const int chunks = (t + GL_ROWS - 1) / GL_ROWS;
  SMT_CHECK_ARG(!logdet || (workspace && workspace_bytes >= (size_t)batch * chunks * sizeof(float)), "smt_glow_coupling_fwd: workspace too small");
  gl_coupling_fwd_kernel<<<batch * chunks, GL_NT, 0, stream>>>(out, x, lens, z, logdet ? (float*)workspace : nullptr, t, channels, chunks,
                                                             sigmoid_scale, reverse);
  SMT_CHECK_LAUNCH("glow_coupling_fwd");
  if (logdet) {
    gl_item_sum_kernel<<<(batch + GL_NT - 1) / GL_NT, GL_NT, 0, stream>>>((const float*)workspace, batch, chunks, logdet);
    SMT_CHECK_LAUNCH("glow_item_sum");
  }
  return 0;
}

extern "C" int smt_glow_coupling_bwd(const float* out, const float* x, const float* dz, const float* dlogdet, const int* lens,
                                     float* dout, float* dx, int batch, int t, int channels, int sigmoid_scale, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((long long)batch * t * channels <= 0) return 0;
  SMT_CHECK_ARG(out && x && dz && dout && dx && channels % 2 == 0, "smt_glow_coupling_bwd: null pointer / odd channel count");
  gl_coupling_bwd_kernel<<<gl_grid((long long)batch * t * channels / 2), GL_NT, 0, stream>>>(out, x, dz, dlogdet, lens, dout, dx, batch, t,
                                                                                          channels, sigmoid_scale);
  SMT_CHECK_LAUNCH("glow_coupling_bwd");
  return 0;
}

static size_t gl_attn_lds(int t, int d, int rows) { return (size_t)(rows * d + t) * sizeof(float); }

extern "C" int smt_glow_attention_fwd(const float* q, const float* k, const float* v, const float* emb_rel_k, const float* emb_rel_v,
                                      const int* lens, float* ctx, float* probs, int batch, int t, int heads, int head_dim, int window,
                                      uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale,
                                      smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0 || t <= 0) return 0;
  SMT_CHECK_ARG(q && k && v && emb_rel_k && emb_rel_v && ctx && probs, "smt_glow_attention_fwd: null pointer");
  SMT_CHECK_ARG(heads >= 1 && head_dim >= 1 && window >= 0 && (long long)batch * heads <= 65535, "smt_glow_attention_fwd: bad sizes");
  const size_t lds = gl_attn_lds(t, head_dim, 1);
  SMT_CHECK_ARG(lds <= 60 * 1024, "smt_glow_attention_fwd: t = %d keys need %zu B of LDS (limit 60 KiB)", t, lds);
  gl_attn_fwd_kernel<<<dim3(t, batch * heads), GL_NT, lds, stream>>>(q, k, v, emb_rel_k, emb_rel_v, lens, ctx, probs, t, heads, head_dim,
                                                                   window, drop_key, drop_key_dev, drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("glow_attention_fwd");
  return 0;
}

extern "C" size_t smt_glow_attention_bwd_workspace_bytes(int batch, int t, int heads, int head_dim, int window) {
  // score gradients [B, heads, T, T] + per-query partials of the two relative-embedding gradients + the reduction's slices
  return ((size_t)batch * heads * t * t + 2 * (size_t)batch * heads * t * (2 * window + 1) * head_dim +
          (size_t)GL_SLICES * (2 * window + 1) * head_dim) * sizeof(float);
}

extern "C" int smt_glow_attention_bwd(const float* q, const float* k, const float* v, const float* emb_rel_k, const float* emb_rel_v,
                                      const float* probs, const float* dctx, float* dq, float* dk, float* dv, float* demb_rel_k,
                                      float* demb_rel_v, int batch, int t, int heads, int head_dim, int window, uint32_t drop_key,
                                      const uint32_t* drop_key_dev, uint32_t drop_thresh16, float drop_scale, void* workspace,
                                      size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(q && k && v && emb_rel_k && emb_rel_v && probs && dctx && dq && dk && dv && demb_rel_k && demb_rel_v && workspace,
                "smt_glow_attention_bwd: null pointer");
  SMT_CHECK_ARG(workspace_bytes >= smt_glow_attention_bwd_workspace_bytes(batch, t, heads, head_dim, window), "smt_glow_attention_bwd: workspace too small");
  const int nrel = (2 * window + 1) * head_dim;
  float* ds = (float*)workspace;
  float* dkp = ds + (size_t)batch * heads * t * t;
  float* dvp = dkp + (size_t)batch * heads * t * nrel;
  if (batch > 0 && t > 0) {
    const size_t lds = gl_attn_lds(t, head_dim, 2);
    SMT_CHECK_ARG(lds <= 60 * 1024, "smt_glow_attention_bwd: t = %d keys need %zu B of LDS (limit 60 KiB)", t, lds);
    gl_attn_bwd_q_kernel<<<dim3(t, batch * heads), GL_NT, lds, stream>>>(q, k, v, emb_rel_k, emb_rel_v, probs, dctx, dq, ds, dkp, dvp, t, heads,
                                                                       head_dim, window, drop_key, drop_key_dev, drop_thresh16, drop_scale);
    SMT_CHECK_LAUNCH("glow_attention_bwd_q");
    gl_attn_bwd_kv_kernel<<<dim3(t, batch * heads), GL_NT, 0, stream>>>(q, probs, ds, dctx, dk, dv, t, heads, head_dim, drop_key, drop_key_dev,
                                                                     drop_thresh16, drop_scale);
    SMT_CHECK_LAUNCH("glow_attention_bwd_kv");
  }
  // the relative embeddings are shared by batch items, heads and queries: fixed-order sums of the per-query partials
  float* scratch = dvp + (size_t)batch * heads * t * nrel;
  int rc = gl_colsum_big(dkp, batch * heads * t, nrel, nrel, demb_rel_k, scratch, stream);
  if (rc) return rc;
  return gl_colsum_big(dvp, batch * heads * t, nrel, nrel, demb_rel_v, scratch, stream);
}

extern "C" int smt_glow_prior_logp(const float* x_m, const float* x_logs, const float* z, float* logp, int batch, int t_x, int t_y,
                                   int dim, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0 || t_x <= 0 || t_y <= 0) return 0;
  SMT_CHECK_ARG(x_m && z && logp && dim >= 1 && batch <= 65535, "smt_glow_prior_logp: null pointer / bad sizes");
  gl_prior_logp_kernel<<<dim3(t_x, batch), GL_NT, 2 * (size_t)dim * sizeof(float), stream>>>(x_m, x_logs, z, logp, t_x, t_y, dim);
  SMT_CHECK_LAUNCH("glow_prior_logp");
  return 0;
}

extern "C" int smt_glow_align_index(const float* path, int* idx, float* durations, int batch, int t_x, int t_y, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0) return 0;
  SMT_CHECK_ARG(path && idx && durations, "smt_glow_align_index: null pointer");
  gl_align_index_kernel<<<batch, GL_NT, 0, stream>>>(path, idx, durations, t_x, t_y);
  SMT_CHECK_LAUNCH("glow_align_index");
  return 0;
}

extern "C" int smt_glow_align_gather(const float* x, const int* idx, float* z, int batch, int t_x, int t_y, int dim, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((long long)batch * t_y * dim <= 0) return 0;
  SMT_CHECK_ARG(x && idx && z, "smt_glow_align_gather: null pointer");
  gl_align_gather_kernel<<<gl_grid((long long)batch * t_y * dim), GL_NT, 0, stream>>>(x, idx, z, batch, t_x, t_y, dim);
  SMT_CHECK_LAUNCH("glow_align_gather");
  return 0;
}

extern "C" int smt_glow_align_scatter(const float* dz, const int* idx, float* dx, int batch, int t_x, int t_y, int dim, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((long long)batch * t_x * dim <= 0) return 0;
  SMT_CHECK_ARG(dz && idx && dx, "smt_glow_align_scatter: null pointer");
  gl_align_scatter_kernel<<<gl_grid((long long)batch * t_x * dim), GL_NT, 0, stream>>>(dz, idx, dx, batch, t_x, t_y, dim);
  SMT_CHECK_LAUNCH("glow_align_scatter");
  return 0;
}

extern "C" size_t smt_glow_mle_workspace_bytes(int64_t n) { return (size_t)gl_grid(n) * 2 * sizeof(float); }

extern "C" int smt_glow_mle_sums(const float* z, const float* z_m, const float* z_logs, int64_t n, float* sums, void* workspace,
                                 size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(z && z_m && sums && workspace && workspace_bytes >= smt_glow_mle_workspace_bytes(n), "smt_glow_mle_sums: null pointer / workspace");
  const unsigned grid = gl_grid(n);
  gl_mle_part_kernel<<<grid, GL_NT, 0, stream>>>(z, z_m, z_logs, n, (float*)workspace);
  SMT_CHECK_LAUNCH("glow_mle_part");
  gl_pair_sum_kernel<<<1, 64, 0, stream>>>((const float*)workspace, (int)grid, sums);
  SMT_CHECK_LAUNCH("glow_pair_sum");
  return 0;
}

extern "C" int smt_glow_mle_bwd(const float* z, const float* z_m, const float* z_logs, const float* coef, int64_t n, float* dz, float* dz_m,
                                float* dz_logs, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0) return 0;
  SMT_CHECK_ARG(z && z_m && coef && dz && dz_m, "smt_glow_mle_bwd: null pointer");
  gl_mle_bwd_kernel<<<gl_grid(n), GL_NT, 0, stream>>>(z, z_m, z_logs, coef, n, dz, dz_m, dz_logs);
  SMT_CHECK_LAUNCH("glow_mle_bwd");
  return 0;
}

extern "C" int smt_glow_length_loss(const float* logw, const float* durations, const int* lens, int batch, int t_x, float* diff, float* sum,
                                    smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(logw && durations && diff && sum, "smt_glow_length_loss: null pointer");
  gl_length_loss_kernel<<<1, 1024, 0, stream>>>(logw, durations, lens, batch, t_x, diff, sum);
  SMT_CHECK_LAUNCH("glow_length_loss");
  return 0;
}

extern "C" int smt_glow_dropout(const float* x, float* y, int64_t n, uint32_t drop_key, const uint32_t* drop_key_dev, uint32_t drop_thresh16,
                                float drop_scale, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0) return 0;
  SMT_CHECK_ARG(x && y, "smt_glow_dropout: null pointer");
  gl_dropout_kernel<<<gl_grid(n), GL_NT, 0, stream>>>(x, y, n, drop_key, drop_key_dev, drop_thresh16, drop_scale);
  SMT_CHECK_LAUNCH("glow_dropout");
  return 0;
}
