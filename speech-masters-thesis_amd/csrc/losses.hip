// MultiNormReconstructionLoss (reference models/vqvae/losses.py:58-80) as two kernels:
//   l1 * mean|d| + l2 * mean d^2 + linf * sum_j mean_b topk_j(d^2),   d = (y - yh) * mask,  on [B, T] signals.
// The reference materialises d^2 and runs torch.topk (k = 2048 of 145,408 per clip).  Here one workgroup per clip finds
// the EXACT k-th largest d^2 by a three-level radix select on the float bits (d^2 >= 0, so uint order == float order),
// recomputing d^2 from y / yh on every pass (nothing but 8 floats per clip is written), then sums the values above the
// threshold in fp64: topk_sum = sum_{d^2 > tau} d^2 + (k - count_gt) * tau.  The backward is one elementwise pass.
// HBM-bound: 8 B/sample per pass, five passes that hit L2 after the first (a clip is 1.1 MB).
#include <algorithm>

#include "smt_common.h"

namespace smt {

constexpr int RL_NT = 1024;

__device__ __forceinline__ float rl_sq(const float* __restrict__ y, const float* __restrict__ yh, int i, int len, float* d_out) {
  const float d = i < len ? y[i] - yh[i] : 0.f;
  if (d_out) *d_out = d;
  return d * d;
}

// block-wide: given hist[nbins] (counts) find the highest bin b with suffix-count(b) >= need; returns b and the count
// strictly above b.  nbins <= 2048, 1024 threads; scratch must hold nbins ints.
__device__ __forceinline__ void rl_find_bin(const int* hist, int* scratch, int nbins, int need, int* bin_out, int* above_out) {
  // suffix sums by a Hillis-Steele scan over the reversed array (two bins per thread)
  for (int i = threadIdx.x; i < nbins; i += RL_NT) scratch[i] = hist[nbins - 1 - i];
  __syncthreads();
  for (int off = 1; off < nbins; off <<= 1) {
    int v0 = 0, v1 = 0;
    const int i0 = threadIdx.x, i1 = threadIdx.x + RL_NT;
    if (i0 < nbins && i0 >= off) v0 = scratch[i0 - off];
    if (i1 < nbins && i1 >= off) v1 = scratch[i1 - off];
    __syncthreads();
    if (i0 < nbins) scratch[i0] += v0;
    if (i1 < nbins) scratch[i1] += v1;
    __syncthreads();
  }
  // scratch[i] = number of elements in bins >= nbins-1-i; the wanted bin is the first i with scratch[i] >= need
  for (int i = threadIdx.x; i < nbins; i += RL_NT) {
    const int incl = scratch[i], excl = i ? scratch[i - 1] : 0;
    if (incl >= need && excl < need) { *bin_out = nbins - 1 - i; *above_out = excl; }
  }
  __syncthreads();
}

__global__ __launch_bounds__(RL_NT) void recon_loss_fwd_kernel(const float* __restrict__ y, const float* __restrict__ yh,
                                                               const int* __restrict__ lens, int T, int topk,
                                                               float* __restrict__ stats) {
  __shared__ int hist[2048];
  __shared__ int scratch[2048];
  __shared__ double red[3][16];
  __shared__ int s_bin, s_above;
  const int b = blockIdx.x;
  const float* yr = y + (size_t)b * T;
  const float* yhr = yh + (size_t)b * T;
  const int len = lens ? min(lens[b], T) : T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // pass A: bits [31:21] + the two plain sums
  for (int i = threadIdx.x; i < 2048; i += RL_NT) hist[i] = 0;
  __syncthreads();
  double s_sq = 0.0, s_abs = 0.0;
  for (int i = threadIdx.x; i < T; i += RL_NT) {
    float d;
    const float q = rl_sq(yr, yhr, i, len, &d);
    s_sq += q; s_abs += fabsf(d);
    atomicAdd(&hist[__float_as_uint(q) >> 21], 1);
  }
  s_sq = wave_sum_d(s_sq); s_abs = wave_sum_d(s_abs);
  if (lane == 0) { red[0][wave] = s_sq; red[1][wave] = s_abs; }
  __syncthreads();
  rl_find_bin(hist, scratch, 2048, topk, &s_bin, &s_above);
  const unsigned b1 = (unsigned)s_bin;
  int need = topk - s_above, gt = s_above;
  __syncthreads();

  // pass B: bits [20:10] of the elements in bin b1
  for (int i = threadIdx.x; i < 2048; i += RL_NT) hist[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < T; i += RL_NT) {
    const unsigned u = __float_as_uint(rl_sq(yr, yhr, i, len, nullptr));
    if ((u >> 21) == b1) atomicAdd(&hist[(u >> 10) & 2047], 1);
  }
  __syncthreads();
  rl_find_bin(hist, scratch, 2048, need, &s_bin, &s_above);
  const unsigned b2 = (unsigned)s_bin;
  need -= s_above; gt += s_above;
  __syncthreads();

  // pass C: bits [9:0]
  for (int i = threadIdx.x; i < 1024; i += RL_NT) hist[i] = 0;
  __syncthreads();
  const unsigned prefix = (b1 << 11) | b2;
  for (int i = threadIdx.x; i < T; i += RL_NT) {
    const unsigned u = __float_as_uint(rl_sq(yr, yhr, i, len, nullptr));
    if ((u >> 10) == prefix) atomicAdd(&hist[u & 1023], 1);
  }
  __syncthreads();
  rl_find_bin(hist, scratch, 1024, need, &s_bin, &s_above);
  const unsigned tau_bits = (prefix << 10) | (unsigned)s_bin;
  gt += s_above;                                            // elements strictly greater than tau
  const int ties = hist[s_bin];
  const float tau = __uint_as_float(tau_bits);
  __syncthreads();

  // pass D: fp64 sum of the values above the threshold
  double s_top = 0.0;
  for (int i = threadIdx.x; i < T; i += RL_NT) {
    const float q = rl_sq(yr, yhr, i, len, nullptr);
    if (__float_as_uint(q) > tau_bits) s_top += q;
  }
  s_top = wave_sum_d(s_top);
  if (lane == 0) red[2][wave] = s_top;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, c = 0, e = 0;
    for (int w = 0; w < 16; ++w) { a += red[0][w]; c += red[1][w]; e += red[2][w]; }
    float* o = stats + (size_t)b * 8;
    o[0] = (float)a;                                        // sum d^2
    o[1] = (float)c;                                        // sum |d|
    o[2] = (float)(e + (double)(topk - gt) * (double)tau);  // sum of the k largest d^2
    o[3] = tau;
    o[4] = (float)gt;
    o[5] = (float)ties;
    o[6] = 0.f; o[7] = 0.f;
  }
}

// dyh = -( c_l1 * sign(d) + c_l2 * d + c_inf * d * w ),  w = 1 above tau, (k - gt) / ties at tau (any split of the tied
// elements is a valid subgradient; real signals have no ties, masked samples have d = 0), 0 below.
__global__ __launch_bounds__(256) void recon_loss_bwd_kernel(const float* __restrict__ y, const float* __restrict__ yh,
                                                             const int* __restrict__ lens, const float* __restrict__ stats,
                                                             const float* __restrict__ coef, int B, int T, int topk,
                                                             float* __restrict__ dyh) {
  const float c1 = coef[0], c2 = coef[1], c3 = coef[2];
  const long long total = (long long)B * T;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int b = (int)(e / T), i = (int)(e - (long long)b * T);
    const int len = lens ? min(lens[b], T) : T;
    float g = 0.f;
    if (i < len) {
      const float d = y[e] - yh[e];
      const float q = d * d;
      const float* s = stats + (size_t)b * 8;
      const unsigned qb = __float_as_uint(q), tb = __float_as_uint(s[3]);
      const float w = qb > tb ? 1.f : (qb == tb ? ((float)topk - s[4]) / fmaxf(s[5], 1.f) : 0.f);
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      g = -(c1 * sg + c2 * d + c3 * d * w);
    }
    dyh[e] = g;
  }
}

}  // namespace smt

using namespace smt;

extern "C" int smt_recon_loss_fwd(const float* y, const float* yh, const int* lens, int batch, int t, int topk, float* stats,
                                  smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(batch >= 0 && t >= 1 && topk >= 1 && topk <= t, "smt_recon_loss_fwd: need 1 <= topk <= t (got topk=%d, t=%d)", topk, t);
  if (batch == 0) return 0;
  SMT_CHECK_ARG(y && yh && stats, "smt_recon_loss_fwd: null pointer");
  recon_loss_fwd_kernel<<<batch, RL_NT, 0, stream>>>(y, yh, lens, t, topk, stats);
  SMT_CHECK_LAUNCH("recon_loss_fwd");
  return 0;
}

extern "C" int smt_recon_loss_bwd(const float* y, const float* yh, const int* lens, const float* stats, const float* coef,
                                  int batch, int t, int topk, float* dyh, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch == 0) return 0;
  SMT_CHECK_ARG(y && yh && stats && coef && dyh, "smt_recon_loss_bwd: null pointer");
  const long long total = (long long)batch * t;
  const unsigned grid = (unsigned)std::min<long long>(4096, (total + 255) / 256);
  recon_loss_bwd_kernel<<<grid, 256, 0, stream>>>(y, yh, lens, stats, coef, batch, t, topk, dyh);
  SMT_CHECK_LAUNCH("recon_loss_bwd");
  return 0;
}
