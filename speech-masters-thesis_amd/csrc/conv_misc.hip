// The HBM-bound members of the conv stack (no matrix-core work):
//   * gate_mix        sum_d tanh(t_d) * softmax_d(s_d)         (resnet.py:229-237) fwd + bwd
//   * conv_in         the encoder's first conv, C_in = 1        (conv.py:61, k=2s stride s) fwd + wgrad
//   * conv_out        the decoder's final 1x1 conv, C_out = 1   (encdec.py:61,82) fwd + bwd
// All are one pass over their operands with 16-byte accesses along the channel axis.
#include <algorithm>

#include "conv_common.h"

namespace smt {

constexpr int GM_MAX_DEPTH = 8;

// ------------------------------------------------------------------ gate_mix ----
// z: [rows, depth*2*w] (pitch ldz): branch d has t at [d*2w, d*2w+w), s at [d*2w+w, (d+1)*2w)
template <typename T>
__global__ __launch_bounds__(256) void gate_mix_fwd_kernel(const T* __restrict__ z, T* __restrict__ g, long long rows,
                                                           int w, int depth, int ldz, int ldg) {
  constexpr int EPV = Tr<T>::EPV;
  const int vpr = w / EPV;
  const long long total = rows * vpr;
  for (long long f = (long long)blockIdx.x * 256 + threadIdx.x; f < total; f += (long long)gridDim.x * 256) {
    const long long row = f / vpr;
    const int c = (int)(f % vpr) * EPV;
    float tv[GM_MAX_DEPTH][EPV], sv[GM_MAX_DEPTH][EPV];
#pragma unroll
    for (int d = 0; d < GM_MAX_DEPTH; ++d) {
      if (d < depth) {
        Vec<T, EPV> a = *reinterpret_cast<const Vec<T, EPV>*>(z + row * ldz + d * 2 * w + c);
        Vec<T, EPV> b = *reinterpret_cast<const Vec<T, EPV>*>(z + row * ldz + d * 2 * w + w + c);
#pragma unroll
        for (int e = 0; e < EPV; ++e) { tv[d][e] = (float)a.v[e]; sv[d][e] = (float)b.v[e]; }
      }
    }
    Vec<T, EPV> out;
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      float m = -INFINITY;
#pragma unroll
      for (int d = 0; d < GM_MAX_DEPTH; ++d) if (d < depth) m = fmaxf(m, sv[d][e]);
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int d = 0; d < GM_MAX_DEPTH; ++d) if (d < depth) {
        float ex = __expf(sv[d][e] - m);
        den += ex;
        num += ex * gate_tanh(tv[d][e]);
      }
      // bf16 results: one reciprocal (1 ulp) instead of an IEEE division (the fused forward, conv_k3gate, does the same: the two
      // stay bit-identical); the fp32 parity path keeps the division
      if constexpr (sizeof(T) == 2) out.v[e] = (T)(num * __builtin_amdgcn_rcpf(den));
      else out.v[e] = (T)(num / den);
    }
    *reinterpret_cast<Vec<T, EPV>*>(g + row * ldg + c) = out;
  }
}

// Backward: one lane per (row, branch d, 16-byte channel vector); the softmax statistics over the
// branches are exchanged with wave shuffles (lanes of one channel vector are `vpr` apart), so a lane
// holds two operand vectors instead of 2*depth of them: 4x the lanes in flight at a third of the
// registers -- this is an HBM-latency-bound kernel.  depth must be a power of two <= 8 and
// depth * (w / EPV) a divisor of 64 (w = 64: 32 lanes per row, two rows per wave).
// lane ^ 8 inside a 16-lane row = rotate the row by 8 (DPP row_ror:8); lane ^ 16 = ds_swizzle in bit mode (xor mask 0x10):
// neither needs an address register or an LDS access slot like ds_bpermute
__device__ __forceinline__ float gm_xor8(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
}
__device__ __forceinline__ float gm_xor16(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}

// VPR / DEPTH > 0 fix the geometry at compile time (the shuffle distances become constants: DPP / swizzle instead of
// ds_bpermute with computed addresses, loops unrolled); 0 = take them from the arguments.
template <typename T, int VPR = 0, int DEPTH = 0>
__global__ __launch_bounds__(256) void gate_mix_bwd_kernel(const T* __restrict__ z, const T* __restrict__ dg,
                                                           T* __restrict__ dz, long long rows, int w_, int depth_,
                                                           int ldz, int ldg, int lddz) {
  constexpr int EPV = Tr<T>::EPV;
  const int w = VPR > 0 ? VPR * EPV : w_;
  const int depth = DEPTH > 0 ? DEPTH : depth_;
  const int vpr = w / EPV;                 // channel vectors per row
  const int lpr = vpr * depth;             // lanes per row (a power of two dividing 64)
  const long long total = rows * lpr;
  const long long stride = (long long)gridDim.x * 256;
  const long long n_iter = (total + stride - 1) / stride;     // uniform trip count: shuffles need every lane
  // 256 % lpr == 0, so a thread keeps its (branch, channel vector) for the whole loop and its row advances by a constant:
  // no division in the loop (this kernel turned out VALU-bound, not HBM-bound: ~1,400 instructions per 80 bytes moved,
  // half of them two IEEE divisions per element and 64-bit index arithmetic)
  const int rem = threadIdx.x & (lpr - 1);
  const int d = rem / vpr, c = (rem % vpr) * EPV;
  const long long rows_per_iter = stride / lpr;
  long long row_it = ((long long)blockIdx.x * 256 + threadIdx.x) / lpr;
  for (long long it = 0; it < n_iter; ++it, row_it += rows_per_iter) {
    const bool live = row_it < rows;
    const long long row = live ? row_it : 0;
    float th[EPV], sx[EPV], go[EPV];
    {
      Vec<T, EPV> a, bb, gv;
#pragma unroll
      for (int e = 0; e < EPV; ++e) { a.v[e] = (T)0.f; bb.v[e] = (T)0.f; gv.v[e] = (T)0.f; }
      if (live) {
        a = *reinterpret_cast<const Vec<T, EPV>*>(z + row * ldz + d * 2 * w + c);
        bb = *reinterpret_cast<const Vec<T, EPV>*>(z + row * ldz + d * 2 * w + w + c);
        gv = *reinterpret_cast<const Vec<T, EPV>*>(dg + row * ldg + c);
      }
#pragma unroll
      for (int e = 0; e < EPV; ++e) { th[e] = gate_tanh((float)a.v[e]); sx[e] = (float)bb.v[e]; go[e] = (float)gv.v[e]; }
    }
    Vec<T, EPV> dt, ds;
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      float m = sx[e], ex, den, dot;
      if constexpr (VPR == 8 && DEPTH == 4) {      // branch d sits 8 d lanes away: xor 8 is a DPP row rotation, xor 16 a swizzle
        m = fmaxf(m, gm_xor8(m)); m = fmaxf(m, gm_xor16(m));
        ex = __expf(sx[e] - m);
        den = ex; dot = ex * th[e];
        den += gm_xor8(den); dot += gm_xor8(dot);
        den += gm_xor16(den); dot += gm_xor16(dot);
      } else {
        for (int o = vpr; o < lpr; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        ex = __expf(sx[e] - m);
        den = ex; dot = ex * th[e];
        for (int o = vpr; o < lpr; o <<= 1) { den += __shfl_xor(den, o, 64); dot += __shfl_xor(dot, o, 64); }
      }
      // bf16 results: one reciprocal (1 ulp) instead of two IEEE divisions; the fp32 parity path keeps the divisions
      float sm;
      if constexpr (sizeof(T) == 2) {
        const float inv = __builtin_amdgcn_rcpf(den);
        sm = ex * inv;
        dot *= inv;                                // sum_d softmax_d * tanh_d
      } else {
        sm = ex / den;
        dot /= den;
      }
      dt.v[e] = (T)(go[e] * sm * (1.f - th[e] * th[e]));
      ds.v[e] = (T)(go[e] * sm * (th[e] - dot));
    }
    if (live) {
      *reinterpret_cast<Vec<T, EPV>*>(dz + row * lddz + d * 2 * w + c) = dt;
      *reinterpret_cast<Vec<T, EPV>*>(dz + row * lddz + d * 2 * w + w + c) = ds;
    }
  }
}

// ------------------------------------------------------------------ conv_in -----
// y[b,t,co] = bias[co] + sum_j x[b, t*s + j - pad] * w[co][j];  x fp32 [B,Tin], rows >= lens[b] read 0
template <typename T>
__global__ __launch_bounds__(256) void conv_in_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const int* __restrict__ lens,
                                                          T* __restrict__ y, int B, int Tin, int Tout, int C, int taps,
                                                          int stride, int pad) {
  constexpr int EPV = Tr<T>::EPV;
  const int vpr = C / EPV;                       // power of two <= 256: a thread keeps its channel vector for all rows
  const int cv = threadIdx.x % vpr;
  float wr[EPV][8], br[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    br[e] = bias[cv * EPV + e];
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[e][j] = (j < taps) ? w[(cv * EPV + e) * taps + j] : 0.f;
  }
  // rows advance by a constant step: (b, t) are updated incrementally, no division in the loop
  const long long rows = (long long)B * Tout;
  const long long step = (long long)gridDim.x * (256 / vpr);
  const int step_b = (int)(step / Tout), step_t = (int)(step % Tout);
  long long bt = (long long)blockIdx.x * (256 / vpr) + threadIdx.x / vpr;
  int b = (int)(bt / Tout), t = (int)(bt - (long long)b * Tout);
  for (; bt < rows; bt += step) {
    const int len = lens ? min(lens[b], Tin) : Tin;
    float o[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) o[e] = br[e];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tin = t * stride + j - pad;
      const float xv = (j < taps && tin >= 0 && tin < len) ? x[(long long)b * Tin + tin] : 0.f;
#pragma unroll
      for (int e = 0; e < EPV; ++e) o[e] = fmaf(xv, wr[e][j], o[e]);
    }
    Vec<T, EPV> out;
#pragma unroll
    for (int e = 0; e < EPV; ++e) out.v[e] = (T)o[e];
    *reinterpret_cast<Vec<T, EPV>*>(y + bt * C + cv * EPV) = out;
    b += step_b; t += step_t;
    if (t >= Tout) { t -= Tout; ++b; }
  }
}

// partial[wg][co][9] (taps 0..7, bias at 8).  A thread owns EPV consecutive channels of every (256 / lanes-per-row)-th
// row of its workgroup's run of rows: 16-byte dy loads, the few x samples of a row are broadcast loads shared by
// the lanes of that row; (b, t) advance incrementally (no division in the loop).  Row groups are summed by
// wave shuffles, waves through LDS -- fixed order, bitwise reproducible.
template <typename T>
__global__ __launch_bounds__(256) void conv_in_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                            const int* __restrict__ lens, float* __restrict__ partial,
                                                            int B, int Tin, int Tout, int C, int taps, int stride,
                                                            int pad, long long rows_per_wg) {
  constexpr int EPV = Tr<T>::EPV;
  __shared__ float red[4][64][9];
  const int lpr = C / EPV;                       // lanes per row (power of two, <= 64)
  const int rpi = 256 / lpr;                     // rows per iteration
  const int lr = threadIdx.x % lpr, rq = threadIdx.x / lpr;
  const long long r0 = (long long)blockIdx.x * rows_per_wg;
  const long long r1 = min((long long)B * Tout, r0 + rows_per_wg);
  float acc[EPV][9];
#pragma unroll
  for (int e = 0; e < EPV; ++e)
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[e][j] = 0.f;
  long long bt = r0 + rq;
  int b = (int)(bt / Tout), t = (int)(bt - (long long)b * Tout);
  for (; bt < r1; bt += rpi) {
    const int len = lens ? min(lens[b], Tin) : Tin;
    const Vec<T, EPV> g = *reinterpret_cast<const Vec<T, EPV>*>(dy + bt * C + lr * EPV);
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tin = t * stride + j - pad;
      xv[j] = (j < taps && tin >= 0 && tin < len) ? x[(long long)b * Tin + tin] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const float gv = (float)g.v[e];
      acc[e][8] += gv;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[e][j] = fmaf(gv, xv[j], acc[e][j]);
    }
    t += rpi;
    while (t >= Tout) { t -= Tout; ++b; }
  }
  // sum the row groups of a wave (lanes lr, lr + lpr, ...), then the four waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < EPV; ++e)
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      float v = acc[e][j];
      for (int o = lpr; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      acc[e][j] = v;
    }
  if (lane < lpr) {
#pragma unroll
    for (int e = 0; e < EPV; ++e)
#pragma unroll
      for (int j = 0; j < 9; ++j) red[wave][(lane * EPV + e) & 63][j] = acc[e][j];
  }
  __syncthreads();
  // C <= 64 * (waves per row when lpr = 64) -- here C <= 64 * EPV / EPV: one thread per channel
  if (threadIdx.x < C) {
    const int nw = (lpr >= 64) ? 1 : 4;          // with 64 lanes per row a wave IS one row group per wave
    for (int j = 0; j < 9; ++j) {
      float s2 = 0.f;
      for (int w2 = 0; w2 < nw; ++w2) s2 += red[w2][threadIdx.x & 63][j];
      partial[((size_t)blockIdx.x * C + threadIdx.x) * 9 + j] = s2;
    }
  }
}

// One workgroup per output: thread i sums partials i, i + 256, ... in index order, then a fixed LDS tree.
__global__ __launch_bounds__(256) void conv_in_wgrad_reduce_kernel(const float* __restrict__ partial, int n_wg, int C,
                                                                   int taps, float* __restrict__ dw,
                                                                   float* __restrict__ db) {
  __shared__ float part[256];
  const int e = blockIdx.x;
  const int co = e / 9, j = e % 9;
  if (j >= taps && j != 8) return;
  float s = 0.f;
  for (int g = threadIdx.x; g < n_wg; g += 256) s += partial[((size_t)g * C + co) * 9 + j];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) { if (j == 8) db[co] = part[0]; else dw[co * taps + j] = part[0]; }
}

// ------------------------------------------------------------------ conv_out ----
// y[b,t] = bias + sum_c x[b,t,c]*m*w[c];  LPR lanes share a row
template <typename T>
__global__ __launch_bounds__(256) void conv_out_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const int* __restrict__ lens,
                                                           float* __restrict__ y, int B, int Tt, int C) {
  constexpr int EPV = Tr<T>::EPV;
  const int lpr = C / EPV;  // lanes per row (power of two, <= 64)
  const int rows_per_blk = 256 / lpr;
  const int lane_in_row = threadIdx.x % lpr;
  const long long total = (long long)B * Tt;
  for (long long row = (long long)blockIdx.x * rows_per_blk + threadIdx.x / lpr; row < total;
       row += (long long)gridDim.x * rows_per_blk) {
    const int t = (int)(row % Tt), b = (int)(row / Tt);
    const bool valid = !lens || t < lens[b];
    float s = 0.f;
    if (valid) {
      Vec<T, EPV> v = *reinterpret_cast<const Vec<T, EPV>*>(x + row * C + lane_in_row * EPV);
#pragma unroll
      for (int e = 0; e < EPV; ++e) s = fmaf((float)v.v[e], w[lane_in_row * EPV + e], s);
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane_in_row == 0) y[row] = s + bias[0];
  }
}

// dx[b,t,c] = dy[b,t]*m*w[c];  partial dw/db per workgroup
template <typename T>
__global__ __launch_bounds__(256) void conv_out_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                           const int* __restrict__ lens, const float* __restrict__ dy,
                                                           T* __restrict__ dx, float* __restrict__ partial, int B,
                                                           int Tt, int C, long long rows_per_wg) {
  constexpr int EPV = Tr<T>::EPV;
  __shared__ float red[256 * 9];
  const int lpr = C / EPV;
  const int rows_per_it = 256 / lpr;
  const int lane_in_row = threadIdx.x % lpr;
  const long long r0 = (long long)blockIdx.x * rows_per_wg;
  const long long r1 = min((long long)B * Tt, r0 + rows_per_wg);
  float acc[EPV], accb = 0.f;
#pragma unroll
  for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
  float wv[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) wv[e] = w[lane_in_row * EPV + e];
  for (long long row = r0 + threadIdx.x / lpr; row < r1; row += rows_per_it) {
    const int t = (int)(row % Tt), b = (int)(row / Tt);
    const bool valid = !lens || t < lens[b];
    const float g = dy[row];
    Vec<T, EPV> out;
    if (valid) {
      Vec<T, EPV> v = *reinterpret_cast<const Vec<T, EPV>*>(x + row * C + lane_in_row * EPV);
#pragma unroll
      for (int e = 0; e < EPV; ++e) { acc[e] = fmaf(g, (float)v.v[e], acc[e]); out.v[e] = (T)(g * wv[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < EPV; ++e) out.v[e] = (T)0.f;
    }
    if (lane_in_row == 0) accb += g;
    *reinterpret_cast<Vec<T, EPV>*>(dx + row * C + lane_in_row * EPV) = out;
  }
#pragma unroll
  for (int e = 0; e < EPV; ++e) red[threadIdx.x * 9 + e] = acc[e];
  red[threadIdx.x * 9 + 8] = accb;
  __syncthreads();
  if (threadIdx.x < lpr) {
    for (int e = 0; e < EPV; ++e) {
      float s = 0.f;
      for (int q = 0; q < rows_per_it; ++q) s += red[(q * lpr + threadIdx.x) * 9 + e];
      partial[(size_t)blockIdx.x * (C + 1) + threadIdx.x * EPV + e] = s;
    }
    if (threadIdx.x == 0) {
      float s = 0.f;
      for (int q = 0; q < rows_per_it; ++q) s += red[(q * lpr) * 9 + 8];
      partial[(size_t)blockIdx.x * (C + 1) + C] = s;
    }
  }
}

__global__ __launch_bounds__(256) void conv_out_reduce_kernel(const float* __restrict__ partial, int n_wg, int C,
                                                              float* __restrict__ dw, float* __restrict__ db) {
  // one workgroup per output: thread i sums partials i, i + 256, ... in index order, then a fixed LDS tree
  __shared__ float part[256];
  const int e = blockIdx.x;
  float s = 0.f;
  for (int g = threadIdx.x; g < n_wg; g += 256) s += partial[(size_t)g * (C + 1) + e];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) { if (e == C) db[0] = part[0]; else dw[e] = part[0]; }
}

static unsigned ew_grid(long long total) { return (unsigned)std::min<long long>(4096, (total + 255) / 256); }

}  // namespace smt

using namespace smt;

extern "C" int smt_gate_mix_fwd(const void* z, void* g, int dtype, int64_t rows, int width, int depth, int ld_z,
                                int ld_g, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int epv = dtype == SMT_BF16 ? 8 : 4;
  SMT_CHECK_ARG(z && g, "smt_gate_mix_fwd: null pointer");
  SMT_CHECK_ARG(depth >= 1 && depth <= GM_MAX_DEPTH && width % epv == 0 && ld_z % epv == 0 && ld_g % epv == 0,
                "smt_gate_mix_fwd: bad geometry");
  if (rows == 0) return 0;
  unsigned grid = ew_grid(rows * (width / epv));
  if (dtype == SMT_BF16)
    gate_mix_fwd_kernel<__bf16><<<grid, 256, 0, stream>>>((const __bf16*)z, (__bf16*)g, rows, width, depth, ld_z, ld_g);
  else
    gate_mix_fwd_kernel<float><<<grid, 256, 0, stream>>>((const float*)z, (float*)g, rows, width, depth, ld_z, ld_g);
  SMT_CHECK_LAUNCH("gate_mix_fwd");
  return 0;
}

extern "C" int smt_gate_mix_bwd(const void* z, const void* dg, void* dz, int dtype, int64_t rows, int width, int depth,
                                int ld_z, int ld_g, int ld_dz, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int epv = dtype == SMT_BF16 ? 8 : 4;
  SMT_CHECK_ARG(z && dg && dz, "smt_gate_mix_bwd: null pointer");
  SMT_CHECK_ARG(depth >= 1 && depth <= GM_MAX_DEPTH && width % epv == 0 && ld_z % epv == 0 && ld_g % epv == 0 &&
                    ld_dz % epv == 0, "smt_gate_mix_bwd: bad geometry");
  if (rows == 0) return 0;
  const int lpr = (width / epv) * depth;
  SMT_CHECK_ARG((depth & (depth - 1)) == 0 && lpr <= 64 && 64 % lpr == 0 && ((width / epv) & (width / epv - 1)) == 0,
                "smt_gate_mix_bwd: depth and width/%d must be powers of two with depth*width/%d <= 64", epv, epv);
  unsigned grid = ew_grid(rows * lpr);
  if (dtype == SMT_BF16 && width == 64 && depth == 4)      // the reference configuration: four branches of width 64
    gate_mix_bwd_kernel<__bf16, 8, 4><<<grid, 256, 0, stream>>>((const __bf16*)z, (const __bf16*)dg, (__bf16*)dz, rows, width,
                                                               depth, ld_z, ld_g, ld_dz);
  else if (dtype == SMT_BF16)
    gate_mix_bwd_kernel<__bf16><<<grid, 256, 0, stream>>>((const __bf16*)z, (const __bf16*)dg, (__bf16*)dz, rows, width,
                                                         depth, ld_z, ld_g, ld_dz);
  else
    gate_mix_bwd_kernel<float><<<grid, 256, 0, stream>>>((const float*)z, (const float*)dg, (float*)dz, rows, width,
                                                        depth, ld_z, ld_g, ld_dz);
  SMT_CHECK_LAUNCH("gate_mix_bwd");
  return 0;
}

extern "C" int smt_conv_in_fwd(const float* x, const float* weight, const float* bias, const int* lens, void* y,
                               int dtype, int batch, int t_in, int t_out, int c_out, int taps, int stride, int padding,
                               smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int epv = dtype == SMT_BF16 ? 8 : 4;
  SMT_CHECK_ARG(x && weight && bias && y, "smt_conv_in_fwd: null pointer");
  SMT_CHECK_ARG(c_out % epv == 0 && taps >= 1 && taps <= 8 && 256 % (c_out / epv) == 0, "smt_conv_in_fwd: bad geometry");
  if (batch == 0 || t_out == 0) return 0;
  unsigned grid = ew_grid((long long)batch * t_out * (c_out / epv));
  if (dtype == SMT_BF16)
    conv_in_fwd_kernel<__bf16><<<grid, 256, 0, stream>>>(x, weight, bias, lens, (__bf16*)y, batch, t_in, t_out, c_out,
                                                        taps, stride, padding);
  else
    conv_in_fwd_kernel<float><<<grid, 256, 0, stream>>>(x, weight, bias, lens, (float*)y, batch, t_in, t_out, c_out,
                                                       taps, stride, padding);
  SMT_CHECK_LAUNCH("conv_in_fwd");
  return 0;
}

static int conv_in_wgs(long long rows) { return (int)std::min<long long>(1024, std::max<long long>(1, rows / 512)); }

extern "C" size_t smt_conv_in_wgrad_workspace_bytes(int batch, int t_out, int c_out) {
  return (size_t)conv_in_wgs((long long)batch * t_out) * c_out * 9 * sizeof(float);
}

extern "C" int smt_conv_in_wgrad(const float* x, const void* dy, const int* lens, float* dweight, float* dbias,
                                 int dtype, int batch, int t_in, int t_out, int c_out, int taps, int stride,
                                 int padding, void* workspace, size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(x && dy && dweight && dbias && workspace, "smt_conv_in_wgrad: null pointer");
  {
    const int epv = dtype == SMT_BF16 ? 8 : 4;
    const int lpr = c_out / epv;
    SMT_CHECK_ARG(c_out % epv == 0 && c_out <= 64 && lpr >= 1 && (lpr & (lpr - 1)) == 0 && taps >= 1 && taps <= 8,
                  "smt_conv_in_wgrad: c_out must be a power-of-two multiple of the vector width, <= 64; taps <= 8");
  }
  const long long rows = (long long)batch * t_out;
  const int n_wg = conv_in_wgs(rows);
  SMT_CHECK_ARG(workspace_bytes >= smt_conv_in_wgrad_workspace_bytes(batch, t_out, c_out),
                "smt_conv_in_wgrad: workspace too small");
  const long long rpw = (rows + n_wg - 1) / n_wg;
  if (dtype == SMT_BF16)
    conv_in_wgrad_kernel<__bf16><<<n_wg, 256, 0, stream>>>(x, (const __bf16*)dy, lens, (float*)workspace, batch, t_in,
                                                          t_out, c_out, taps, stride, padding, rpw);
  else
    conv_in_wgrad_kernel<float><<<n_wg, 256, 0, stream>>>(x, (const float*)dy, lens, (float*)workspace, batch, t_in,
                                                         t_out, c_out, taps, stride, padding, rpw);
  SMT_CHECK_LAUNCH("conv_in_wgrad");
  conv_in_wgrad_reduce_kernel<<<c_out * 9, 256, 0, stream>>>((const float*)workspace, n_wg, c_out, taps,
                                                                           dweight, dbias);
  SMT_CHECK_LAUNCH("conv_in_wgrad_reduce");
  return 0;
}

extern "C" int smt_conv_out_fwd(const void* x, const float* weight, const float* bias, const int* lens, float* y,
                                int dtype, int batch, int t, int c_in, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int epv = dtype == SMT_BF16 ? 8 : 4;
  SMT_CHECK_ARG(x && weight && bias && y, "smt_conv_out_fwd: null pointer");
  const int lpr = c_in / epv;
  SMT_CHECK_ARG(c_in % epv == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0, "smt_conv_out_fwd: bad c_in=%d", c_in);
  if (batch == 0 || t == 0) return 0;
  unsigned grid = ew_grid((long long)batch * t * lpr);
  if (dtype == SMT_BF16)
    conv_out_fwd_kernel<__bf16><<<grid, 256, 0, stream>>>((const __bf16*)x, weight, bias, lens, y, batch, t, c_in);
  else
    conv_out_fwd_kernel<float><<<grid, 256, 0, stream>>>((const float*)x, weight, bias, lens, y, batch, t, c_in);
  SMT_CHECK_LAUNCH("conv_out_fwd");
  return 0;
}

extern "C" size_t smt_conv_out_bwd_workspace_bytes(int batch, int t, int c_in) {
  return (size_t)conv_in_wgs((long long)batch * t) * (c_in + 1) * sizeof(float);
}

extern "C" int smt_conv_out_bwd(const void* x, const float* weight, const int* lens, const float* dy, void* dx,
                                float* dweight, float* dbias, int dtype, int batch, int t, int c_in, void* workspace,
                                size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int epv = dtype == SMT_BF16 ? 8 : 4;
  SMT_CHECK_ARG(x && weight && dy && dx && dweight && dbias && workspace, "smt_conv_out_bwd: null pointer");
  const int lpr = c_in / epv;
  SMT_CHECK_ARG(c_in % epv == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0, "smt_conv_out_bwd: bad c_in=%d", c_in);
  const long long rows = (long long)batch * t;
  const int n_wg = conv_in_wgs(rows);
  SMT_CHECK_ARG(workspace_bytes >= smt_conv_out_bwd_workspace_bytes(batch, t, c_in), "smt_conv_out_bwd: workspace too small");
  const long long rpw = (rows + n_wg - 1) / n_wg;
  if (dtype == SMT_BF16)
    conv_out_bwd_kernel<__bf16><<<n_wg, 256, 0, stream>>>((const __bf16*)x, weight, lens, dy, (__bf16*)dx,
                                                         (float*)workspace, batch, t, c_in, rpw);
  else
    conv_out_bwd_kernel<float><<<n_wg, 256, 0, stream>>>((const float*)x, weight, lens, dy, (float*)dx,
                                                        (float*)workspace, batch, t, c_in, rpw);
  SMT_CHECK_LAUNCH("conv_out_bwd");
  conv_out_reduce_kernel<<<c_in + 1, 256, 0, stream>>>((const float*)workspace, n_wg, c_in, dweight, dbias);
  SMT_CHECK_LAUNCH("conv_out_reduce");
  return 0;
}
