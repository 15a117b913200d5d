// Vector-quantiser kernels for gfx950: exact nearest-code search, dequantise,
// commit/fit terms, straight-through backward, codebook EMA statistics and update.
//
// Replaces BottleneckBlock.quantize/dequantize/update_k of the reference
// (models/vqvae/bottleneck.py:60-90, 126-145, 171-201).
//
// Index semantics (oracle/vqvae_oracle.py: vq_argmin_exact): idx = the exact
// argmin_j ||x - k_j||^2 of the fp32 inputs, lowest j on ties.  Implementation:
//   1. vq_score   filter scores on the matrix cores (bf16-pair split of the fp32 operands,
//                 3 x v_mfma_f32_32x32x16_bf16 per k-step), per row best + runner-up per 128-code slice;
//   2. vq_finalize merges the slices; a row whose best/runner-up gap is inside the
//                 rigorous fp32 round-off bound is queued, every other row is final;
//   3. vq_rescore re-scores queued rows over ALL codes in fp64 (index order, no
//                 contraction) -- the oracle's arithmetic;
//   4. vq_reduce  fixed-order sums (deterministic commit / fit).
// The [N, K] distance matrix is never materialised.
#include "smt_common.h"

namespace smt {

constexpr int VQ_ROWS_PER_WG = 128;   // 4 waves x 32 rows
constexpr int VQ_SLICE = 128;         // codes per workgroup (4 chunks of 32)
constexpr int VQ_CHUNK = 32;

// ---------------------------------------------------------------- prep ------
// Distances are translation invariant, so the fp32 scoring pass runs on data centred at the codebook
// mean mu: trained encoders emit rows with a large common offset, and the round-off bound that
// decides which rows need fp64 re-scoring scales with the NORMS of the operands, not their spread.
// mu[i] = mean_j k[j][i] (one thread per dimension, index order).
__global__ __launch_bounds__(1024) void vq_mean_kernel(const float* __restrict__ cb, int K, int D,
                                                       float* __restrict__ mu) {
  __shared__ float part[1024];
  const int parts = 1024 / D;                     // D in {32, 64, 128}
  const int i = threadIdx.x % D, pt = threadIdx.x / D;
  float s = 0.f;
  for (int j = pt; j < K; j += parts) s += cb[(size_t)j * D + i];   // fixed order per part
  part[threadIdx.x] = s;
  __syncthreads();
  if (pt == 0) {
    float t = 0.f;
    for (int q = 0; q < parts; ++q) t += part[q * D + i];            // parts combined in index order
    mu[i] = t / (float)K;
  }
}
// kc[j] = k[j] - mu, stored as the bf16 pair (kh, kl);  khalf[j] = 0.5 * |kc[j]|^2 (fp32, index order per lane then wave tree);
// kmax2 = max_j |kc[j]|^2.  One wave per code.
__global__ __launch_bounds__(256) void vq_prep_kernel(const float* __restrict__ cb, const float* __restrict__ mu,
                                                      int K, int D, __bf16* __restrict__ kh, __bf16* __restrict__ kl,
                                                      float* __restrict__ khalf, unsigned* __restrict__ kmax2_bits) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (wave >= K) return;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) {
    float v = cb[(size_t)wave * D + i] - mu[i];
    const __bf16 hi = (__bf16)v;                       // v = hi + lo + eps, |eps| <= 2^-18 |v|
    kh[(size_t)wave * D + i] = hi;
    kl[(size_t)wave * D + i] = (__bf16)(v - (float)hi);
    s = fmaf(v, v, s);
  }
  s = wave_sum(s);
  if (lane == 0) {
    khalf[wave] = 0.5f * s;
    atomicMax(kmax2_bits, __float_as_uint(s));  // s >= 0: uint order == float order
  }
}

// ---------------------------------------------------------------- score -----
// Workgroup = 4 waves = 128 rows x one 128-code slice.  The codebook is the MFMA A operand (code on the
// row index i), x the B operand (row on the column index j = lane & 31), so every lane owns ONE x row
// and sees 16 codes per chunk in its accumulator registers: the running best / runner-up is pure in-lane
// work.  Arithmetic: each fp32 operand is split into a bf16 pair (v = hi + lo + eps, |eps| <= 2^-18 |v|)
// and x.k is evaluated as kh.xh + kh.xl + kl.xh on v_mfma_f32_32x32x16_bf16 with fp32 accumulation --
// 3 bf16 MFMAs (16x the fp32-MFMA rate each) replace 8 fp32 MFMAs.  The score is only a FILTER: its
// error bound (vq_finalize) decides which rows are re-scored exactly, so the index semantics stay exact.
typedef __bf16 vq_bf16x8 __attribute__((ext_vector_type(8)));

template <int D>
__global__ __launch_bounds__(256) void vq_score_kernel(const float* __restrict__ x, const __bf16* __restrict__ kh,
                                                       const __bf16* __restrict__ kl, const float* __restrict__ mu,
                                                       const float* __restrict__ khalf, long long N, int K, int S,
                                                       float* __restrict__ p_best, int* __restrict__ p_idx,
                                                       float* __restrict__ p_second) {
  constexpr int NS = D / 16;                       // k-steps per chunk
  constexpr int LDW = D + 8;                       // +16 B pad: conflict-free ds_read_b128
  constexpr int V_PER_TILE = VQ_CHUNK * D / 8;     // 16-byte vectors in one staged 32-code tile (hi or lo)
  constexpr int STAGE = (2 * V_PER_TILE + 255) / 256;
  __shared__ __attribute__((aligned(16))) __bf16 lds[2][2][VQ_CHUNK * LDW];   // [buffer][hi|lo]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int tile = blockIdx.x / S, slice = blockIdx.x % S;
  const long long row = (long long)tile * VQ_ROWS_PER_WG + wave * 32 + j;
  const int code0 = slice * VQ_SLICE;
  const int nchunk = min(VQ_SLICE / VQ_CHUNK, (K - code0 + VQ_CHUNK - 1) / VQ_CHUNK);

  // this lane's share of its x row, centred and split: dims 16 s + 8 h .. + 7 for every k-step s
  vq_bf16x8 xh[NS], xl[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (row < N) {
      const f32x4* src = reinterpret_cast<const f32x4*>(x + row * D + 16 * s + 8 * h);
      const f32x4* msrc = reinterpret_cast<const f32x4*>(mu + 16 * s + 8 * h);
      v0 = src[0] - msrc[0];
      v1 = src[1] - msrc[1];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = e < 4 ? v0[e & 3] : v1[e & 3];
      const __bf16 hi = (__bf16)v;
      xh[s][e] = hi;
      xl[s][e] = (__bf16)(v - (float)hi);
    }
  }

  f32x4 stage[STAGE];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int r = 0; r < STAGE; ++r) {
      const int f = threadIdx.x + 256 * r;                 // [hi tile vectors | lo tile vectors]
      const int which = f / V_PER_TILE, g = f % V_PER_TILE;
      const int code = g / (D / 8), c8 = g % (D / 8);
      const int gcode = code0 + c * VQ_CHUNK + code;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (f < 2 * V_PER_TILE && gcode < K)
        v = *reinterpret_cast<const f32x4*>((which ? kl : kh) + (size_t)gcode * D + 8 * c8);
      stage[r] = v;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int r = 0; r < STAGE; ++r) {
      const int f = threadIdx.x + 256 * r;
      const int which = f / V_PER_TILE, g = f % V_PER_TILE;
      const int code = g / (D / 8), c8 = g % (D / 8);
      if (f < 2 * V_PER_TILE) *reinterpret_cast<f32x4*>(&lds[buf][which][code * LDW + 8 * c8]) = stage[r];
    }
  };

  float best = -INFINITY, second = -INFINITY;
  int bidx = 0x7fffffff;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) load_chunk(c + 1);
    const int cbase = code0 + c * VQ_CHUNK;
    // accumulator starts at -0.5*|k_i|^2 so that acc = x.k_i - 0.5|k_i|^2 (argmax == argmin distance)
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int code = cbase + 8 * g + 4 * h + e;
        acc[4 * g + e] = (code < K) ? -khalf[code] : 0.f;
      }
    }
    const __bf16* ah = &lds[buf][0][j * LDW + 8 * h];
    const __bf16* al = &lds[buf][1][j * LDW + 8 * h];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const vq_bf16x8 fh = *reinterpret_cast<const vq_bf16x8*>(ah + 16 * s);
      const vq_bf16x8 fl = *reinterpret_cast<const vq_bf16x8*>(al + 16 * s);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, xh[s], acc, 0, 0, 0);   // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, xl[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, xh[s], acc, 0, 0, 0);
    }
    // in-lane running best / runner-up; codes visited in increasing order, strict '>' keeps
    // the lowest index among equal scores
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int code = cbase + 8 * (r >> 2) + 4 * h + (r & 3);
      float v = (code < K) ? acc[r] : -INFINITY;
      if (v > best) {
        second = best; best = v; bidx = code;
      } else if (v > second) {
        second = v;
      }
    }
    if (c + 1 < nchunk) store_chunk(buf ^ 1);
    __syncthreads();
  }
  // merge the two lane halves of a row (same row, disjoint codes)
  float ob = __shfl_xor(best, 32, 64), os = __shfl_xor(second, 32, 64);
  int oi = __shfl_xor(bidx, 32, 64);
  if (ob > best || (ob == best && oi < bidx)) {
    second = fmaxf(best, os); best = ob; bidx = oi;
  } else {
    second = fmaxf(second, ob);
  }
  if (h == 0 && row < N) {
    size_t o = (size_t)row * S + slice;
    p_best[o] = best; p_idx[o] = bidx; p_second[o] = second;
  }
}

// ---------------------------------------------------------------- finalize --
// One wave per row: merge slices, test the gap against the round-off bound, and
// for final rows write idx / min_dist / x_d.  Ambiguous rows go to the queue.
__global__ __launch_bounds__(256) void vq_finalize_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                          const float* __restrict__ mu,
                                                          const float* __restrict__ row_mask,
                                                          const unsigned* __restrict__ kmax2_bits,
                                                          const float* __restrict__ p_best, const int* __restrict__ p_idx,
                                                          const float* __restrict__ p_second, long long N, int D, int S,
                                                          long long* __restrict__ idx, float* __restrict__ min_dist,
                                                          float* __restrict__ x_d, int* __restrict__ q_count,
                                                          int* __restrict__ q_rows) {
  const int lane = threadIdx.x & 63;
  const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= N) return;
  const float* xr = x + row * D;
  float xv[4];
  float xx = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    int i = lane + 64 * u;
    xv[u] = (i < D) ? xr[i] : 0.f;
    const float xc = (i < D) ? xv[u] - mu[i] : 0.f;   // centred, as the scoring pass saw it
    xx = fmaf(xc, xc, xx);
  }
  xx = wave_sum(xx);
  float best = -INFINITY, second = -INFINITY;
  int bidx = 0;
  for (int s = 0; s < S; ++s) {  // slice order == code order; strict '>' keeps the lowest index
    size_t o = (size_t)row * S + s;
    float b = p_best[o], se = p_second[o];
    if (b > best) {
      second = fmaxf(best, se); best = b; bidx = p_idx[o];
    } else {
      second = fmaxf(second, b);
    }
  }
  // Error of the filter score against the exact acc = x~.k~ - |k~|^2/2 on the centred operands
  // (norms below are the centred ones, Cauchy-Schwarz turns sums of products into norm products):
  //   bf16-pair split, three of the four partial products kept:  <= 3.01 * 2^-18 |x~| |k~|
  //   fp32 accumulation of 3D exact products + the initial term:  <= 1.05 (3D+2) 2^-24 (|x~||k~| + |k~|^2/2)
  //   fp32 rounding of khalf and of the centring x - mu, k - mu:  <= 2^-24 ((D+2)|k~|^2/2 + (|x~| + |k~|)^2)
  // with |k~| <= |k~|max; a factor 1.25 of slack covers the MFMA's internal summation order.
  const float kmax2 = __uint_as_float(*kmax2_bits);
  const float xk = sqrtf(xx * kmax2);
  const float u24 = 5.9604645e-8f;
  const float err = 1.25f * (3.01f * 64.f * u24 * xk + 1.05f * (float)(3 * D + 2) * u24 * (xk + 0.5f * kmax2) +
                             u24 * (0.5f * (float)(D + 2) * kmax2 + xx + 2.f * xk + kmax2));
  const bool ambiguous = !((best - second) > 2.0f * err);  // also catches NaN / inf
  if (ambiguous) {
    if (lane == 0) q_rows[atomicAdd(q_count, 1)] = (int)row;
    return;
  }
  const float m = row_mask ? row_mask[row] : 1.f;
  const float* kr = cb + (size_t)bidx * D;
  float dsum = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    int i = lane + 64 * u;
    if (i < D) {
      float kv = kr[i];
      float df = xv[u] - kv;
      dsum = fmaf(df, df, dsum);
      if (x_d) x_d[row * D + i] = kv * m;
    }
  }
  dsum = wave_sum(dsum);
  if (lane == 0) {
    idx[row] = bidx;
    min_dist[row] = dsum;
  }
}

// ---------------------------------------------------------------- rescore ---
// Exact fp64 re-scoring of queued rows over ALL codes: d_j = sum_i (x_i - k_ji)^2 accumulated in index
// order without fma contraction (== the numpy float64 loop of the oracle).  A workgroup takes up to RB queued
// rows at a time; thread t owns code (block*256 + t) and scores it against all of them, so one pass of the
// codebook through LDS ([256 codes][32 dims], pitch 33 floats: coalesced 16-byte global reads, conflict-free
// per-code walks) serves RB rows -- with one row per workgroup the 512 KiB codebook was re-staged per row,
// which made this kernel the largest part of smt_vq_forward on an untrained encoder (many near-tie rows).
constexpr int VQ_RB = 8;
template <int NR>
__device__ __forceinline__ void vq_rescore_accum(const float* kr, const double (*xs)[256], int d0, int dn, double* acc) {
  for (int i = 0; i < dn; ++i) {
    const double kv = (double)kr[i];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      const double df = __dsub_rn(xs[rr][d0 + i], kv);
      acc[rr] = __dadd_rn(acc[rr], __dmul_rn(df, df));
    }
  }
}
__global__ __launch_bounds__(256) void vq_rescore_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                         const float* __restrict__ row_mask, long long N, int K, int D,
                                                         const int* __restrict__ q_count, const int* __restrict__ q_rows,
                                                         long long* __restrict__ idx, float* __restrict__ min_dist,
                                                         float* __restrict__ x_d) {
  constexpr int HB = 32, PITCH = 33, RB = VQ_RB;     // 33.8 KiB tile + 16 KiB of rows: three workgroups per CU
  __shared__ double xs[RB][256];
  __shared__ float tile[256 * PITCH];
  __shared__ double red_d[256];
  __shared__ int red_i[256];
  const int n_q = *q_count;
  const int halves = (D + HB - 1) / HB;
  // rows per workgroup and pass: as few as keeps every workgroup busy, at most RB
  const int rpw = max(1, min(RB, (n_q + (int)gridDim.x - 1) / (int)gridDim.x));
  for (int q0 = blockIdx.x * rpw; q0 < n_q; q0 += gridDim.x * rpw) {
    const int nr = min(rpw, n_q - q0);
    __syncthreads();
    for (int f = threadIdx.x; f < nr * D; f += 256) {
      const int rr = f / D, i = f - rr * D;
      xs[rr][i] = (double)x[(long long)q_rows[q0 + rr] * D + i];
    }
    double bd[RB];
    int bi[RB];
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) { bd[rr] = INFINITY; bi[rr] = 0x7fffffff; }
    for (int c0 = 0; c0 < K; c0 += 256) {
      double acc[RB];
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) acc[rr] = 0.0;
      for (int hf = 0; hf < halves; ++hf) {
        const int d0 = hf * HB, dn = min(HB, D - d0);          // dims [d0, d0 + dn), dn % 4 == 0
        __syncthreads();
        const int v_per_code = dn / 4;
        for (int f = threadIdx.x; f < 256 * v_per_code; f += 256) {
          const int code = f / v_per_code, c4 = f % v_per_code;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (c0 + code < K) v = *reinterpret_cast<const f32x4*>(cb + (size_t)(c0 + code) * D + d0 + 4 * c4);
          float* dst = &tile[code * PITCH + 4 * c4];
          dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
        __syncthreads();
        const float* kr = &tile[threadIdx.x * PITCH];
        switch (nr) {                                            // workgroup-uniform: straight-line code per row count
          case 1: vq_rescore_accum<1>(kr, xs, d0, dn, acc); break;
          case 2: vq_rescore_accum<2>(kr, xs, d0, dn, acc); break;
          case 3: vq_rescore_accum<3>(kr, xs, d0, dn, acc); break;
          case 4: vq_rescore_accum<4>(kr, xs, d0, dn, acc); break;
          case 5: vq_rescore_accum<5>(kr, xs, d0, dn, acc); break;
          case 6: vq_rescore_accum<6>(kr, xs, d0, dn, acc); break;
          case 7: vq_rescore_accum<7>(kr, xs, d0, dn, acc); break;
          default: vq_rescore_accum<8>(kr, xs, d0, dn, acc); break;
        }
      }
      const int code = c0 + threadIdx.x;
#pragma unroll
      for (int rr = 0; rr < RB; ++rr)
        if (code < K && acc[rr] < bd[rr]) { bd[rr] = acc[rr]; bi[rr] = code; }   // increasing code order: lowest index on ties
    }
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) {
      if (rr >= nr) break;                                       // workgroup-uniform
      __syncthreads();
      red_d[threadIdx.x] = bd[rr];
      red_i[threadIdx.x] = bi[rr];
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
          double od = red_d[threadIdx.x + o];
          int oi = red_i[threadIdx.x + o];
          if (od < red_d[threadIdx.x] || (od == red_d[threadIdx.x] && oi < red_i[threadIdx.x])) {
            red_d[threadIdx.x] = od; red_i[threadIdx.x] = oi;
          }
        }
        __syncthreads();
      }
      const long long row = q_rows[q0 + rr];
      const int wi = red_i[0];
      const float m = row_mask ? row_mask[row] : 1.f;
      if (x_d && threadIdx.x < D) x_d[row * D + threadIdx.x] = cb[(size_t)wi * D + threadIdx.x] * m;
      if (threadIdx.x == 0) {
        idx[row] = wi;
        min_dist[row] = (float)red_d[0];   // the exact distance, rounded once
      }
    }
  }
}

// ---------------------------------------------------------------- reduce ----
// Single workgroup, fixed order: sums[0] = sum_all min_dist, sums[1] = sum_masked,
// sums[2] = sum mask, sums[3] = queued rows.
__global__ __launch_bounds__(1024) void vq_reduce_kernel(const float* __restrict__ min_dist,
                                                         const float* __restrict__ row_mask, long long N,
                                                         const int* __restrict__ q_count, float* __restrict__ sums) {
  __shared__ double sh[3][16];
  double a = 0.0, b = 0.0, c = 0.0;
  for (long long r = threadIdx.x; r < N; r += 1024) {
    float d = min_dist[r];
    float m = row_mask ? row_mask[r] : 1.f;
    a += d; b += (m != 0.f) ? d : 0.f; c += m;
  }
  a = wave_sum_d(a); b = wave_sum_d(b); c = wave_sum_d(c);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[0][wave] = a; sh[1][wave] = b; sh[2][wave] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0, tb = 0, tc = 0;
    for (int w = 0; w < 16; ++w) { ta += sh[0][w]; tb += sh[1][w]; tc += sh[2][w]; }
    sums[0] = (float)ta; sums[1] = (float)tb; sums[2] = (float)tc; sums[3] = (float)(*q_count);
  }
}

// ---------------------------------------------------------------- backward --
__global__ __launch_bounds__(256) void vq_backward_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                          const long long* __restrict__ idx,
                                                          const float* __restrict__ row_mask, const float* __restrict__ dy,
                                                          const float* __restrict__ g_commit, const float* __restrict__ sums,
                                                          long long N, int D, float* __restrict__ dx) {
  const long long total4 = N * D / 4;
  const float gc = g_commit ? (*g_commit) * 2.0f / (sums[2] * (float)D) : 0.f;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total4;
       e += (long long)gridDim.x * blockDim.x) {
    const long long row = (e * 4) / D;
    const int col = (int)((e * 4) % D);
    const float m = row_mask ? row_mask[row] : 1.f;
    f32x4 xv = *reinterpret_cast<const f32x4*>(x + e * 4);
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    if (dy) {
      f32x4 g = *reinterpret_cast<const f32x4*>(dy + e * 4);
      out = g * m;
    }
    if (g_commit && m != 0.f) {
      f32x4 kv = *reinterpret_cast<const f32x4*>(cb + (size_t)idx[row] * D + col);
      out += (xv - kv) * gc;
    }
    *reinterpret_cast<f32x4*>(dx + e * 4) = out;
  }
}

// ---------------------------------------------------------------- EMA -------
// One wave per row; 256 contiguous bytes per atomic wave-instruction (the shape the
// memory-side f32 atomic unit runs at full rate).
__global__ __launch_bounds__(256) void vq_ema_accumulate_kernel(const float* __restrict__ x,
                                                                const long long* __restrict__ idx,
                                                                const float* __restrict__ row_mask, long long N, int K,
                                                                int D, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long row = wave0; row < N; row += nwaves) {
    if (row_mask && row_mask[row] == 0.f) continue;
    const long long code = idx[row];
    float* dst = stats + (size_t)code * D;
    for (int i = lane; i < D; i += 64) atomicAdd(dst + i, x[row * D + i]);
    if (lane == 0) atomicAdd(stats + (size_t)K * D + code, 1.0f);
  }
}

// Single workgroup (K*D is ~1e5): EMA mix, revival, metrics -- all reductions in fixed order.
__global__ __launch_bounds__(1024) void vq_ema_apply_kernel(float* __restrict__ cb, float* __restrict__ k_sum,
                                                            float* __restrict__ k_elem, const float* __restrict__ stats,
                                                            const float* __restrict__ k_rand, float mu, float threshold,
                                                            int K, int D, float* __restrict__ metrics) {
  __shared__ double sh[16];
  __shared__ double bc;
  auto block_sum = [&](double v) -> double {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
      bc = t;
    }
    __syncthreads();
    return bc;
  };
  const float* cnt = stats + (size_t)K * D;
  double tot = 0.0;
  for (int j = threadIdx.x; j < K; j += blockDim.x) tot += cnt[j];
  const float total = (float)block_sum(tot);

  double ent = 0.0, used = 0.0, usage_n = 0.0;
  for (int j = threadIdx.x; j < K; j += blockDim.x) {
    float c = cnt[j];
    float prob = c / total;
    ent += -(double)(prob * logf(fmaxf(prob, 1e-5f)));
    used += (c >= threshold) ? 1.0 : 0.0;
    float ne = mu * k_elem[j] + (1.f - mu) * c;
    usage_n += (ne >= threshold) ? 1.0 : 0.0;
  }
  ent = block_sum(ent);
  used = block_sum(used);
  usage_n = block_sum(usage_n);

  double dk2 = 0.0;
  for (int e = threadIdx.x; e < K * D; e += blockDim.x) {
    int j = e / D;
    float ne = mu * k_elem[j] + (1.f - mu) * cnt[j];   // k_elem is rewritten only after this loop
    float ns = mu * k_sum[e] + (1.f - mu) * stats[e];
    float usage = (ne >= threshold) ? 1.f : 0.f;
    float nk = usage * (ns / ne) + (1.f - usage) * k_rand[e];
    float d = nk - cb[e];
    dk2 += (double)d * d;
    k_sum[e] = ns;
    cb[e] = nk;
  }
  dk2 = block_sum(dk2);
  __syncthreads();
  for (int j = threadIdx.x; j < K; j += blockDim.x) k_elem[j] = mu * k_elem[j] + (1.f - mu) * cnt[j];
  if (threadIdx.x == 0) {
    metrics[0] = (float)ent;
    metrics[1] = (float)used;
    metrics[2] = (float)usage_n;
    metrics[3] = (float)(sqrt(dk2) / sqrt((double)K * D));
  }
}

struct VqWorkspace {
  float* khalf; unsigned* kmax2; int* q_count; float* p_best; int* p_idx; float* p_second; int* q_rows;
  float* mu; __bf16* kh; __bf16* kl;
};

static size_t vq_layout(long long N, int K, int D, int S, void* base, VqWorkspace* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return (char*)base + o; };
  char* p;
  p = take((size_t)K * 4); if (w) w->khalf = (float*)p;
  p = take(256);           if (w) { w->kmax2 = (unsigned*)p; w->q_count = (int*)(p + 128); }
  p = take((size_t)N * S * 4); if (w) w->p_best = (float*)p;
  p = take((size_t)N * S * 4); if (w) w->p_idx = (int*)p;
  p = take((size_t)N * S * 4); if (w) w->p_second = (float*)p;
  p = take((size_t)N * 4);     if (w) w->q_rows = (int*)p;
  p = take((size_t)D * 4);     if (w) w->mu = (float*)p;
  p = take((size_t)K * D * 2); if (w) w->kh = (__bf16*)p;
  p = take((size_t)K * D * 2); if (w) w->kl = (__bf16*)p;
  return off;
}

}  // namespace smt

using namespace smt;

extern "C" size_t smt_vq_forward_workspace_bytes(int64_t n_rows, int k_bins, int dim) {
  int S = (k_bins + VQ_SLICE - 1) / VQ_SLICE;
  return vq_layout(n_rows, k_bins, dim, S, nullptr, nullptr);
}

extern "C" int smt_vq_forward(const float* x, const float* codebook, const float* row_mask, int64_t n_rows, int k_bins,
                              int dim, int64_t* idx, float* min_dist, float* x_d, float* sums, void* workspace,
                              size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(dim == 32 || dim == 64 || dim == 128, "smt_vq_forward: dim must be 32, 64 or 128 (got %d)", dim);
  SMT_CHECK_ARG(k_bins >= 1 && n_rows >= 0, "smt_vq_forward: bad sizes n_rows=%lld k_bins=%d", (long long)n_rows, k_bins);
  SMT_CHECK_ARG(n_rows < (1ll << 31), "smt_vq_forward: n_rows must be < 2^31");
  SMT_CHECK_ARG(codebook && sums && workspace, "smt_vq_forward: null pointer");
  SMT_CHECK_ARG(n_rows == 0 || (x && idx && min_dist), "smt_vq_forward: null pointer");
  const int S = (k_bins + VQ_SLICE - 1) / VQ_SLICE;
  SMT_CHECK_ARG(workspace_bytes >= vq_layout(n_rows, k_bins, dim, S, nullptr, nullptr), "smt_vq_forward: workspace too small");
  VqWorkspace w;
  vq_layout(n_rows, k_bins, dim, S, workspace, &w);
  (void)hipMemsetAsync(w.kmax2, 0, 256, stream);
  if (n_rows == 0) {
    (void)hipMemsetAsync(sums, 0, 16, stream);
    return 0;
  }
  vq_mean_kernel<<<1, 1024, 0, stream>>>(codebook, k_bins, dim, w.mu);
  SMT_CHECK_LAUNCH("vq_mean");
  vq_prep_kernel<<<(k_bins * 64 + 255) / 256, 256, 0, stream>>>(codebook, w.mu, k_bins, dim, w.kh, w.kl, w.khalf, w.kmax2);
  SMT_CHECK_LAUNCH("vq_prep");
  const long long tiles = (n_rows + VQ_ROWS_PER_WG - 1) / VQ_ROWS_PER_WG;
  const unsigned grid = (unsigned)(tiles * S);
  if (dim == 128)
    vq_score_kernel<128><<<grid, 256, 0, stream>>>(x, w.kh, w.kl, w.mu, w.khalf, n_rows, k_bins, S, w.p_best, w.p_idx, w.p_second);
  else if (dim == 64)
    vq_score_kernel<64><<<grid, 256, 0, stream>>>(x, w.kh, w.kl, w.mu, w.khalf, n_rows, k_bins, S, w.p_best, w.p_idx, w.p_second);
  else
    vq_score_kernel<32><<<grid, 256, 0, stream>>>(x, w.kh, w.kl, w.mu, w.khalf, n_rows, k_bins, S, w.p_best, w.p_idx, w.p_second);
  SMT_CHECK_LAUNCH("vq_score");
  vq_finalize_kernel<<<(unsigned)((n_rows * 64 + 255) / 256), 256, 0, stream>>>(
      x, codebook, w.mu, row_mask, w.kmax2, w.p_best, w.p_idx, w.p_second, n_rows, dim, S, (long long*)idx, min_dist, x_d,
      w.q_count, w.q_rows);
  SMT_CHECK_LAUNCH("vq_finalize");
  vq_rescore_kernel<<<512, 256, 0, stream>>>(x, codebook, row_mask, n_rows, k_bins, dim, w.q_count, w.q_rows,
                                             (long long*)idx, min_dist, x_d);
  SMT_CHECK_LAUNCH("vq_rescore");
  vq_reduce_kernel<<<1, 1024, 0, stream>>>(min_dist, row_mask, n_rows, w.q_count, sums);
  SMT_CHECK_LAUNCH("vq_reduce");
  return 0;
}

extern "C" int smt_vq_backward(const float* x, const float* codebook, const int64_t* idx, const float* row_mask,
                               const float* dy, const float* g_commit, const float* sums, int64_t n_rows, int dim,
                               float* dx, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(dim % 4 == 0, "smt_vq_backward: dim must be a multiple of 4");
  if (n_rows == 0) return 0;
  SMT_CHECK_ARG(x && codebook && idx && sums && dx, "smt_vq_backward: null pointer");
  long long total4 = n_rows * dim / 4;
  unsigned grid = (unsigned)min((long long)2048, (total4 + 255) / 256);
  vq_backward_kernel<<<grid, 256, 0, stream>>>(x, codebook, (const long long*)idx, row_mask, dy, g_commit, sums, n_rows,
                                               dim, dx);
  SMT_CHECK_LAUNCH("vq_backward");
  return 0;
}

extern "C" int smt_vq_ema_accumulate(const float* x, const int64_t* idx, const float* row_mask, int64_t n_rows,
                                     int k_bins, int dim, float* stats, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(stats && (n_rows == 0 || (x && idx)), "smt_vq_ema_accumulate: null pointer");
  (void)hipMemsetAsync(stats, 0, ((size_t)k_bins * dim + k_bins) * sizeof(float), stream);
  if (n_rows == 0) return 0;
  unsigned grid = (unsigned)min((long long)4096, (n_rows * 64 + 255) / 256);
  vq_ema_accumulate_kernel<<<grid, 256, 0, stream>>>(x, (const long long*)idx, row_mask, n_rows, k_bins, dim, stats);
  SMT_CHECK_LAUNCH("vq_ema_accumulate");
  return 0;
}

extern "C" int smt_vq_ema_apply(float* codebook, float* k_sum, float* k_elem, const float* stats, const float* k_rand,
                                float mu, float threshold, int k_bins, int dim, float* metrics, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(codebook && k_sum && k_elem && stats && k_rand && metrics, "smt_vq_ema_apply: null pointer");
  vq_ema_apply_kernel<<<1, 1024, 0, stream>>>(codebook, k_sum, k_elem, stats, k_rand, mu, threshold, k_bins, dim, metrics);
  SMT_CHECK_LAUNCH("vq_ema_apply");
  return 0;
}
