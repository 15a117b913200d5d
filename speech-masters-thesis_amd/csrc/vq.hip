// Vector-quantiser kernels for gfx950: exact nearest-code search, dequantise,
// commit/fit terms, straight-through backward, codebook EMA statistics and update.
//
// Replaces BottleneckBlock.quantize/dequantize/update_k of the reference
// (models/vqvae/bottleneck.py:60-90, 126-145, 171-201).
//
// Index semantics (oracle/vqvae_oracle.py: vq_argmin_exact): idx = the exact
// argmin_j ||x - k_j||^2 of the fp32 inputs, lowest j on ties.  Implementation:
//   0. prepare     (only when the codebook changes: smt_vq_prepare / smt_vq_ema_apply) codebook mean, centred bf16-pair
//                  split of the codes, -|k~|^2/2, max |k~|^2 -- kept in a persistent `prep` buffer;
//   1. vq_search   one workgroup sweeps ALL codes for its 64 rows on the matrix cores (3 x v_mfma_f32_32x32x16_bf16 per
//                  k-step on the bf16-pair split of the fp32 operands = a FILTER score); best / runner-up never leave
//                  registers; rows whose gap clears the rigorous round-off bound are finished in the same kernel
//                  (idx, min_dist, x_d), the others are queued with their threshold best - 2 err;
//   2. vq_candidates  queued rows only: the same MFMA sweep (code range split over workgroups) collects every code whose
//                  filter score reaches the row's threshold -- the exact argmin is provably among them;
//   3. vq_exact    scores just those in fp64, index order, no contraction (the oracle's arithmetic), lowest index on
//                  ties; a row with too many candidates (degenerate codebooks) scans ALL codes the same way;
//   4. vq_reduce   fixed-order sums (deterministic commit / fit).
// The [N, K] distance matrix is never materialised.
#include <algorithm>

#include "smt_common.h"

#ifndef VQ_ABL
#define VQ_ABL 0      // timing experiments (tools/ablate_vq.sh, results invalid): 1 no MFMAs, 2 no re-staging of the codebook,
#endif                // 4 no best/runner-up folding, 8 no epilogue, 16 phase timestamps (tools/vq_phases.py), 32 no fragment reads

namespace smt {

#if VQ_ABL & 16     // phase timestamps of vq_search_kernel (100 MHz wall clock), tools/ablate_vq.sh only
__device__ long long vq_dbg[8192 * 6];
#define VQ_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) vq_dbg[blockIdx.x * 6 + (k)] = wall_clock64(); } while (0)
#else
#define VQ_STAMP(k) do { } while (0)
#endif

constexpr int VQ_MAXRG = 5;           // 32-row MFMA column groups per workgroup of the search kernel (two waves each)
constexpr int VQ_SSUP = 128;          // codes staged per step of the search kernel: two 32-code chunks per wave
constexpr int VQ_CSUP = 64;           // ... of the candidates kernel: one chunk for each of its two waves
constexpr int VQ_KPAD = 256;          // the prep pads the codebook to a multiple of this (two search steps)
constexpr int VQ_CHUNK = 32;
constexpr int VQ_SPLITS = 8;          // code-range splits of the candidate sweep (one workgroup each)
constexpr int VQ_CAPS = 4;            // candidate codes kept per queued row and split
constexpr int VQ_PART = 8;            // codes per partial column sum (prepare)

typedef __bf16 vq_bf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------- prepare ---
// Distances are translation invariant, so the filter runs on data centred at the codebook mean mu: trained encoders
// emit rows with a large common offset, and the round-off bound that decides which rows need exact re-scoring scales
// with the NORMS of the operands, not their spread.  Column sums: per-part in index order, parts in index order.
struct VqPrep {
  float* mu;          // [D]
  float* nkhalf;      // [Kpad]   -0.5 |k~_j|^2, -3e38 for the padding codes j >= K
  unsigned* kmax2;    // max_j |k~_j|^2 (float bits); kmax2[32] = number of queued rows of the forward in flight (zero between calls)
  __bf16* kh;         // [Kpad][D] high halves of k~ = k - mu (zero rows for padding)
  __bf16* kl;         // [Kpad][D] low halves
  float* part;        // [ceil(K / VQ_PART)][D] partial column sums
  double* dkpart;     // [ceil(K / VQ_PART)]    partial sums of (k_new - k_old)^2 (EMA apply)
  int kpad, nparts;
};
static size_t vq_prep_layout(int K, int D, void* base, VqPrep* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return (char*)base + o; };
  const int kpad = (K + VQ_KPAD - 1) / VQ_KPAD * VQ_KPAD, nparts = (K + VQ_PART - 1) / VQ_PART;
  char* p;
  p = take((size_t)D * 4);            if (w) w->mu = (float*)p;
  p = take((size_t)kpad * 4);         if (w) w->nkhalf = (float*)p;
  p = take(256);                      if (w) w->kmax2 = (unsigned*)p;
  p = take((size_t)kpad * D * 2);     if (w) w->kh = (__bf16*)p;
  p = take((size_t)kpad * D * 2);     if (w) w->kl = (__bf16*)p;
  p = take((size_t)nparts * D * 4);   if (w) w->part = (float*)p;
  p = take((size_t)nparts * 8);       if (w) w->dkpart = (double*)p;
  if (w) { w->kpad = kpad; w->nparts = nparts; }
  return off;
}

// part[p][i] = sum of k[j][i] over the codes j of part p (index order)
__global__ __launch_bounds__(128) void vq_colsum_kernel(const float* __restrict__ cb, int K, int D, float* __restrict__ part) {
  const int i = threadIdx.x;
  if (i >= D) return;
  const int j0 = blockIdx.x * VQ_PART;
  float s = 0.f;
  for (int j = j0; j < min(K, j0 + VQ_PART); ++j) s += cb[(size_t)j * D + i];
  part[(size_t)blockIdx.x * D + i] = s;
}

// Single workgroup: mu from the partial column sums; with `cnt` also the metrics of update_k (bottleneck.py:85-90),
// all reductions in fixed order.
__global__ __launch_bounds__(1024) void vq_mu_kernel(const float* __restrict__ part, int nparts, int K, int D,
                                                     float* __restrict__ mu, unsigned* __restrict__ kmax2_bits,
                                                     const float* __restrict__ cnt, const float* __restrict__ k_elem,
                                                     const double* __restrict__ dkpart, float threshold,
                                                     float* __restrict__ metrics) {
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
  if ((int)threadIdx.x < D) {
    float t = 0.f;
#pragma unroll 16
    for (int p = 0; p < nparts; ++p) t += part[(size_t)p * D + threadIdx.x];   // index order; 16 loads in flight
    mu[threadIdx.x] = t / (float)K;
  }
  if (threadIdx.x == 0) { kmax2_bits[0] = 0u; kmax2_bits[32] = 0u; }     // + the queue counter of smt_vq_forward
  if (!cnt) return;                                        // workgroup-uniform
  double tot = 0.0;
  for (int j = threadIdx.x; j < K; j += blockDim.x) tot += cnt[j];
  const float total = (float)block_sum(tot);
  double ent = 0.0, used = 0.0, usage_n = 0.0, dk2 = 0.0;
  for (int j = threadIdx.x; j < K; j += blockDim.x) {
    const float c = cnt[j];
    const float prob = c / total;
    ent += -(double)(prob * logf(fmaxf(prob, 1e-5f)));
    used += (c >= threshold) ? 1.0 : 0.0;
    usage_n += (k_elem[j] >= threshold) ? 1.0 : 0.0;       // k_elem already holds the mixed value
  }
  for (int p = threadIdx.x; p < nparts; p += blockDim.x) dk2 += dkpart[p];
  ent = block_sum(ent);
  used = block_sum(used);
  usage_n = block_sum(usage_n);
  dk2 = block_sum(dk2);
  if (threadIdx.x == 0) {
    metrics[0] = (float)ent;
    metrics[1] = (float)used;
    metrics[2] = (float)usage_n;
    metrics[3] = (float)(sqrt(dk2) / sqrt((double)K * D));
  }
}

// Position of dim i of code j inside its [D] row of the kh / kl tiles: rows are NOT padded (they are copied to LDS by
// linear LDS-DMA), so the 16-byte chunk index is XORed with the row index instead -- the 32 lanes of an MFMA A-fragment
// read (32 consecutive codes, same chunk) then fall into 16 different 16-byte bank groups.
__host__ __device__ __forceinline__ int vq_swz(int j, int D) { return (j / (128 / D)) & (D / 8 - 1); }

// k~[j] = k[j] - mu as the bf16 pair (kh, kl) (v = hi + lo + eps, |eps| <= 2^-18 |v|), chunk-swizzled; nkhalf[j] =
// -0.5 |k~[j]|^2 (fp32, index order per lane then wave tree); kmax2 = max_j |k~[j]|^2.  One wave per code, 16 codes per
// workgroup (one atomicMax per workgroup); padding codes get zero rows and -inf.
__global__ __launch_bounds__(1024) void vq_split_kernel(const float* __restrict__ cb, const float* __restrict__ mu, int K,
                                                        int Kpad, int D, __bf16* __restrict__ kh, __bf16* __restrict__ kl,
                                                        float* __restrict__ nkhalf, unsigned* __restrict__ kmax2_bits) {
  __shared__ float wmax[16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int code = blockIdx.x * 16 + wave;
  float s = 0.f;
  if (code < Kpad) {
    const int sw = vq_swz(code, D);
    for (int i = lane; i < D; i += 64) {
      const float v = code < K ? cb[(size_t)code * D + i] - mu[i] : 0.f;
      const __bf16 hi = (__bf16)v;
      const size_t o = (size_t)code * D + 8 * ((i >> 3) ^ sw) + (i & 7);
      kh[o] = hi;
      kl[o] = (__bf16)(v - (float)hi);
      s = fmaf(v, v, s);
    }
    s = wave_sum(s);
    if (lane == 0) nkhalf[code] = code < K ? -0.5f * s : -3.0e38f;   // padding: never the best, still a finite float
  }
  if (lane == 0) wmax[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = 0.f;
    for (int w = 0; w < 16; ++w) m = fmaxf(m, wmax[w]);
    atomicMax(kmax2_bits, __float_as_uint(m));  // m >= 0: uint order == float order
  }
}

// ---------------------------------------------------------------- search ----
// The codebook is the MFMA A operand (code on the row index i), x the B operand (row on the column index j = lane & 31),
// so every lane owns ONE x row per column group and sees 16 codes per chunk in its accumulator registers: the running
// best / runner-up is pure in-lane work.  Each fp32 operand is split into a bf16 pair and x~.k~ is evaluated as
// kl.xh + kh.xl + kh.xh with fp32 accumulation on top of -|k~|^2/2 -- 3 bf16 MFMAs (16x the fp32-MFMA rate each)
// instead of 8 fp32 MFMAs.  The score is only a FILTER: its error bound (vq_filter_err) decides which rows are
// re-scored exactly, so the index semantics stay exact.
//
// Shape: 2 waves, 64 rows.  Each wave keeps BOTH 32-row column groups of the tile in registers (bf16 pairs of its
// share of the rows) and takes one of the two 32-code chunks of every staged 64-code step, so one A fragment read from
// LDS feeds 6 MFMAs and a workgroup stages the whole codebook exactly once for its 64 rows.  LDS: two stages of
// [hi | lo][64][D + 8] bf16 + 64 floats.
template <int D, int SUP> struct VqGeom {
  static constexpr int NS = D / 16;                       // k-steps per chunk
  static constexpr int TILE_BYTES = SUP * D * 2;          // one staged tile (hi or lo)
  static constexpr int NDMA = TILE_BYTES / 1024;          // 1-KiB LDS-DMA wave-instructions per tile
  static constexpr int BUF_BYTES = 2 * TILE_BYTES + SUP * 4;
};

__device__ __forceinline__ void vq_dma16(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_dst_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void vq_dma4(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_dst_wave_base, 4, 0, 0);
}
// Stage step `sc` (SUP codes: hi tile, lo tile, -|k~|^2/2) into `buf` with LDS-DMA, no register round trip: the nw
// waves of the workgroup issue the 1-KiB pieces in turn.  The data has landed after every issuing wave's
// `s_waitcnt vmcnt(0)` + a barrier.
template <int D, int SUP>
__device__ __forceinline__ void vq_stage(const __bf16* kh, const __bf16* kl, const float* nkhalf, int sc, char* buf,
                                         int nw, int wave, int lane) {
  using G = VqGeom<D, SUP>;
  if ((VQ_ABL & 2) && sc > 1) return;
  for (int q = wave; q < 2 * G::NDMA; q += nw) {
    const int which = q / G::NDMA, piece = q % G::NDMA;
    const __bf16* src = (which ? kl : kh) + (size_t)sc * SUP * D + piece * 512 + lane * 8;
    vq_dma16(src, buf + which * G::TILE_BYTES + piece * 1024);
  }
  if (wave == nw - 1) {
#pragma unroll
    for (int i = 0; i < SUP / 64; ++i) vq_dma4(nkhalf + sc * SUP + 64 * i + lane, buf + 2 * G::TILE_BYTES + 256 * i);
  }
}
__device__ __forceinline__ void vq_stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// this lane's share of row `row` (dims 16 s + 8 h .. + 7 for every k-step s), centred and split; returns its share of
// |x~|^2.  All loads are issued before the first use (callers pass a row index that is always in range).
template <int D>
__device__ __forceinline__ float vq_load_row(const float* __restrict__ x, const float* __restrict__ mu, long long row,
                                             int h, vq_bf16x8* xh, vq_bf16x8* xl) {
  constexpr int NS = D / 16, HB = NS < 4 ? NS : 4;           // k-steps per batch of loads (bounds the live registers)
  float xx = 0.f;
#pragma unroll
  for (int s0 = 0; s0 < NS; s0 += HB) {
    f32x4 v[HB][2], m[HB][2];
#pragma unroll
    for (int s = 0; s < HB; ++s) {
      const f32x4* src = reinterpret_cast<const f32x4*>(x + row * D + 16 * (s0 + s) + 8 * h);
      const f32x4* msrc = reinterpret_cast<const f32x4*>(mu + 16 * (s0 + s) + 8 * h);
      v[s][0] = src[0]; v[s][1] = src[1]; m[s][0] = msrc[0]; m[s][1] = msrc[1];
    }
#pragma unroll
    for (int s = 0; s < HB; ++s) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float c = v[s][e >> 2][e & 3] - m[s][e >> 2][e & 3];
        const __bf16 hi = (__bf16)c;
        xh[s0 + s][e] = hi;
        xl[s0 + s][e] = (__bf16)(c - (float)hi);
        xx = fmaf(c, c, xx);
      }
    }
  }
  return xx;
}

// Error of the filter score against the exact acc = x~.k~ - |k~|^2/2 on the centred operands (norms are the centred
// ones, Cauchy-Schwarz turns sums of products into norm products):
//   bf16-pair split, three of the four partial products kept:  <= 3.01 * 2^-18 |x~| |k~|
//   fp32 accumulation of 3D exact products + the initial term:  <= 1.05 (3D+2) 2^-24 (|x~||k~| + |k~|^2/2)
//   fp32 rounding of khalf and of the centring x - mu, k - mu:  <= 2^-24 ((D+2)|k~|^2/2 + (|x~| + |k~|)^2)
//   position tag in the 4 low mantissa bits of a score (vq_search_kernel):   <= 2^-19 (|x~||k~| + |k~|^2/2)
// with |k~| <= |k~|max; a factor 1.25 of slack covers the MFMA's internal summation order and the fp32 rounding of xx.
__device__ __forceinline__ float vq_filter_err(float xx, float kmax2, int D) {
  const float xk = sqrtf(xx * kmax2);
  const float u24 = 5.9604645e-8f;
  return 1.25f * (3.01f * 64.f * u24 * xk + (1.05f * (float)(3 * D + 2) + 32.f) * u24 * (xk + 0.5f * kmax2) +
                  u24 * (0.5f * (float)(D + 2) * kmax2 + xx + 2.f * xk + kmax2));
}

// One 32-code chunk against NG column groups: acc[g] = -|k~|^2/2 + sum_s (kl.xh + kh.xl + kh.xh), small terms first.
template <int D, int SUP, int NG>
__device__ __forceinline__ void vq_chunk_scores(const char* buf, int chunk, int j, int h, const vq_bf16x8 (*xh)[D / 16],
                                                const vq_bf16x8 (*xl)[D / 16], f32x16* acc) {
  using G = VqGeom<D, SUP>;
  constexpr int NS = D / 16;
  const float* nk = reinterpret_cast<const float*>(buf + 2 * G::TILE_BYTES) + chunk * VQ_CHUNK + 4 * h;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(nk + 8 * q);       // codes 8 q + 4 h + e of the chunk
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      acc[g][4 * q + 0] = v.x; acc[g][4 * q + 1] = v.y; acc[g][4 * q + 2] = v.z; acc[g][4 * q + 3] = v.w;
    }
  }
  const int rowi = chunk * VQ_CHUNK + j, sw = vq_swz(rowi, D);
  const char* ah = buf + rowi * (2 * D);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int so = (VQ_ABL & 32) ? 0 : s;
    const vq_bf16x8 fh = *reinterpret_cast<const vq_bf16x8*>(ah + 16 * ((2 * so + h) ^ sw));
    const vq_bf16x8 fl = *reinterpret_cast<const vq_bf16x8*>(ah + G::TILE_BYTES + 16 * ((2 * so + h) ^ sw));
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (!(VQ_ABL & 1)) {
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, xh[g][s], acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, xl[g][s], acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, xh[g][s], acc[g], 0, 0, 0);
      }
    }
  }
}

// top-2 merge of (best, second, idx) with another candidate triple; equal scores keep the lower index (the gap is then
// zero and the row is re-scored exactly anyway)
__device__ __forceinline__ void vq_merge(float& best, float& second, int& bidx, float ob, float os, int oi) {
  if (ob > best || (ob == best && oi < bidx)) {
    second = fmaxf(best, os); best = ob; bidx = oi;
  } else {
    second = fmaxf(second, ob);
  }
}

// Shape: RG = blockDim / 128 column groups of 32 rows, two waves each.  Wave (rg, c) keeps column group rg in
// registers (bf16 pairs of its share of 32 rows) and takes chunks c and c + 2 of every staged 128-code step.  The host
// picks RG = ceil(rows / 32 / 256) (up to VQ_MAXRG) so that ONE round of workgroups covers all rows with at most one
// column group of imbalance, and a workgroup stages the whole codebook exactly once for its rows (LDS-DMA, double
// buffered, 2 x 64.5 KiB for D = 128: one workgroup per CU, 2 RG waves on its four SIMDs).
template <int D>
__global__ __launch_bounds__(128 * VQ_MAXRG) void vq_search_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                        const float* __restrict__ row_mask, const float* __restrict__ mu,
                                                        const float* __restrict__ nkhalf, unsigned* __restrict__ kmax2_bits,
                                                        const __bf16* __restrict__ kh, const __bf16* __restrict__ kl,
                                                        long long N, int Kpad, long long* __restrict__ idx,
                                                        float* __restrict__ min_dist, float* __restrict__ x_d,
                                                        int* __restrict__ q_rows, float* __restrict__ q_thr) {
  constexpr int SUP = VQ_SSUP;
  using G = VqGeom<D, SUP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nw = blockDim.x >> 6, tile = 16 * nw;           // waves, rows per workgroup
  float* m_best = reinterpret_cast<float*>(smem + 2 * G::BUF_BYTES);
  float* m_second = m_best + tile;
  int* m_idx = reinterpret_cast<int*>(m_second + tile);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int rg = wave >> 1, c = wave & 1;
  const long long row0 = (long long)blockIdx.x * tile;

  char* buf0 = smem;
  char* buf1 = smem + G::BUF_BYTES;
  VQ_STAMP(0);
  vq_stage<D, SUP>(kh, kl, nkhalf, 0, buf0, nw, wave, lane);
  vq_stage<D, SUP>(kh, kl, nkhalf, 1, buf1, nw, wave, lane);
  vq_bf16x8 xh[1][G::NS], xl[1][G::NS];
  float xx;
  {
    const long long row = min(row0 + 32 * rg + j, N - 1);   // rows past the end repeat the last row and are never written
    xx = vq_load_row<D>(x, mu, row, h, xh[0], xl[0]);
    xx += __shfl_xor(xx, 32, 64);
  }
  vq_stage_wait();
  __syncthreads();
  VQ_STAMP(1);

  float best[1] = {-INFINITY}, second[1] = {-INFINITY};
  int bidx[1] = {0x7fffffff};
  const int nsc = Kpad / SUP;                               // even (the prep pads to 256 codes)
  f32x16 acc[1];
  // The running best / runner-up costs VALU issue slots that the matrix pipe cannot hide (measured: it adds to the MFMA
  // time), so it is cut to 3 instructions per score: the score's position r in the chunk is written into its 4 low
  // mantissa bits (error <= 2^-19 |score|, accounted for in vq_filter_err), after which max / med3 on the tagged floats
  // carry the position along; which chunk holds the best is noted once per chunk.
  int bchunk = 0;
  float pinf = INFINITY;
  asm volatile("" : "+v"(pinf));
  auto score_step = [&](const char* buf, int sc) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int chunk = c + 2 * q;
      vq_chunk_scores<D, SUP, 1>(buf, chunk, j, h, xh, xl, acc);
      if (!(VQ_ABL & 4)) {
        const float before = best[0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {            // v_and_or_b32, v_med3_f32, v_med3_f32 (plain builtins: the compiler
          const float t = __int_as_float((__float_as_int(acc[0][r]) & ~15) | r);   // pads the MFMA -> VALU hazards itself)
          second[0] = __builtin_amdgcn_fmed3f(best[0], second[0], t);
          best[0] = __builtin_amdgcn_fmed3f(best[0], t, pinf);                     // = max; pinf is opaque, so it stays one med3
        }
        bchunk = best[0] != before ? sc * (SUP / VQ_CHUNK) + chunk : bchunk;
      } else {
        best[0] += acc[0][0] + acc[0][5] + acc[0][10] + acc[0][15];      // keeps the MFMA chain alive
      }
    }
  };
  for (int sc = 0; sc < nsc; sc += 2) {
    score_step(buf0, sc);
    vq_stage_wait();                                        // step sc + 1 (issued one scoring phase ago) has landed in buf1
    __syncthreads();                                        // buf0 released
    if (sc + 2 < nsc) vq_stage<D, SUP>(kh, kl, nkhalf, sc + 2, buf0, nw, wave, lane);
    score_step(buf1, sc + 1);
    vq_stage_wait();
    __syncthreads();                                        // buf1 released, step sc + 2 landed in buf0
    if (sc + 3 < nsc) vq_stage<D, SUP>(kh, kl, nkhalf, sc + 3, buf1, nw, wave, lane);   // lands while step sc + 2 is scored
  }
  VQ_STAMP(2);
  {  // decode the winner: chunk, position tag, lane half
    const int r = __float_as_int(best[0]) & 15;
    bidx[0] = bchunk * VQ_CHUNK + 8 * (r >> 2) + 4 * h + (r & 3);
  }
  // merge the two lane halves of a row (same row, disjoint codes), then the two chunk waves of the column group
  {
    const float ob = __shfl_xor(best[0], 32, 64), os = __shfl_xor(second[0], 32, 64);
    const int oi = __shfl_xor(bidx[0], 32, 64);
    vq_merge(best[0], second[0], bidx[0], ob, os, oi);
  }
  float b = best[0], s2 = second[0];
  int bi = bidx[0];
  if (c == 1 && h == 0) { m_best[32 * rg + j] = b; m_second[32 * rg + j] = s2; m_idx[32 * rg + j] = bi; }
  __syncthreads();
  if (c == 0 && h == 0) {
    const int r = 32 * rg + j;
    vq_merge(b, s2, bi, m_best[r], m_second[r], m_idx[r]);
    const long long row = row0 + r;
    const float err = vq_filter_err(xx, __uint_as_float(kmax2_bits[0]), D);
    const bool ambiguous = !((b - s2) > 2.0f * err);        // also catches NaN / inf
    int res = bi;
    if (row < N && ambiguous) {
      const int q = atomicAdd(reinterpret_cast<int*>(kmax2_bits + 32), 1);
      q_rows[q] = (int)row;
      q_thr[q] = b - 2.0f * err;
      res = -1;
    }
    m_idx[r] = res;
  }
  __syncthreads();
  VQ_STAMP(3);
  // finish the unambiguous rows: idx, min_dist = |x - k_idx|^2 (fp32 direct form), x_d = k_idx * mask.
  // LPR lanes per row, RPI rows per wave-instruction, 16 rows per wave; loads of a batch are issued together.
  constexpr int RPW = 16, LPR = D / 4, RPI = 64 / LPR, NIT = RPW / RPI, BATCH = NIT < 4 ? NIT : 4;
  const int c4 = lane % LPR;
#pragma unroll 1
  for (int it0 = 0; it0 < ((VQ_ABL & 8) ? 0 : NIT); it0 += BATCH) {
    f32x4 xv[BATCH], kv[BATCH];
    int code[BATCH];
    long long rows[BATCH];
#pragma unroll
    for (int q = 0; q < BATCH; ++q) {
      const int r = RPW * wave + (it0 + q) * RPI + lane / LPR;
      rows[q] = row0 + r;
      code[q] = m_idx[r];
      xv[q] = *reinterpret_cast<const f32x4*>(x + min(rows[q], N - 1) * D + 4 * c4);
      kv[q] = *reinterpret_cast<const f32x4*>(cb + (size_t)max(code[q], 0) * D + 4 * c4);
    }
#pragma unroll
    for (int q = 0; q < BATCH; ++q) {
      const f32x4 df = xv[q] - kv[q];
      float ds = fmaf(df.w, df.w, fmaf(df.z, df.z, fmaf(df.y, df.y, df.x * df.x)));
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) ds += __shfl_xor(ds, o, 64);
      if (rows[q] < N && code[q] >= 0) {
        if (x_d) {
          const float m = row_mask ? row_mask[rows[q]] : 1.f;
          *reinterpret_cast<f32x4*>(x_d + rows[q] * D + 4 * c4) = kv[q] * m;
        }
        if (c4 == 0) { idx[rows[q]] = code[q]; min_dist[rows[q]] = ds; }
      }
    }
  }
  VQ_STAMP(4);
}

// ---------------------------------------------------------------- candidates
// Queued rows, 32 at a time, the code range cut into `splits` pieces (one workgroup per (row group, piece)): with filter
// scores f_j (|f_j - s_j| <= err against the exact scores s_j) and the first pass's maximum fmax, the exact winner j*
// satisfies f_j* >= s_j* - err >= s_jmax - err >= fmax - 2 err for ANY filter values within err of the truth -- so every
// code whose score here reaches thr = fmax - 2 err is recorded (up to VQ_CAPS per row and piece; the count is recorded
// in full so that vq_exact sees an overflow).
template <int D>
__global__ __launch_bounds__(128) void vq_candidates_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                            const float* __restrict__ nkhalf, const unsigned* __restrict__ kmax2_bits,
                                                            const __bf16* __restrict__ kh, const __bf16* __restrict__ kl,
                                                            int Kpad, int splits, const int* __restrict__ q_rows,
                                                            const float* __restrict__ q_thr, int* __restrict__ c_count,
                                                            int* __restrict__ c_codes) {
  constexpr int SUP = VQ_CSUP;
  using G = VqGeom<D, SUP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int l_cnt[32];
  __shared__ int l_code[32][VQ_CAPS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int n_q = (int)kmax2_bits[32];
  const int nsc = Kpad / SUP, per = nsc / splits;      // staged steps per piece
  const int ntask = ((n_q + 31) / 32) * splits;
  for (int task = blockIdx.x; task < ntask; task += gridDim.x) {
    const int rg = task / splits, sp = task % splits;
    __syncthreads();                                        // the previous task's lists are no longer read
    if (threadIdx.x < 32) l_cnt[threadIdx.x] = 0;
    const int qi = min(rg * 32 + j, n_q - 1);
    const float thr = rg * 32 + j < n_q ? q_thr[qi] : INFINITY;
    vq_stage<D, SUP>(kh, kl, nkhalf, sp * per, smem, 2, wave, lane);
    vq_bf16x8 xh[1][G::NS], xl[1][G::NS];
    (void)vq_load_row<D>(x, mu, (long long)q_rows[qi], h, xh[0], xl[0]);
    vq_stage_wait();
    __syncthreads();
    for (int i = 0; i < per; ++i) {
      const int buf = i & 1, sc = sp * per + i;
      if (i + 1 < per) vq_stage<D, SUP>(kh, kl, nkhalf, sc + 1, smem + (buf ^ 1) * G::BUF_BYTES, 2, wave, lane);
      f32x16 acc[1];
      vq_chunk_scores<D, SUP, 1>(smem + buf * G::BUF_BYTES, wave, j, h, xh, xl, acc);
      const int cbase = sc * SUP + wave * VQ_CHUNK + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (acc[0][r] >= thr) {                             // padding codes score -inf, NaN compares false
          const int pos = atomicAdd(&l_cnt[j], 1);
          if (pos < VQ_CAPS) l_code[j][pos] = cbase + 8 * (r >> 2) + (r & 3);
        }
      }
      vq_stage_wait();
      __syncthreads();
    }
    if (threadIdx.x < 32 && rg * 32 + (int)threadIdx.x < n_q) {
      const size_t slot = (size_t)(rg * 32 + threadIdx.x) * splits + sp;
      c_count[slot] = l_cnt[threadIdx.x];
#pragma unroll
      for (int c = 0; c < VQ_CAPS; ++c) c_codes[slot * VQ_CAPS + c] = l_code[threadIdx.x][c];
    }
  }
}

// ---------------------------------------------------------------- exact -----
// d_j = sum_i (x_i - k_ji)^2 in fp64, index order, no fma contraction (the numpy float64 loop of the oracle)
__device__ __forceinline__ double vq_exact_dist(const float* __restrict__ xr, const float* __restrict__ kr, int D) {
  double d = 0.0;
#pragma unroll 4
  for (int i = 0; i < D; i += 4) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + i), kv = *reinterpret_cast<const f32x4*>(kr + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double df = __dsub_rn((double)xv[e], (double)kv[e]);
      d = __dadd_rn(d, __dmul_rn(df, df));
    }
  }
  return d;
}
// Half a wave per queued row, one lane per candidate slot (splits * VQ_CAPS <= 32); lowest index among equal distances.
// A row with no candidate or an overflowing piece (NaN input, degenerate codebook) scans all K codes the same way.
__global__ __launch_bounds__(256) void vq_exact_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                       const float* __restrict__ row_mask, const unsigned* __restrict__ kmax2_bits,
                                                       int K, int D, int splits, const int* __restrict__ q_rows,
                                                       const int* __restrict__ c_count, const int* __restrict__ c_codes,
                                                       long long* __restrict__ idx, float* __restrict__ min_dist,
                                                       float* __restrict__ x_d) {
  const int n_q = (int)kmax2_bits[32];
  const int lane = threadIdx.x & 63, half = lane >> 5, l = lane & 31;
  const int hw0 = (((int)blockIdx.x * 256 + (int)threadIdx.x) >> 6) * 2, nhw = (int)gridDim.x * 8;
  for (int q0 = hw0; q0 < n_q; q0 += nhw) {                  // wave-uniform trip count; the second half may idle
    const int qi = q0 + half;
    const bool rvalid = qi < n_q;
    const long long row = rvalid ? q_rows[qi] : 0;
    const float* xr = x + row * D;
    const int sp = l / VQ_CAPS, c = l % VQ_CAPS;
    int n = 0;
    if (rvalid && sp < splits) n = c_count[(size_t)qi * splits + sp];
    // any piece over capacity, or no candidate at all, anywhere in this half-wave's row?
    const unsigned long long over = __ballot(n > VQ_CAPS), some = __ballot(n > 0);
    const unsigned long long hmask = half ? 0xffffffff00000000ull : 0x00000000ffffffffull;
    const bool full = rvalid && ((over & hmask) != 0 || (some & hmask) == 0);
    double d = INFINITY;
    int code = 0x7fffffff;
    if (rvalid && !full && c < n) {
      code = c_codes[((size_t)qi * splits + sp) * VQ_CAPS + c];
      d = vq_exact_dist(xr, cb + (size_t)code * D, D);
    }
    if (full) {
      for (int c0 = l; c0 < K; c0 += 32) {                   // increasing code order per lane: strict '<' keeps the lowest
        const double dc = vq_exact_dist(xr, cb + (size_t)c0 * D, D);
        if (dc < d) { d = dc; code = c0; }
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const double od = __shfl_xor(d, o, 64);
      const int oc = __shfl_xor(code, o, 64);
      if (od < d || (od == d && oc < code)) { d = od; code = oc; }
    }
    if (rvalid && code != 0x7fffffff) {
      if (x_d) {
        const float m = row_mask ? row_mask[row] : 1.f;
        for (int i = l; i < D; i += 32) x_d[row * D + i] = cb[(size_t)code * D + i] * m;
      }
      if (l == 0) { idx[row] = code; min_dist[row] = (float)d; }   // the exact distance, rounded once
    } else if (rvalid && l == 0) {                                 // NaN row: keep the outputs defined
      idx[row] = 0; min_dist[row] = __builtin_nanf("");
    }
  }
}

// ---------------------------------------------------------------- reduce ----
// Single workgroup, fixed order: sums[0] = sum_all min_dist, sums[1] = sum_masked, sums[2] = sum mask,
// sums[3] = rows that were re-scored exactly.  Resets the queue counter for the next forward.
__global__ __launch_bounds__(1024) void vq_reduce_kernel(const float* __restrict__ min_dist,
                                                         const float* __restrict__ row_mask, long long N,
                                                         unsigned* __restrict__ kmax2_bits, float* __restrict__ sums) {
  __shared__ double sh[3][16];
  double a = 0.0, b = 0.0, c = 0.0;
  constexpr int U = 10;                                     // 16-byte loads in flight per thread: 40,960 rows per pass
  for (long long r0 = 4ll * threadIdx.x; r0 < N; r0 += 4096ll * U) {
    f32x4 d[U], m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long r = r0 + 4096ll * u;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f}, one = {1.f, 1.f, 1.f, 1.f};
      if (r + 3 < N) {
        d[u] = *reinterpret_cast<const f32x4*>(min_dist + r);
        m[u] = row_mask ? *reinterpret_cast<const f32x4*>(row_mask + r) : one;
      } else {
        d[u] = z; m[u] = z;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (r + e < N) { d[u][e] = min_dist[r + e]; m[u][e] = row_mask ? row_mask[r + e] : 1.f; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) { a += d[u][e]; b += (m[u][e] != 0.f) ? d[u][e] : 0.f; c += m[u][e]; }
  }
  a = wave_sum_d(a); b = wave_sum_d(b); c = wave_sum_d(c);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[0][wave] = a; sh[1][wave] = b; sh[2][wave] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0, tb = 0, tc = 0;
    for (int w = 0; w < 16; ++w) { ta += sh[0][w]; tb += sh[1][w]; tc += sh[2][w]; }
    sums[0] = (float)ta; sums[1] = (float)tb; sums[2] = (float)tc; sums[3] = (float)kmax2_bits[32];
    kmax2_bits[32] = 0u;
  }
}

// ---------------------------------------------------------------- backward --
__global__ __launch_bounds__(256) void vq_backward_kernel(const float* __restrict__ x, const float* __restrict__ x_d,
                                                          const float* __restrict__ row_mask, const float* __restrict__ dy,
                                                          const float* __restrict__ g_commit, const float* __restrict__ sums,
                                                          long long N, int D, float* __restrict__ dx) {
  const long long total4 = N * D / 4;
  const float gc = g_commit ? (*g_commit) * 2.0f / (sums[2] * (float)D) : 0.f;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total4;
       e += (long long)gridDim.x * blockDim.x) {
    const long long row = (e * 4) / D;
    const float m = row_mask ? row_mask[row] : 1.f;
    f32x4 xv = *reinterpret_cast<const f32x4*>(x + e * 4);
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    if (dy) {
      f32x4 g = *reinterpret_cast<const f32x4*>(dy + e * 4);
      out = g * m;
    }
    if (g_commit && m != 0.f) {
      f32x4 kv = *reinterpret_cast<const f32x4*>(x_d + e * 4);      // = k[idx[row]] on unmasked rows (0/1 masks)
      out += (xv - kv) * gc;
    }
    *reinterpret_cast<f32x4*>(dx + e * 4) = out;
  }
}

// ---------------------------------------------------------------- EMA -------
// _k_sum = onehot^T x, _k_elem = column sums of onehot (bottleneck.py:64-68) as a scatter-add.  The sums are accumulated
// in 64-bit FIXED POINT (2^-24 units) with integer atomics: integer addition is associative, so the result does not
// depend on the order in which rows arrive -- bit-reproducible from run to run, which f32 atomics are not -- and is
// converted to f32 once at the end.  A value is exact when it is a multiple of 2^-24 below 2^15 in magnitude (every bf16
// and every f32 encoder output of ordinary size); larger magnitudes saturate at +-2^39 units so that 2^23 rows cannot
// overflow the accumulator.  One wave per row; 512 contiguous bytes per atomic wave-instruction.
constexpr float VQ_FX_SCALE = 16777216.f;            // 2^24
constexpr float VQ_FX_LIMIT = 549755813888.f;        // 2^39

// Round 3: the scatter-add no longer issues one atomic per (row, channel).  Under skewed usage -- a trained codebook
// concentrates on a few codes, a collapsing one on very few -- those atomics pile up on the same lines (measured: 47-400 us
// over four train steps, 548 us in the round-2 driver run, against 75 us on uniform usage).  The rows are first grouped by
// code with a counting sort whose histogram and cursors are privatised in LDS (integer atomics only, at most one global
// atomic per (workgroup, code)), then waves walk equal shares of the sorted order and keep the running sum of the current
// code in registers: one 64-bit integer atomic per channel per (share, code) boundary instead of per row -- 23x fewer on
// uniform usage, and the more skewed the usage, the fewer.  Integer sums: any order gives the same bits.
constexpr int VQ_EMA_HIST_NT = 1024;
// counts[code] += rows of this workgroup's share with that code (masked rows excluded)
__global__ __launch_bounds__(VQ_EMA_HIST_NT) void vq_ema_count_kernel(const long long* __restrict__ idx, const float* __restrict__ row_mask,
                                                                     long long N, int K, int rows_per_wg, int* __restrict__ counts) {
  extern __shared__ int hist[];
  for (int k = threadIdx.x; k < K; k += VQ_EMA_HIST_NT) hist[k] = 0;
  __syncthreads();
  const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(N, r0 + rows_per_wg);
  for (long long r = r0 + threadIdx.x; r < r1; r += VQ_EMA_HIST_NT)
    if (!row_mask || row_mask[r] != 0.f) atomicAdd(&hist[(int)idx[r]], 1);
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += VQ_EMA_HIST_NT)
    if (hist[k]) atomicAdd(&counts[k], hist[k]);
}
// exclusive scan of counts -> cursor (one workgroup; K <= a few thousand); total[0] = number of unmasked rows
__global__ __launch_bounds__(1024) void vq_ema_scan_kernel(const int* __restrict__ counts, int K, int* __restrict__ cursor, int* __restrict__ total) {
  __shared__ int part[1024];
  const int per = (K + 1023) / 1024, k0 = threadIdx.x * per;
  int s = 0;
  for (int k = k0; k < min(K, k0 + per); ++k) s += counts[k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;
  for (int k = k0; k < min(K, k0 + per); ++k) { cursor[k] = run; run += counts[k]; }
  if (threadIdx.x == 1023) total[0] = part[1023];
}
// order[cursor[code]++] = row, with the cursor advanced once per (workgroup, code)
__global__ __launch_bounds__(VQ_EMA_HIST_NT) void vq_ema_scatter_kernel(const long long* __restrict__ idx, const float* __restrict__ row_mask,
                                                                       long long N, int K, int rows_per_wg, int* __restrict__ cursor,
                                                                       int* __restrict__ order) {
  extern __shared__ int sm[];            // hist [K] | base [K]
  int* hist = sm;
  int* base = sm + K;
  for (int k = threadIdx.x; k < K; k += VQ_EMA_HIST_NT) hist[k] = 0;
  __syncthreads();
  const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(N, r0 + rows_per_wg);
  for (long long r = r0 + threadIdx.x; r < r1; r += VQ_EMA_HIST_NT)
    if (!row_mask || row_mask[r] != 0.f) atomicAdd(&hist[(int)idx[r]], 1);
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += VQ_EMA_HIST_NT) {
    base[k] = hist[k] ? atomicAdd(&cursor[k], hist[k]) : 0;
    hist[k] = 0;
  }
  __syncthreads();
  for (long long r = r0 + threadIdx.x; r < r1; r += VQ_EMA_HIST_NT)
    if (!row_mask || row_mask[r] != 0.f) {
      const int code = (int)idx[r];
      order[base[code] + atomicAdd(&hist[code], 1)] = (int)r;
    }
}
// one wave per share of VQ_EMA_SHARE sorted rows; PER = ceil(D / 64) channels per lane (D = 32: the upper half of the wave idles)
constexpr int VQ_EMA_SHARE = 64;
// A wave owns 64 consecutive positions of the sorted order.  Lane l fetches the row id and the code of position r0 + l once
// (one coalesced load + one gather); the rows are then read sixteen at a time with wave-uniform row ids (v_readlane), so a
// share costs ~6 memory round trips.  (The first version walked 128 positions four at a time with order -> idx -> x as
// dependent loads: 2 x 32 round trips per wave, 119 us of latency for 19 MB.)
template <int D>
__global__ __launch_bounds__(256) void vq_ema_accumulate_kernel(const float* __restrict__ x, const long long* __restrict__ idx,
                                                                const int* __restrict__ order, const int* __restrict__ total, int K,
                                                                unsigned long long* __restrict__ acc) {
  constexpr int PER = (D + 63) / 64;
  const int lane = threadIdx.x & 63;
  const bool live = lane < D;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long long r0 = wave * VQ_EMA_SHARE, r1 = min((long long)total[0], r0 + VQ_EMA_SHARE);
  if (r0 >= r1) return;
  const int n = (int)(r1 - r0);
  const int my_row = lane < n ? order[r0 + lane] : 0;
  const int my_code = lane < n ? (int)idx[my_row] : -1;
  long long sum[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) sum[q] = 0;
  int cur = -1, cnt = 0;
  auto flush = [&]() {
    if (cur < 0) return;
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (live) atomicAdd(acc + (size_t)cur * D + lane + 64 * q, (unsigned long long)sum[q]);   // two's complement
    if (lane == 0) atomicAdd(acc + (size_t)K * D + cur, (unsigned long long)cnt);
  };
  for (int g = 0; g < n; g += 16) {
    float v[16][PER];
    int code[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int pos = min(g + u, n - 1);                             // wave-uniform
      const int row = __builtin_amdgcn_readlane(my_row, pos);
      code[u] = g + u < n ? __builtin_amdgcn_readlane(my_code, pos) : -1;
#pragma unroll
      for (int q = 0; q < PER; ++q) v[u][q] = live ? x[(size_t)row * D + lane + 64 * q] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (code[u] < 0) continue;                                     // uniform
      if (code[u] != cur) {
        flush();
        cur = code[u]; cnt = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) sum[q] = 0;
      }
#pragma unroll
      for (int q = 0; q < PER; ++q)
        sum[q] += __float2ll_rn(fminf(fmaxf(v[u][q] * VQ_FX_SCALE, -VQ_FX_LIMIT), VQ_FX_LIMIT));
      ++cnt;
    }
  }
  flush();
}
__global__ __launch_bounds__(256) void vq_ema_convert_kernel(const unsigned long long* __restrict__ acc, float* __restrict__ stats,
                                                             int n_sums, int n_total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_total) return;
  const long long v = (long long)acc[e];
  stats[e] = e < n_sums ? (float)((double)v * (1.0 / 16777216.0)) : (float)v;
}

// EMA mix + revival for VQ_PART codes per workgroup (bottleneck.py:78-84); leaves the partial column sums of the NEW
// codebook and the partial sums of (k_new - k_old)^2 for vq_mu_kernel, which finishes the metrics in fixed order.
__global__ __launch_bounds__(256) void vq_ema_apply_kernel(float* __restrict__ cb, float* __restrict__ k_sum,
                                                           float* __restrict__ k_elem, const float* __restrict__ stats,
                                                           const float* __restrict__ k_rand, float mu, float threshold,
                                                           int K, int D, float* __restrict__ part, double* __restrict__ dkpart) {
  __shared__ float slab[VQ_PART * 128];
  __shared__ double red[4];
  const int j0 = blockIdx.x * VQ_PART;
  const int n = min(VQ_PART, K - j0) * D;
  const float* cnt = stats + (size_t)K * D;
  double dk2 = 0.0;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int j = j0 + e / D;
    const size_t ge = (size_t)j0 * D + e;
    const float ne = mu * k_elem[j] + (1.f - mu) * cnt[j];   // k_elem is rewritten only after the barrier below
    const float ns = mu * k_sum[ge] + (1.f - mu) * stats[ge];
    const float usage = (ne >= threshold) ? 1.f : 0.f;
    const float nk = usage * (ns / ne) + (1.f - usage) * k_rand[ge];
    const float d = nk - cb[ge];
    dk2 += (double)d * d;
    k_sum[ge] = ns;
    cb[ge] = nk;
    slab[e] = nk;
  }
  dk2 = wave_sum_d(dk2);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dk2;
  __syncthreads();
  if ((int)threadIdx.x < min(VQ_PART, K - j0)) {
    const int j = j0 + threadIdx.x;
    k_elem[j] = mu * k_elem[j] + (1.f - mu) * cnt[j];
  }
  if (threadIdx.x == 0) dkpart[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
  if ((int)threadIdx.x < D) {
    float t = 0.f;
    for (int q = 0; q < min(VQ_PART, K - j0); ++q) t += slab[q * D + threadIdx.x];
    part[(size_t)blockIdx.x * D + threadIdx.x] = t;
  }
}

struct VqWorkspace { int* q_rows; float* q_thr; int* c_count; int* c_codes; void* prep; };

static size_t vq_layout(long long N, int K, int D, void* base, VqWorkspace* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return (char*)base + o; };
  char* p;
  p = take((size_t)N * 4); if (w) w->q_rows = (int*)p;
  p = take((size_t)N * 4); if (w) w->q_thr = (float*)p;
  p = take((size_t)N * VQ_SPLITS * 4);           if (w) w->c_count = (int*)p;
  p = take((size_t)N * VQ_SPLITS * VQ_CAPS * 4); if (w) w->c_codes = (int*)p;
  p = take(vq_prep_layout(K, D, nullptr, nullptr)); if (w) w->prep = p;   // used when the caller passes no prep buffer
  return off;
}

// column sums -> mu (-> metrics) -> centred split: the three launches behind smt_vq_prepare and smt_vq_ema_apply
static int vq_finish_prepare(const float* cb, int K, int D, const VqPrep& pr, const float* cnt, const float* k_elem,
                             float threshold, float* metrics, hipStream_t stream) {
  vq_mu_kernel<<<1, 1024, 0, stream>>>(pr.part, pr.nparts, K, D, pr.mu, pr.kmax2, cnt, k_elem, pr.dkpart, threshold, metrics);
  SMT_CHECK_LAUNCH("vq_mu");
  vq_split_kernel<<<(pr.kpad + 15) / 16, 1024, 0, stream>>>(cb, pr.mu, K, pr.kpad, D, pr.kh, pr.kl, pr.nkhalf, pr.kmax2);
  SMT_CHECK_LAUNCH("vq_split");
  return 0;
}

template <int D>
static int vq_launch_search(const float* x, const float* cb, const float* row_mask, const VqPrep& pr, long long N, int K,
                            long long* idx, float* min_dist, float* x_d, const VqWorkspace& w, hipStream_t stream) {
  using GS = VqGeom<D, VQ_SSUP>;
  using GC = VqGeom<D, VQ_CSUP>;
  const long long groups = (N + 31) / 32;
  // column groups per workgroup: one round of at most 256 workgroups when that fits (<= VQ_MAXRG groups each)
  const int rgs = (int)std::min<long long>(VQ_MAXRG, (groups + 255) / 256);
  const size_t lds = 2 * GS::BUF_BYTES + 3 * 32 * rgs * 4;
  (void)hipFuncSetAttribute((const void*)vq_search_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * GS::BUF_BYTES + 3 * 32 * VQ_MAXRG * 4));
  (void)hipFuncSetAttribute((const void*)vq_candidates_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GC::BUF_BYTES);
  vq_search_kernel<D><<<(unsigned)((groups + rgs - 1) / rgs), 128 * rgs, lds, stream>>>(
      x, cb, row_mask, pr.mu, pr.nkhalf, pr.kmax2, pr.kh, pr.kl, N, pr.kpad, idx, min_dist, x_d, w.q_rows, w.q_thr);
  SMT_CHECK_LAUNCH("vq_search");
  const int nsc = pr.kpad / VQ_CSUP;
  int splits = VQ_SPLITS;
  while (nsc % splits) splits >>= 1;                         // nsc is a multiple of 4
  vq_candidates_kernel<D><<<(unsigned)std::min<long long>(512, groups * splits), 128, 2 * GC::BUF_BYTES, stream>>>(
      x, pr.mu, pr.nkhalf, pr.kmax2, pr.kh, pr.kl, pr.kpad, splits, w.q_rows, w.q_thr, w.c_count, w.c_codes);
  SMT_CHECK_LAUNCH("vq_candidates");
  vq_exact_kernel<<<(unsigned)std::min<long long>(256, (N + 7) / 8), 256, 0, stream>>>(
      x, cb, row_mask, pr.kmax2, K, D, splits, w.q_rows, w.c_count, w.c_codes, idx, min_dist, x_d);
  SMT_CHECK_LAUNCH("vq_exact");
  return 0;
}

}  // namespace smt

using namespace smt;

#if VQ_ABL & 16
extern "C" int smt_vq_debug_dump(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vq_dbg), sizeof(long long) * n);
}
#endif

extern "C" size_t smt_vq_prep_bytes(int k_bins, int dim) { return vq_prep_layout(k_bins, dim, nullptr, nullptr); }

extern "C" int smt_vq_prepare(const float* codebook, int k_bins, int dim, void* prep, size_t prep_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(dim == 32 || dim == 64 || dim == 128, "smt_vq_prepare: dim must be 32, 64 or 128 (got %d)", dim);
  SMT_CHECK_ARG(codebook && prep && k_bins >= 1, "smt_vq_prepare: null pointer / bad size");
  SMT_CHECK_ARG(prep_bytes >= vq_prep_layout(k_bins, dim, nullptr, nullptr), "smt_vq_prepare: prep buffer too small");
  VqPrep pr;
  vq_prep_layout(k_bins, dim, prep, &pr);
  vq_colsum_kernel<<<pr.nparts, 128, 0, stream>>>(codebook, k_bins, dim, pr.part);
  SMT_CHECK_LAUNCH("vq_colsum");
  return vq_finish_prepare(codebook, k_bins, dim, pr, nullptr, nullptr, 0.f, nullptr, stream);
}

extern "C" size_t smt_vq_forward_workspace_bytes(int64_t n_rows, int k_bins, int dim) {
  return vq_layout(n_rows, k_bins, dim, nullptr, nullptr);
}

extern "C" int smt_vq_forward(const float* x, const float* codebook, void* prep, const float* row_mask,
                              int64_t n_rows, int k_bins, int dim, int64_t* idx, float* min_dist, float* x_d, float* sums,
                              void* workspace, size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(dim == 32 || dim == 64 || dim == 128, "smt_vq_forward: dim must be 32, 64 or 128 (got %d)", dim);
  SMT_CHECK_ARG(k_bins >= 1 && n_rows >= 0, "smt_vq_forward: bad sizes n_rows=%lld k_bins=%d", (long long)n_rows, k_bins);
  SMT_CHECK_ARG(n_rows < (1ll << 31), "smt_vq_forward: n_rows must be < 2^31");
  SMT_CHECK_ARG(codebook && sums && workspace, "smt_vq_forward: null pointer");
  SMT_CHECK_ARG(n_rows == 0 || (x && idx && min_dist), "smt_vq_forward: null pointer");
  SMT_CHECK_ARG(workspace_bytes >= vq_layout(n_rows, k_bins, dim, nullptr, nullptr), "smt_vq_forward: workspace too small");
  VqWorkspace w;
  vq_layout(n_rows, k_bins, dim, workspace, &w);
  if (n_rows == 0) {
    (void)hipMemsetAsync(sums, 0, 16, stream);
    return 0;
  }
  VqPrep pr;
  if (prep) {
    vq_prep_layout(k_bins, dim, prep, &pr);
  } else {                                   // no cached split of this codebook: build one in the workspace
    int rc = smt_vq_prepare(codebook, k_bins, dim, w.prep, vq_prep_layout(k_bins, dim, nullptr, nullptr), stream_);
    if (rc) return rc;
    vq_prep_layout(k_bins, dim, w.prep, &pr);
  }
  int rc;
  if (dim == 128) rc = vq_launch_search<128>(x, codebook, row_mask, pr, n_rows, k_bins, (long long*)idx, min_dist, x_d, w, stream);
  else if (dim == 64) rc = vq_launch_search<64>(x, codebook, row_mask, pr, n_rows, k_bins, (long long*)idx, min_dist, x_d, w, stream);
  else rc = vq_launch_search<32>(x, codebook, row_mask, pr, n_rows, k_bins, (long long*)idx, min_dist, x_d, w, stream);
  if (rc) return rc;
  vq_reduce_kernel<<<1, 1024, 0, stream>>>(min_dist, row_mask, n_rows, pr.kmax2, sums);
  SMT_CHECK_LAUNCH("vq_reduce");
  return 0;
}

extern "C" int smt_vq_backward(const float* x, const float* x_d, const float* row_mask, const float* dy,
                               const float* g_commit, const float* sums, int64_t n_rows, int dim, float* dx,
                               smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(dim % 4 == 0, "smt_vq_backward: dim must be a multiple of 4");
  if (n_rows == 0) return 0;
  SMT_CHECK_ARG(x && x_d && sums && dx, "smt_vq_backward: null pointer");
  long long total4 = n_rows * dim / 4;
  unsigned grid = (unsigned)min((long long)2048, (total4 + 255) / 256);
  vq_backward_kernel<<<grid, 256, 0, stream>>>(x, x_d, row_mask, dy, g_commit, sums, n_rows, dim, dx);
  SMT_CHECK_LAUNCH("vq_backward");
  return 0;
}

// workspace: acc u64 [K D + K] | counts int [K] | cursor int [K] | total int [64] | order int [N]
static size_t vq_ema_ws_layout(long long N, int K, int D, size_t* off_counts, size_t* off_cursor, size_t* off_total, size_t* off_order) {
  size_t off = align_up(((size_t)K * D + K) * sizeof(unsigned long long), 256);
  if (off_counts) *off_counts = off;
  off += align_up((size_t)K * sizeof(int), 256);
  if (off_cursor) *off_cursor = off;
  off += align_up((size_t)K * sizeof(int), 256);
  if (off_total) *off_total = off;
  off += 256;
  if (off_order) *off_order = off;
  off += align_up((size_t)std::max<long long>(N, 1) * sizeof(int), 256);
  return off;
}

extern "C" size_t smt_vq_ema_accumulate_workspace_bytes(int64_t n_rows, int k_bins, int dim) {
  return vq_ema_ws_layout(n_rows, k_bins, dim, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int smt_vq_ema_accumulate(const float* x, const int64_t* idx, const float* row_mask, int64_t n_rows, int k_bins,
                                     int dim, float* stats, void* workspace, size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(stats && workspace && (n_rows == 0 || (x && idx)), "smt_vq_ema_accumulate: null pointer");
  SMT_CHECK_ARG(dim == 64 || dim == 128 || dim == 32, "smt_vq_ema_accumulate: dim must be 32, 64 or 128 (got %d)", dim);
  SMT_CHECK_ARG(n_rows < (1ll << 31) && k_bins >= 1 && k_bins <= 16384, "smt_vq_ema_accumulate: bad sizes");
  size_t o_counts, o_cursor, o_total, o_order;
  SMT_CHECK_ARG(workspace_bytes >= vq_ema_ws_layout(n_rows, k_bins, dim, &o_counts, &o_cursor, &o_total, &o_order),
                "smt_vq_ema_accumulate: workspace too small");
  const int n_sums = k_bins * dim, n_total = n_sums + k_bins;
  unsigned long long* acc = (unsigned long long*)workspace;
  int* counts = (int*)((char*)workspace + o_counts);
  int* cursor = (int*)((char*)workspace + o_cursor);
  int* total = (int*)((char*)workspace + o_total);
  int* order = (int*)((char*)workspace + o_order);
  (void)hipMemsetAsync(workspace, 0, o_order, stream);                // accumulators, counts, cursors, total
  if (n_rows > 0) {
    const int rows_per_wg = (int)std::max<long long>(VQ_EMA_HIST_NT, (n_rows + 255) / 256);
    const unsigned nwg = (unsigned)((n_rows + rows_per_wg - 1) / rows_per_wg);
    vq_ema_count_kernel<<<nwg, VQ_EMA_HIST_NT, (size_t)k_bins * sizeof(int), stream>>>((const long long*)idx, row_mask, n_rows, k_bins, rows_per_wg,
                                                                                      counts);
    SMT_CHECK_LAUNCH("vq_ema_count");
    vq_ema_scan_kernel<<<1, 1024, 0, stream>>>(counts, k_bins, cursor, total);
    SMT_CHECK_LAUNCH("vq_ema_scan");
    vq_ema_scatter_kernel<<<nwg, VQ_EMA_HIST_NT, 2 * (size_t)k_bins * sizeof(int), stream>>>((const long long*)idx, row_mask, n_rows, k_bins,
                                                                                          rows_per_wg, cursor, order);
    SMT_CHECK_LAUNCH("vq_ema_scatter");
    const long long waves = (n_rows + VQ_EMA_SHARE - 1) / VQ_EMA_SHARE;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    if (dim == 128) vq_ema_accumulate_kernel<128><<<grid, 256, 0, stream>>>(x, (const long long*)idx, order, total, k_bins, acc);
    else if (dim == 64) vq_ema_accumulate_kernel<64><<<grid, 256, 0, stream>>>(x, (const long long*)idx, order, total, k_bins, acc);
    else vq_ema_accumulate_kernel<32><<<grid, 256, 0, stream>>>(x, (const long long*)idx, order, total, k_bins, acc);
    SMT_CHECK_LAUNCH("vq_ema_accumulate");
  }
  vq_ema_convert_kernel<<<(n_total + 255) / 256, 256, 0, stream>>>(acc, stats, n_sums, n_total);
  SMT_CHECK_LAUNCH("vq_ema_convert");
  return 0;
}

extern "C" int smt_vq_ema_apply(float* codebook, float* k_sum, float* k_elem, const float* stats, const float* k_rand,
                                float mu, float threshold, int k_bins, int dim, float* metrics, void* prep,
                                size_t prep_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(codebook && k_sum && k_elem && stats && k_rand && metrics && prep, "smt_vq_ema_apply: null pointer");
  SMT_CHECK_ARG(dim == 32 || dim == 64 || dim == 128, "smt_vq_ema_apply: dim must be 32, 64 or 128 (got %d)", dim);
  SMT_CHECK_ARG(prep_bytes >= vq_prep_layout(k_bins, dim, nullptr, nullptr), "smt_vq_ema_apply: prep buffer too small");
  VqPrep pr;
  vq_prep_layout(k_bins, dim, prep, &pr);
  vq_ema_apply_kernel<<<pr.nparts, 256, 0, stream>>>(codebook, k_sum, k_elem, stats, k_rand, mu, threshold, k_bins, dim,
                                                     pr.part, pr.dkpart);
  SMT_CHECK_LAUNCH("vq_ema_apply");
  // the codebook has changed: refresh its centred split for the next forward in the same call
  return vq_finish_prepare(codebook, k_bins, dim, pr, stats + (size_t)k_bins * dim, k_elem, threshold, metrics, stream);
}
