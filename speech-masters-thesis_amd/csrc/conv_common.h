// Pieces shared by the convolution kernels: element traits, 16-byte vectors, counter-based dropout.
#pragma once
#include "smt_common.h"

namespace smt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// ---- counter-based dropout (spec: include/smt_hip.h "dropout") -------------------------------
__device__ __forceinline__ unsigned fmix32(unsigned h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
// keep-bit of linear element index i: 16 random bits per element, two elements per hash
__device__ __forceinline__ bool drop_keep(unsigned long long i, unsigned key, unsigned thresh16) {
  unsigned h = fmix32((unsigned)(i >> 1) * 0x9E3779B1u + key);
  unsigned bits = (h >> (16 * (unsigned)(i & 1))) & 0xFFFFu;
  return bits >= thresh16;
}

template <typename T> struct Tr;
template <> struct Tr<__bf16> {
  static constexpr int EPV = 8;      // elements per 16-byte vector
  static constexpr int CCH = 128;    // input channels staged per LDS A chunk
  static constexpr int KC = 128;     // K per weight chunk (one barrier per tap for 128 input channels)
  static constexpr int BM = 128;
};
template <> struct Tr<float> {
  static constexpr int EPV = 4;
  static constexpr int CCH = 64;
  static constexpr int KC = 64;
  static constexpr int BM = 64;
};

template <typename T, int EPV> struct Vec { T v[EPV]; } __attribute__((aligned(16)));

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

// tanh for the gate (resnet.py:233): 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp -- 5 instructions instead of the
// ~40 of tanhf; absolute error <= 2e-7 (the gate's output is O(1) and is stored in bf16 on the fast path), exact limits
// +-1 at +-inf.  Every gate kernel (forward, fused forward, backward) uses this one function, so they stay consistent.
__device__ __forceinline__ float gate_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);        // exp(2x) = 2^(2x log2 e)
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// LDS-DMA: one wave-instruction copies 64 x 16 B (per-lane global source) to 1 KiB of LDS at a wave-uniform base
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_dst_wave_base, 16, 0, 0);
}

// Fixed-order reduction of weight-gradient partial slabs (conv_wgrad.hip): slab[chunk][blk = co/64 * nblk_ci + ci/cib]
// [plane = tap | bias][64 co][cib ci] -> dw[co*so + ci*si + jmap[tap]*sj], db[co] (column 0 of the bias plane).
int launch_wgrad_reduce(const float* slab, float* dw, float* db, int n_chunks, int nblk_co, int nblk_ci, int taps,
                        int c_in, int c_out, int cib, long long so, long long si, long long sj, const int* jmap,
                        hipStream_t stream, int bias_cols = 1);   // bias_cols > 1: db = sum of columns 0, 32, .. of the bias plane


}  // namespace smt
