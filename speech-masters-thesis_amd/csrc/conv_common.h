// Pieces shared by the convolution kernels: element traits, 16-byte vectors, counter-based dropout.
#pragma once
#include "smt_common.h"

namespace smt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
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


}  // namespace smt
