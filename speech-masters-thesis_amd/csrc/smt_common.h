// Shared helpers for the libsmt_hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/smt_hip.h"

namespace smt {

void set_error(const char* fmt, ...);

#define SMT_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      ::smt::set_error(__VA_ARGS__);        \
      return 1;                             \
    }                                       \
  } while (0)

#define SMT_CHECK_LAUNCH(name)                                                  \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      ::smt::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
      return 2;                                                                 \
    }                                                                           \
  } while (0)

__host__ __device__ static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace smt
