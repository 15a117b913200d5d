// Monotonic alignment search of GlowTTS (reference models/glow_tts/submodules.py:28-67, `maximum_path`): the reference
// copies the [b, t_x, t_y] log-likelihood matrix to the host, runs a numpy dynamic programme over t_y and copies the path
// back -- a device -> host -> device round trip on every train step (glow_tts.py:87-97).  Here one workgroup per batch item
// runs the same recurrence on the device:
//   v_j[x] = (x <= j) ? max(v_{j-1}[x], v_{j-1}[x-1]) + value[x, j] * mask[x, j] : max_neg_val,   direction[x, j] = v[x] >= v[x-1]
// with v in an LDS double buffer (one barrier per column j), the direction and mask bits in LDS bitmaps (one wave ballot
// per 64 rows) and the backtrack from LDS by one lane.  Every operation is the fp32 operation numpy performs, so the 0/1 path is bit-identical to the reference's.
#include <algorithm>

#include "smt_common.h"

namespace smt {

constexpr int MAS_NT = 256, MAS_SLAB = 16;

__global__ __launch_bounds__(MAS_NT) void maximum_path_kernel(const float* __restrict__ value, const float* __restrict__ mask,
                                                              int t_x, int t_y, float max_neg_val, float* __restrict__ path) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int words = (t_x + 63) / 64;                         // 64-bit bitmap words per column
  float* v0 = reinterpret_cast<float*>(smem);               // [2][t_x]
  unsigned long long* dirb = reinterpret_cast<unsigned long long*>(smem + 2 * (size_t)((t_x + 3) / 4 * 4) * sizeof(float));
  unsigned long long* maskb = dirb + (size_t)t_y * words;
  __shared__ int first_col_count;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const size_t base = (size_t)b * t_x * t_y;
  const float* val = value + base;
  const float* msk = mask + base;
  float* out = path + base;
  const int xs = (t_x + MAS_NT - 1) / MAS_NT;               // rows per thread: x = tid + MAS_NT * q (whole waves cover 64-row words)
  const int pitch = (t_x + 3) / 4 * 4;

  for (int x = tid; x < t_x; x += MAS_NT) v0[x] = 0.f;
  if (tid == 0) first_col_count = 0;
  for (size_t e = tid; e < (size_t)t_x * t_y; e += MAS_NT) out[e] = 0.f;
  __syncthreads();

  int cur = 0;
  for (int j0 = 0; j0 < t_y; j0 += MAS_SLAB) {
    const int jn = min(MAS_SLAB, t_y - j0);
    for (int jj = 0; jj < jn; ++jj) {
      const int j = j0 + jj;
      float* vc = v0 + cur * pitch;
      float* vn = v0 + (cur ^ 1) * pitch;
      for (int q = 0; q < xs; ++q) {
        const int x = tid + MAS_NT * q;
        const bool in = x < t_x;
        float m = 0.f, pv = 0.f;
        if (in) { m = msk[(size_t)x * t_y + j]; pv = val[(size_t)x * t_y + j] * m; }
        const float v1 = in ? vc[x] : 0.f;
        const float vprev = (in && x > 0) ? vc[x - 1] : max_neg_val;
        const bool keep = v1 >= vprev;
        const float vmax = keep ? v1 : vprev;
        if (in) vn[x] = (x <= j) ? vmax + pv : max_neg_val;
        const unsigned long long kb = __ballot(in && keep), mb = __ballot(in && m != 0.f);
        const int word = (tid >> 6) + (MAS_NT / 64) * q;
        if (lane == 0 && word < words) { dirb[(size_t)j * words + word] = kb; maskb[(size_t)j * words + word] = mb; }
        if (j == 0 && in && m != 0.f) atomicAdd(&first_col_count, 1);
      }
      cur ^= 1;
      __syncthreads();
    }
  }
  if (tid == 0) {
    int index = first_col_count - 1;                          // mask[:, :, 0].sum(1) - 1
    for (int j = t_y - 1; j >= 0; --j) {
      const int p = index < 0 ? index + t_x : index;          // numpy wraps a negative index ONCE ...
      if (p < 0 || p >= t_x) break;                           // ... and raises IndexError beyond that (NaN likelihoods, an empty first
                                                              // mask column with t_y > t_x): stop here, the rest of the path stays 0
      const unsigned long long mw = maskb[(size_t)j * words + (p >> 6)], dw = dirb[(size_t)j * words + (p >> 6)];
      const int mbit = (int)((mw >> (p & 63)) & 1), dbit = (int)((dw >> (p & 63)) & 1);
      if (mbit) out[(size_t)p * t_y + j] = 1.f;               // path * mask
      index = index + (mbit ? dbit : 1) - 1;                  // direction = where(mask, direction, 1)
    }
  }
}

static size_t mas_lds_bytes(int t_x, int t_y) {
  const size_t words = (t_x + 63) / 64;
  return 2 * (size_t)((t_x + 3) / 4 * 4) * sizeof(float) + 2 * (size_t)t_y * words * 8;
}

}  // namespace smt

using namespace smt;

extern "C" int smt_maximum_path(const float* value, const float* mask, int batch, int t_x, int t_y, float max_neg_val,
                                float* path, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0 || t_x <= 0 || t_y <= 0) return 0;
  SMT_CHECK_ARG(value && mask && path, "smt_maximum_path: null pointer");
  const size_t lds = mas_lds_bytes(t_x, t_y);
  SMT_CHECK_ARG(lds <= 160 * 1024 - 64, "smt_maximum_path: t_x=%d x t_y=%d needs %zu B of LDS for its bitmaps (limit 160 KiB)", t_x,
                t_y, lds);
  (void)hipFuncSetAttribute((const void*)maximum_path_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  maximum_path_kernel<<<batch, MAS_NT, lds, stream>>>(value, mask, t_x, t_y, max_neg_val, path);
  SMT_CHECK_LAUNCH("maximum_path");
  return 0;
}
