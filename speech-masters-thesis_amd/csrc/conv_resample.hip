// The k = 4 / stride 2 / padding 1 resampling convolutions of the encoder and decoder (reference models/vqvae/conv.py:61-78,
// 111-137: MaskedConv1d(.., stride_t * 2, stride_t, pad_t) and MaskedConvTranspose1d of the same geometry) at width 64, bf16.
//
// They move 0.4 - 1.5 GB per launch for a few GFLOP: HBM-bound.  On the generic implicit-GEMM kernel the 14 layers cost
// 3.8 ms per train step in forward + data gradient (6 % MFMA-busy, ~1.5 TB/s; the transposed conv ran as one launch per
// output phase and read its input twice).  Two persistent streaming kernels replace those launches, each with its whole
// weight block in registers, LDS-DMA double-buffered input tiles (chunk-swizzled, rows outside [0, len) from a zero page),
// transposed MFMA tiles (A = weights, B = input rows) and 16-byte stores straight from registers:
//
//   convt4s2  y[2m]   = W1 x[m] + W3 x[m-1] + b          x [B, Tin, 64]  ->  y [B, 2 Tin, COUT], COUT in {64, 128}
//             y[2m+1] = W2 x[m] + W0 x[m+1] + b          (both phases from ONE read of x; also the data gradient of conv4s2)
//   conv4s2   y[t]    = sum_j Wj x[2t + j - 1] + b        x [B, Tin, CIN], CIN in {64, 128}  ->  y [B, Tin / 2, 64]
//                                                         (also the data gradient of convt4s2)
// Weights: w[j][COUT][CIN] bf16 (smt_pack_weight layout, swizzle 0), j = kernel tap.  Rows of y at t >= lens_out[b] are
// written as zero (the data-gradient row mask); rows of x at t >= lens_in[b] read as zero (the forward row mask).
#include <algorithm>

#include "conv_common.h"

namespace smt {

struct RsArgs {
  const __bf16* x; const __bf16* w; const float* bias; __bf16* y; const int* lens_in; const int* lens_out;
  long long x_bs, y_bs;
  int ldx, ldy;
  int B, Tin, Tout, tiles_per_batch;
};

constexpr int RS_NT = 512;
typedef unsigned rs_u32x4 __attribute__((ext_vector_type(4)));

// pack the 16 accumulator values of a lane (+ bias, row mask) into two 16-byte pieces of 8 consecutive channels each
// (stored through a range-checked V#: rows beyond the output are dropped by the hardware, the instruction is always issued)
__device__ __forceinline__ void rs_store(const f32x16& acc, const float* bval, float keep, __amdgpu_buffer_rsrc_t ry, unsigned vo) {
  unsigned yp[8];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    yp[2 * g] = pack_bf16x2((acc[4 * g] + bval[4 * g]) * keep, (acc[4 * g + 1] + bval[4 * g + 1]) * keep);
    yp[2 * g + 1] = pack_bf16x2((acc[4 * g + 2] + bval[4 * g + 2]) * keep, (acc[4 * g + 3] + bval[4 * g + 3]) * keep);
  }
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      auto sw = __builtin_amdgcn_permlane32_swap(yp[4 * h2 + d], yp[4 * h2 + 2 + d], false, false);
      yp[4 * h2 + d] = sw[0]; yp[4 * h2 + 2 + d] = sw[1];
    }
  __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[0], (int)yp[1], (int)yp[2], (int)yp[3]}, ry, (int)vo, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[4], (int)yp[5], (int)yp[6], (int)yp[7]}, ry, (int)(vo + 32u), 0, 0);
}

// Round 3: both tile loops own their vector-memory waits (conv_common.h, conv_k3gate.hip): untracked LDS-DMA through a V#
// (rows before the item wrap to huge offsets, rows >= len are beyond it: both read as zero), scalar lens loads, one counted
// wait per tile that leaves the tile's stores in flight.
#define RS_MARK_LOADED(arr, n) _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) asm volatile("" : "+v"((arr)[i_]))

// ------------------------------------------------------------------------------------------ transposed, C_in = 64
constexpr int CT_TM = 128;                                   // input rows per tile (256 output rows)
constexpr int CT_ROWS = 136;                                 // staged rows: m0 - 1 .. m0 + 128, rounded up to 8-row DMA pieces
constexpr int CT_BUF = CT_ROWS * 128;

template <int COUT>
__global__ __launch_bounds__(RS_NT) void convt4s2_kernel(RsArgs p, const __bf16* __restrict__ zero_page, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // 2 x [136 rows x 128 B]
  constexpr int NCG = COUT / 32;                             // 32-channel groups
  constexpr int NRH = 8 / (2 * NCG);                         // row halves sharing the tile (1 for 128 channels, 2 for 64)
  constexpr int RGW = CT_TM / 32 / NRH;                      // 32-row groups per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int ph = wave & 1, cg = (wave >> 1) % NCG, rh = (wave >> 1) / NCG;
  const int rb = rh * (CT_TM / NRH);                         // first tile row of this wave

  const int ntiles = p.tiles_per_batch * p.B;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = wg * tiles_per_wg;
  const int tile_end = min(ntiles, tile_begin + tiles_per_wg);
  if (tile_begin >= tile_end) return;

  // phase 0: taps (3, offset -1), (1, offset 0); phase 1: taps (2, offset 0), (0, offset +1) -- in increasing input offset,
  // the accumulation order of the per-phase launches this kernel replaces (bit-identical results)
  const int tap_a = ph ? 2 : 3, tap_b = ph ? 0 : 1, off_a = ph ? 0 : -1, off_b = ph ? 1 : 0;
  bf16x8 wa[4], wb[4];
  {
    const int co = cg * 32 + r;
    const unsigned char* wra = reinterpret_cast<const unsigned char*>(p.w) + ((size_t)tap_a * COUT + co) * 128;
    const unsigned char* wrb = reinterpret_cast<const unsigned char*>(p.w) + ((size_t)tap_b * COUT + co) * 128;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      wa[kk] = *reinterpret_cast<const bf16x8*>(wra + ((2 * kk + hh) << 4));
      wb[kk] = *reinterpret_cast<const bf16x8*>(wrb + ((2 * kk + hh) << 4));
    }
  }
  float bval[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bval[e] = p.bias ? p.bias[cg * 32 + 4 * hh + 8 * (e >> 2) + (e & 3)] : 0.f;

  auto decode = [&](int tile, int& b, int& m0) {
    b = __builtin_amdgcn_readfirstlane(tile / p.tiles_per_batch);
    m0 = __builtin_amdgcn_readfirstlane((tile - b * p.tiles_per_batch) * CT_TM);
  };
  const unsigned pitch_x = (unsigned)p.ldx * 2u, pitch_y = (unsigned)p.ldy * 2u;
  auto stage = [&](int tile, int buf) {
    int b, m0;
    decode(tile, b, m0);
    const int len = p.lens_in ? min(scalar_load_i32(p.lens_in + b), p.Tin) : p.Tin;
    const UntrackedRsrc rx = untracked_rsrc(p.x, (long long)b * p.x_bs * 2, (unsigned)len * pitch_x);
    unsigned char* base = smem + (size_t)buf * CT_BUF;
#pragma unroll
    for (int q = 0; q < (CT_ROWS / 8 + RS_NT / 64 - 1) / (RS_NT / 64); ++q) {   // 8 rows x 8 chunks per wave-instruction (17 pieces:
      const int gi = wave + (RS_NT / 64) * q;                                   // the third round is wave 0's alone -- see the wait)
      if (gi < CT_ROWS / 8) {
        const int row = 8 * gi + (lane >> 3), pos = lane & 7;
        untracked_dma16(rx, (unsigned)(m0 - 1 + row) * pitch_x + (unsigned)((pos ^ ((row >> 1) & 7)) << 4), base + gi * 1024);
      }
    }
  };

  stage(tile_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the first tile, weights, biases; later tiles: counted wait at the END
  RS_MARK_LOADED(wa, 4); RS_MARK_LOADED(wb, 4); RS_MARK_LOADED(bval, 16);
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    int b, m0;
    decode(tile, b, m0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // every wave's part of this tile landed; the other buffer is free again
    if (tile + 1 < tile_end) stage(tile + 1, buf ^ 1);
    const unsigned char* xt = smem + (size_t)buf * CT_BUF;
    const __amdgpu_buffer_rsrc_t ry = ws_rsrc(p.y, (long long)b * p.y_bs * 2, (unsigned)p.Tout * pitch_y);
    const int len_out = p.lens_out ? scalar_load_i32(p.lens_out + b) : 0x7fffffff;
#pragma unroll
    for (int i = 0; i < RGW; ++i) {
      const int lm = rb + 32 * i + r;                        // tile row m - m0 of this lane
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      const int la = lm + 1 + off_a, lb = lm + 1 + off_b;     // staged rows of the two taps
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 va = *reinterpret_cast<const bf16x8*>(xt + la * 128 + (((2 * kk + hh) ^ ((la >> 1) & 7)) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[kk], va, acc, 0, 0, 0);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 vb = *reinterpret_cast<const bf16x8*>(xt + lb * 128 + (((2 * kk + hh) ^ ((lb >> 1) & 7)) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[kk], vb, acc, 0, 0, 0);
      }
      const int m = m0 + lm, t = 2 * m + ph;
      rs_store(acc, bval, t < len_out ? 1.f : 0.f, ry, (unsigned)t * pitch_y + (unsigned)(cg * 32 + 8 * hh) * 2u);
    }
    // the next tile's DMA is older than this tile's 2 RGW stores
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * RGW) : "memory");
  }
}

// ------------------------------------------------------------------------------------------ strided, C_out = 64
constexpr int CS_TO = 128;                                   // output rows per tile (input rows 2 t0 - 1 .. 2 t0 + 256)

template <int CIN>
__global__ __launch_bounds__(RS_NT) void conv4s2_kernel(RsArgs p, const __bf16* __restrict__ zero_page, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int ROWB = CIN * 2;                              // bytes per staged row
  constexpr int RPI = 1024 / ROWB;                           // rows per DMA wave-instruction (8 or 4)
  constexpr int CPR = ROWB / 16;                             // 16-byte chunks per row (8 or 16)
  constexpr int ROWS = (2 * CS_TO + 2 + RPI - 1) / RPI * RPI;
  constexpr int BUF = ROWS * ROWB;
  constexpr int KS = CIN / 16;                               // k-steps per tap
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int cg = wave & 1, rg = wave >> 1;                   // 32 output channels, 32 output rows

  const int ntiles = p.tiles_per_batch * p.B;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = wg * tiles_per_wg;
  const int tile_end = min(ntiles, tile_begin + tiles_per_wg);
  if (tile_begin >= tile_end) return;

  bf16x8 wf[4][KS];
  {
    const int co = cg * 32 + r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.w) + ((size_t)j * 64 + co) * ROWB;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) wf[j][kk] = *reinterpret_cast<const bf16x8*>(wrow + ((2 * kk + hh) << 4));
    }
  }
  float bval[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bval[e] = p.bias ? p.bias[cg * 32 + 4 * hh + 8 * (e >> 2) + (e & 3)] : 0.f;

  // chunk c of staged row l sits at c ^ swz(l): consecutive output rows read staged rows 2 apart
  auto swz = [](int l) { return (l >> 1) & (CPR - 1); };
  auto decode = [&](int tile, int& b, int& t0) {
    b = __builtin_amdgcn_readfirstlane(tile / p.tiles_per_batch);
    t0 = __builtin_amdgcn_readfirstlane((tile - b * p.tiles_per_batch) * CS_TO);
  };
  const unsigned pitch_x = (unsigned)p.ldx * 2u, pitch_y = (unsigned)p.ldy * 2u;
  auto stage = [&](int tile, int buf) {
    int b, t0;
    decode(tile, b, t0);
    const int len = p.lens_in ? min(scalar_load_i32(p.lens_in + b), p.Tin) : p.Tin;
    const UntrackedRsrc rx = untracked_rsrc(p.x, (long long)b * p.x_bs * 2, (unsigned)len * pitch_x);
    unsigned char* base = smem + (size_t)buf * BUF;
#pragma unroll
    for (int q = 0; q < (ROWS / RPI + RS_NT / 64 - 1) / (RS_NT / 64); ++q) {
      const int gi = wave + (RS_NT / 64) * q;
      if (gi < ROWS / RPI) {
        const int row = RPI * gi + lane / CPR, pos = lane % CPR;
        untracked_dma16(rx, (unsigned)(2 * t0 - 1 + row) * pitch_x + (unsigned)((pos ^ swz(row)) << 4), base + gi * 1024);
      }
    }
  };

  stage(tile_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the first tile, weights, biases; later tiles: counted wait at the END
#pragma unroll
  for (int j = 0; j < 4; ++j) { RS_MARK_LOADED(wf[j], KS); }
  RS_MARK_LOADED(bval, 16);
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    int b, t0;
    decode(tile, b, t0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + 1 < tile_end) stage(tile + 1, buf ^ 1);
    const unsigned char* xt = smem + (size_t)buf * BUF;
    const int lt = 32 * rg + r;                              // output row of this lane inside the tile
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int l = 2 * lt + j;                              // staged row of input row 2 t + j - 1
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(xt + l * ROWB + (((2 * kk + hh) ^ swz(l)) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][kk], v, acc, 0, 0, 0);
      }
    }
    const int t = t0 + lt;
    const int len_out = p.lens_out ? scalar_load_i32(p.lens_out + b) : 0x7fffffff;
    const __amdgpu_buffer_rsrc_t ry = ws_rsrc(p.y, (long long)b * p.y_bs * 2, (unsigned)p.Tout * pitch_y);
    rs_store(acc, bval, t < len_out ? 1.f : 0.f, ry, (unsigned)t * pitch_y + (unsigned)(cg * 32 + 8 * hh) * 2u);
    // the next tile's DMA is older than this tile's 2 stores
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
}

static int rs_grid(int ntiles, int* tpw) {
  int nwg = std::min(256, std::max(8, ntiles));
  nwg = (nwg + 7) / 8 * 8;
  *tpw = (ntiles + nwg - 1) / nwg;
  return nwg;
}

}  // namespace smt

using namespace smt;

extern "C" int smt_convt4s2(const void* x, int64_t bs_x, int ld_x, const void* w_packed, const float* bias, void* y,
                            int64_t bs_y, int ld_y, const int* lens_in, const int* lens_out, int batch, int t_in, int c_out,
                            const void* zero_page, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(x && w_packed && y && zero_page, "smt_convt4s2: null pointer");
  SMT_CHECK_ARG(c_out == 64 || c_out == 128, "smt_convt4s2: c_out must be 64 or 128 (got %d)", c_out);
  SMT_CHECK_ARG(ld_x % 8 == 0 && ld_y % 8 == 0 && ld_x >= 64 && ld_y >= c_out, "smt_convt4s2: bad pitches");
  if (batch <= 0 || t_in <= 0) return 0;
  RsArgs p;
  p.x = (const __bf16*)x; p.w = (const __bf16*)w_packed; p.bias = bias; p.y = (__bf16*)y; p.lens_in = lens_in; p.lens_out = lens_out;
  p.x_bs = bs_x; p.y_bs = bs_y; p.ldx = ld_x; p.ldy = ld_y;
  p.B = batch; p.Tin = t_in; p.Tout = 2 * t_in; p.tiles_per_batch = (t_in + CT_TM - 1) / CT_TM;
  int tpw;
  const int nwg = rs_grid(p.tiles_per_batch * batch, &tpw);
  if (c_out == 128) {
    (void)hipFuncSetAttribute((const void*)convt4s2_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CT_BUF);
    convt4s2_kernel<128><<<nwg, RS_NT, 2 * CT_BUF, stream>>>(p, (const __bf16*)zero_page, tpw);
  } else {
    (void)hipFuncSetAttribute((const void*)convt4s2_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CT_BUF);
    convt4s2_kernel<64><<<nwg, RS_NT, 2 * CT_BUF, stream>>>(p, (const __bf16*)zero_page, tpw);
  }
  SMT_CHECK_LAUNCH("convt4s2");
  return 0;
}

extern "C" int smt_conv4s2(const void* x, int64_t bs_x, int ld_x, const void* w_packed, const float* bias, void* y,
                           int64_t bs_y, int ld_y, const int* lens_in, const int* lens_out, int batch, int t_in, int c_in,
                           const void* zero_page, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(x && w_packed && y && zero_page, "smt_conv4s2: null pointer");
  SMT_CHECK_ARG(c_in == 64 || c_in == 128, "smt_conv4s2: c_in must be 64 or 128 (got %d)", c_in);
  SMT_CHECK_ARG(t_in % 2 == 0, "smt_conv4s2: t_in must be even (got %d)", t_in);
  SMT_CHECK_ARG(ld_x % 8 == 0 && ld_y % 8 == 0 && ld_x >= c_in && ld_y >= 64, "smt_conv4s2: bad pitches");
  if (batch <= 0 || t_in <= 0) return 0;
  RsArgs p;
  p.x = (const __bf16*)x; p.w = (const __bf16*)w_packed; p.bias = bias; p.y = (__bf16*)y; p.lens_in = lens_in; p.lens_out = lens_out;
  p.x_bs = bs_x; p.y_bs = bs_y; p.ldx = ld_x; p.ldy = ld_y;
  p.B = batch; p.Tin = t_in; p.Tout = t_in / 2; p.tiles_per_batch = (p.Tout + CS_TO - 1) / CS_TO;
  int tpw;
  const int nwg = rs_grid(p.tiles_per_batch * batch, &tpw);
  if (c_in == 128) {
    constexpr int BUF = (2 * CS_TO + 2 + 3) / 4 * 4 * 256;
    (void)hipFuncSetAttribute((const void*)conv4s2_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    conv4s2_kernel<128><<<nwg, RS_NT, 2 * BUF, stream>>>(p, (const __bf16*)zero_page, tpw);
  } else {
    constexpr int BUF = (2 * CS_TO + 2 + 7) / 8 * 8 * 128;
    (void)hipFuncSetAttribute((const void*)conv4s2_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    conv4s2_kernel<64><<<nwg, RS_NT, 2 * BUF, stream>>>(p, (const __bf16*)zero_page, tpw);
  }
  SMT_CHECK_LAUNCH("conv4s2");
  return 0;
}
