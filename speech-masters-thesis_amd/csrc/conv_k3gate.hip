// K3 of ALL FOUR branches of a GatedHiFiBlock plus the tanh * softmax gate in one pass (bf16, width 64; reference
// models/vqvae/resnet.py:224-237):
//
//   z_d = W3_d u2_d + b3_d + (W1_d x + b1_d)        d = 0..3     (the K1 residual h1_d is recomputed from x: K1 is linear)
//   g   = sum_d tanh(t_d) * softmax_d(s_d),          z_d = [t_d | s_d]  (64 + 64 channels)
//
// Unfused, each branch's K3 wrote z_d (4 launches) and a fifth kernel read all of z back (1 KiB per row) only to emit the
// 128-byte g.  Here a persistent workgroup streams 128-row tiles: for each tile the four u2_d slices (32 KiB each) pass
// through an LDS-DMA double buffer one branch after the other, x (16 KiB) once; z_d is stored from registers (backward
// needs it) and the bf16-rounded (t_d, s_d) pairs stay in registers until the fourth branch, where the gate is formed with
// exactly the arithmetic of gate_mix_fwd_kernel -- g is bit-identical to the unfused path.  Per row: 1 KiB + 128 B read,
// 1 KiB + 128 B written (was 3.7 KB moved): HBM-bound.
//
// Output channels are assigned to accumulator rows so that a lane holds t_c and s_c of the SAME channel c: wave (wm, wn)
// computes rows 64 wm.. and channels {16 wn + i} (t) and {64 + 16 wn + i} (s), i < 16; A-row r < 16 is t-channel
// 16 wn + r, A-row r >= 16 is s-channel 64 + 16 wn + (r - 16).  The weights of a branch (48 registers per lane) are
// re-read from L2 for every (tile, branch) step; the loads are issued right after the step's MFMAs, so they travel while
// the epilogue stores z.
//
// Round 3: the step loop owns its vector-memory waits.  The first version let the compiler place them, and with LDS-DMA in
// flight hipcc answers the first use of ANY ordinary load result (the weights, the biases) with s_waitcnt vmcnt(0) -- which
// also drains the slice prefetched for the NEXT step: every step waited for its own prefetch and the double buffer bought
// nothing (3.2 TB/s).  Now the weights come through loads the compiler does not track (untracked_load16), the biases sit in
// LDS, slices and stores go through range-checked V#s (always issued), and ONE counted wait per step -- vmcnt(stores of the
// last row group) -- lets the z stores drain under the next step while the prefetched slice and weights are known to be in.
#include <algorithm>

#include "conv_common.h"

#ifndef KG_ABL
#define KG_ABL 0      // timing experiments only (results invalid): 1 weights loaded once, 2 no gate math, 4 no z stores
#endif

namespace smt {

struct K3GateArgs {
  const __bf16* u2; const __bf16* x; const __bf16* w3; const __bf16* w1; const float* b3; const float* b1;
  __bf16* z; __bf16* g; const int* lens;
  long long bs_u2, bs_x, bs_z, bs_g;
  int ld_u2, ld_x, ld_z, ld_g;
  int B, T, tiles_per_batch;
};

constexpr int KG_ROWS = 128, KG_NT = 512, KG_U = KG_ROWS * 256, KG_X = KG_ROWS * 128, KG_BIAS = 4 * 128 * 4,
              KG_LDS = 2 * KG_U + 2 * KG_X + KG_BIAS;       // [u buf 0 | u buf 1 | x buf 0 | x buf 1 | b3 + b1 of the four branches]

__device__ __forceinline__ float kg_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float kg_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

__global__ __launch_bounds__(KG_NT) void conv_k3gate_kernel(K3GateArgs p, const __bf16* __restrict__ zero_page,
                                                            int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // [u buf 0 | u buf 1 | x buf 0 | x buf 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 31, hh = lane >> 5;

  const int ntiles = p.tiles_per_batch * p.B;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = wg * tiles_per_wg;
  const int tile_end = min(ntiles, tile_begin + tiles_per_wg);
  if (tile_begin >= tile_end) return;

  const int co = (r >> 4) * 64 + 16 * wn + (r & 15);       // this lane's weight row within a branch (see header)
  bf16x8 wf[8], w2f[4];
  float bcur[16];
  // lane offsets are 32-bit and branch offsets uniform, so every load is "scalar base + 32-bit lane offset": 12 + 4
  // offset registers in all (with per-lane 64-bit pointers the 48 weight-row addresses get hoisted and spilled)
  unsigned woff[8], w2off[4], boff[4];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) woff[kk] = (unsigned)co * 256u + (unsigned)(((2 * kk + hh) ^ (co & 15)) << 4);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) w2off[kk] = (unsigned)co * 128u + (unsigned)((2 * kk + hh) << 4);
#pragma unroll
  for (int q = 0; q < 4; ++q) boff[q] = (unsigned)(((q >> 1) * 64 + 16 * wn + 8 * (q & 1) + 4 * hh) * 4);
  auto load_w = [&](int d) {                                // untracked: the step's counted wait covers them
    const unsigned char* w3b = reinterpret_cast<const unsigned char*>(p.w3) + (size_t)d * (128 * 256);
    const unsigned char* w1b = reinterpret_cast<const unsigned char*>(p.w1) + (size_t)d * (128 * 128);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wf[kk] = __builtin_bit_cast(bf16x8, untracked_load16(w3b, woff[kk]));
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) w2f[kk] = __builtin_bit_cast(bf16x8, untracked_load16(w1b, w2off[kk]));
  };
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * KG_U + 2 * KG_X);
  if (tid < 512) bias_lds[tid] = p.b3[tid] + p.b1[tid];      // [branch][channel]
  auto load_b = [&](int d, float* bv) {
    // accumulator element 4 q + k = A-row 8 q + 4 hh + k = channel (q >> 1) * 64 + 16 wn + 8 (q & 1) + 4 hh + k
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(bias_lds + d * 128) + boff[q]);
      bv[4 * q + 0] = a.x; bv[4 * q + 1] = a.y; bv[4 * q + 2] = a.z; bv[4 * q + 3] = a.w;
    }
  };
  // slices through V#s: rows >= T (u2) / >= len (x) are out of range and read as zero
  const unsigned pitch_u = (unsigned)p.ld_u2 * 2u, pitch_x = (unsigned)p.ld_x * 2u;
  const unsigned pitch_z = (unsigned)p.ld_z * 2u, pitch_g = (unsigned)p.ld_g * 2u;
  const unsigned uoff0 = (unsigned)(4 * wave + (lane >> 4)) * pitch_u + (unsigned)(((lane & 15) ^ ((4 * wave + (lane >> 4)) & 15)) << 4);
  const unsigned xoff0 = (unsigned)(8 * wave + (lane >> 3)) * pitch_x + (unsigned)(((lane & 7) ^ (((8 * wave + (lane >> 3)) >> 1) & 7)) << 4);
  // tile -> (batch item, first row), as SCALARS: the V#s below must live in SGPRs (a divergent-looking descriptor costs a
  // waterfall loop per DMA instruction, and a vector load of lens[b] would be a tracked load again)
  auto decode = [&](int tile, int& b, int& t0) {
    b = __builtin_amdgcn_readfirstlane(tile / p.tiles_per_batch);
    t0 = __builtin_amdgcn_readfirstlane((tile - b * p.tiles_per_batch) * KG_ROWS);
  };
  auto stage_u = [&](int tile, int d, int buf) {
    int b, t0;
    decode(tile, b, t0);
    const UntrackedRsrc ru = untracked_rsrc(p.u2, ((long long)b * p.bs_u2 + d * 128) * 2, (unsigned)p.T * pitch_u);
    unsigned char* base = smem + (size_t)buf * KG_U + wave * 1024;
    unsigned vo = uoff0 + (unsigned)t0 * pitch_u;
#pragma unroll
    for (int q = 0; q < (KG_ROWS / 4) / (KG_NT / 64); ++q) {       // 4 rows x 16 chunks per wave-instruction; group wave + 8 q:
      untracked_dma16(ru, vo, base + q * (KG_NT / 64) * 1024);           // rows 4 (wave + 8 q) + .., so (row & 15) does not depend on q
      vo += 32u * pitch_u;
    }
  };
  auto stage_x = [&](int tile, int buf) {
    int b, t0;
    decode(tile, b, t0);
    const int len = p.lens ? min(scalar_load_i32(p.lens + b), p.T) : p.T;
    const UntrackedRsrc rx = untracked_rsrc(p.x, (long long)b * p.bs_x * 2, (unsigned)len * pitch_x);
    unsigned char* base = smem + 2 * KG_U + (size_t)buf * KG_X + wave * 1024;
    unsigned vo = xoff0 + (unsigned)t0 * pitch_x;
#pragma unroll
    for (int q = 0; q < (KG_ROWS / 8) / (KG_NT / 64); ++q) {       // 8 rows x 8 chunks per wave-instruction; 128-byte rows: two
      untracked_dma16(rx, vo, base + q * (KG_NT / 64) * 1024);           // rows share a 256-byte bank window, chunk c of row n at c ^ ((n >> 1) & 7)
      vo += 64u * pitch_x;
    }
  };

  stage_u(tile_begin, 0, 0);
  stage_x(tile_begin, 0);
  load_w(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the first slice, x tile and weights; later steps wait at their END
  int ubuf = 0;
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int xbuf = (tile - tile_begin) & 1;
    int b, t0;
    decode(tile, b, t0);
    unsigned hist[3][2][8];                                  // bf16 pairs of (t | s) of branches 0..2 per row group
    const __amdgpu_buffer_rsrc_t rg = ws_rsrc(p.g, (long long)b * p.bs_g * 2, (unsigned)p.T * pitch_g);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      // every wave's slice of this step landed (its own counted wait at the end of the previous step) and every wave is
      // done reading the buffers the prefetch below overwrites
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const bool more = d < 3 || tile + 1 < tile_end;
      if (more) {
        if (d < 3) stage_u(tile, d + 1, ubuf ^ 1);
        else { stage_u(tile + 1, 0, ubuf ^ 1); stage_x(tile + 1, xbuf ^ 1); }
      }
      const unsigned char* ut = smem + (size_t)ubuf * KG_U;
      const unsigned char* xt = smem + 2 * KG_U + (size_t)xbuf * KG_X;
      load_b(d, bcur);
      const __amdgpu_buffer_rsrc_t rz = ws_rsrc(p.z, ((long long)b * p.bs_z + d * 128) * 2, (unsigned)p.T * pitch_z);
#pragma unroll
      for (int i = 0; i < 2; ++i) {                           // the two 32-row groups one after the other (registers)
        const int row = wm * 64 + 32 * i + r;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const bf16x8 bv = *reinterpret_cast<const bf16x8*>(ut + row * 256 + (((2 * kk + hh) ^ (row & 15)) << 4));
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk], bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const bf16x8 bv = *reinterpret_cast<const bf16x8*>(xt + row * 128 + (((2 * kk + hh) ^ ((row >> 1) & 7)) << 4));
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[kk], bv, acc, 0, 0, 0);
        }
        if (i == 1 && more && !(KG_ABL & 1)) {
          __builtin_amdgcn_sched_barrier(0);                  // after the MFMAs that read the weight registers ...
          load_w((d + 1) & 3);                                // ... fetch the next branch's into them (12 untracked loads)
          __builtin_amdgcn_sched_barrier(0);
        }
        const int t = t0 + row;
        unsigned yp[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          yp[2 * q] = pack_bf16x2(acc[4 * q] + bcur[4 * q], acc[4 * q + 1] + bcur[4 * q + 1]);
          yp[2 * q + 1] = pack_bf16x2(acc[4 * q + 2] + bcur[4 * q + 2], acc[4 * q + 3] + bcur[4 * q + 3]);
        }
        unsigned last[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (d < 3) hist[d][i][e] = yp[e];
          last[e] = yp[e];
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int dd = 0; dd < 2; ++dd) {
            auto sw = __builtin_amdgcn_permlane32_swap(yp[4 * h2 + dd], yp[4 * h2 + 2 + dd], false, false);
            yp[4 * h2 + dd] = sw[0]; yp[4 * h2 + 2 + dd] = sw[1];
          }
        if (!(KG_ABL & 4)) {                                  // rows >= T are out of the V#'s range: dropped, but ISSUED
          const unsigned vo = (unsigned)t * pitch_z + (unsigned)(16 * wn + 8 * hh) * 2u;   // 8 consecutive t channels, then the s ones
          __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[0], (int)yp[1], (int)yp[2], (int)yp[3]}, rz, (int)vo, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[4], (int)yp[5], (int)yp[6], (int)yp[7]}, rz, (int)(vo + 128u), 0, 0);
        }
        if (d == 3 && !(KG_ABL & 2)) {
          // the gate, with the arithmetic of gate_mix_fwd_kernel on the bf16-rounded z (what backward will read)
          unsigned gq[4];
#pragma unroll
          for (int q = 0; q < 2; ++q)                        // channel group 16 wn + 8 q + 4 hh + k
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
              float o[2];
#pragma unroll
              for (int half = 0; half < 2; ++half) {
                float tv[4], sv[4];
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                  const unsigned tw = dd < 3 ? hist[dd < 3 ? dd : 0][i][2 * q + kp] : last[2 * q + kp];
                  const unsigned sw = dd < 3 ? hist[dd < 3 ? dd : 0][i][2 * (q + 2) + kp] : last[2 * (q + 2) + kp];
                  tv[dd] = half ? kg_hi(tw) : kg_lo(tw);
                  sv[dd] = half ? kg_hi(sw) : kg_lo(sw);
                }
                float m = -INFINITY;
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) m = fmaxf(m, sv[dd]);
                float den = 0.f, num = 0.f;
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                  float ex = __expf(sv[dd] - m);
                  den += ex;
                  num += ex * gate_tanh(tv[dd]);
                }
                o[half] = num * __builtin_amdgcn_rcpf(den);           // bf16 result: one reciprocal (1 ulp), as gate_mix_fwd_kernel<bf16> does
                __builtin_amdgcn_sched_barrier(0);            // one element at a time: the temporaries of several
              }                                               // elements in flight at once would spill the history
              gq[2 * q + kp] = pack_bf16x2(o[0], o[1]);
            }
#pragma unroll
          for (int dd = 0; dd < 2; ++dd) {
            auto sw = __builtin_amdgcn_permlane32_swap(gq[dd], gq[2 + dd], false, false);
            gq[dd] = sw[0]; gq[2 + dd] = sw[1];
          }
          __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)gq[0], (int)gq[1], (int)gq[2], (int)gq[3]}, rg,
                                                 (int)((unsigned)t * pitch_g + (unsigned)(16 * wn + 8 * hh) * 2u), 0, 0);
        }
      }
      // The slice and weights of the next step were issued BEFORE the last row group's stores (2 of z, + 1 of g in the gate
      // step): everything older than those stores is done after this wait; the stores drain under the next step.
      __builtin_amdgcn_sched_barrier(0);
      if (KG_ABL & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (d == 3 && !(KG_ABL & 2)) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      ubuf ^= 1;
    }
  }
}

}  // namespace smt

using namespace smt;

extern "C" int smt_conv_k3gate_fwd(const void* u2, int64_t bs_u2, int ld_u2, const void* x, int64_t bs_x, int ld_x,
                                   const void* w3_packed, const void* w1_packed, const float* b3, const float* b1, void* z,
                                   int64_t bs_z, int ld_z, void* g, int64_t bs_g, int ld_g, const int* lens, int batch,
                                   int t, const void* zero_page, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(u2 && x && w3_packed && w1_packed && b3 && b1 && z && g, "smt_conv_k3gate_fwd: null pointer");
  SMT_CHECK_ARG(ld_u2 % 8 == 0 && ld_x % 8 == 0 && ld_z % 8 == 0 && ld_g % 8 == 0 && ld_u2 >= 512 && ld_z >= 512 &&
                    ld_x >= 64 && ld_g >= 64,
                "smt_conv_k3gate_fwd: pitches must keep 16-byte alignment and hold 512 / 64 channels");
  if (batch <= 0 || t <= 0) return 0;
  K3GateArgs p;
  p.u2 = (const __bf16*)u2; p.x = (const __bf16*)x; p.w3 = (const __bf16*)w3_packed; p.w1 = (const __bf16*)w1_packed;
  p.b3 = b3; p.b1 = b1; p.z = (__bf16*)z; p.g = (__bf16*)g; p.lens = lens;
  p.bs_u2 = bs_u2; p.bs_x = bs_x; p.bs_z = bs_z; p.bs_g = bs_g;
  p.ld_u2 = ld_u2; p.ld_x = ld_x; p.ld_z = ld_z; p.ld_g = ld_g;
  p.B = batch; p.T = t; p.tiles_per_batch = (t + KG_ROWS - 1) / KG_ROWS;
  const int ntiles = p.tiles_per_batch * batch;
  int nwg = std::min(256, std::max(8, ntiles));              // one workgroup per CU (96 KiB of LDS)
  nwg = (nwg + 7) / 8 * 8;
  const int tpw = (ntiles + nwg - 1) / nwg;
  (void)hipFuncSetAttribute((const void*)conv_k3gate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, KG_LDS);
  conv_k3gate_kernel<<<nwg, KG_NT, KG_LDS, stream>>>(p, (const __bf16*)zero_page, tpw);
  SMT_CHECK_LAUNCH("conv_k3gate");
  return 0;
}
