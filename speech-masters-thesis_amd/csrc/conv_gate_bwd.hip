// Fused backward of the 64 -> 64 1x1 "gate" convolution that closes a GatedHiFiBlock (reference
// models/vqvae/resnet.py:238-241: `self.gate_conv(g)` followed by the block residual):
//
//   dg[t, ci]   = keep(t) * sum_co dout[t, co] * W[co][ci]                      (keep = t < lens[b])
//   dW[co][ci]  = sum_t dout[t, co] * g[t, ci],   db[co] = sum_t dout[t, co]    (g rows >= lens read as 0)
//
// A 1x1 layer at width 64 moves 384 B per row for 16 KFLOP: it is HBM-bound, and the data gradient and the
// weight gradient both read dout.  One persistent kernel streams 128-row tiles of dout and g (16 KiB each)
// through an LDS-DMA double buffer once:
//   * data gradient : eight 32 x 32 tiles of dg, one per wave, transposed MFMA (A = W^T fragments in registers,
//                     B = dout rows), row mask, v_permlane32_swap pairing, 16-byte stores from registers;
//   * weight gradient: four 32 x 32 tiles of dW, each accumulated by two waves over the two halves of a tile's rows
//                     (each half leaves its own partial slab), fragments of dout^T and g via ds_read_b64_tr_b16; the
//                     bias gradient is one more MFMA against a constant-one operand.
// Rows are 128 B: chunk c of row r sits at c ^ (((r >> 1) & 3) << 1) (transposed fragments conflict-free, row
// fragments 2-way).  conv_wgrad_reduce_kernel sums the slabs in fixed order.
#include <algorithm>

#include "conv_common.h"

namespace smt {

struct GateBwdArgs {
  const void* dy; const void* g; const void* w; void* dx; float* slab;
  const int* lens;
  long long dy_bs, g_bs, dx_bs;
  int lddy, ldg, lddx;
  int B, T, tiles_per_batch, tiles_per_wg, with_bias;
};

constexpr int GB_ROWS = 128, GB_C = 64, GB_NT = 512, GB_TILE = GB_ROWS * GB_C * 2, GB_STAGE = 2 * GB_TILE;

__device__ __forceinline__ int gb_swz(int row) { return ((row >> 1) & 3) << 1; }

__device__ __forceinline__ bf16x8 gb_tr2(const unsigned char* pa, const unsigned char* pb) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pa);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pb);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(GB_NT) void conv_gate_bwd_kernel(GateBwdArgs p, const __bf16* __restrict__ zero_page) {
  typedef __bf16 T;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // 2 x [dout tile | g tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;

  const int ntiles = p.tiles_per_batch * p.B;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = min(ntiles, wg * p.tiles_per_wg);
  const int tile_end = min(ntiles, tile_begin + p.tiles_per_wg);

  // data-gradient tile of this wave: rows 32 ri.., input channels 32 dc..; W^T packed [ci][co], plain
  const int ri = wave >> 1, dc = wave & 1;
  bf16x8 wfrag[GB_C / 16];
  {
    const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.w) + (size_t)(32 * dc + r) * (GB_C * 2);
#pragma unroll
    for (int kk = 0; kk < GB_C / 16; ++kk) wfrag[kk] = *reinterpret_cast<const bf16x8*>(wrow + ((2 * kk + hh) << 4));
  }
  // weight-gradient tile of this wave: output channels 32 wa.., input channels 32 wc.., k-steps 4 kh .. 4 kh + 3
  const int wa = (wave >> 1) & 1, wc = wave & 1, kh = wave >> 2;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  f32x16 accw, accb;
#pragma unroll
  for (int e = 0; e < 16; ++e) { accw[e] = 0.f; accb[e] = 0.f; }

  // Round 3: the tile loop owns its vector-memory waits (conv_common.h, conv_k3gate.hip): untracked LDS-DMA through V#s,
  // scalar lens loads, dx stores through a V# (always issued), one counted wait per tile.
  auto decode = [&](int tile, int& b, int& t0) {
    b = __builtin_amdgcn_readfirstlane(tile / p.tiles_per_batch);
    t0 = __builtin_amdgcn_readfirstlane((tile - b * p.tiles_per_batch) * GB_ROWS);
  };
  const unsigned pitch_dy = (unsigned)p.lddy * 2u, pitch_g = (unsigned)p.ldg * 2u, pitch_dx = (unsigned)p.lddx * 2u;
  auto stage = [&](int tile, int buf) {
    int b, t0;
    decode(tile, b, t0);
    const int len = p.lens ? min(scalar_load_i32(p.lens + b), p.T) : p.T;
    const UntrackedRsrc rdy = untracked_rsrc(p.dy, (long long)b * p.dy_bs * 2, (unsigned)p.T * pitch_dy);
    const UntrackedRsrc rg = untracked_rsrc(p.g, (long long)b * p.g_bs * 2, (unsigned)len * pitch_g);
    unsigned char* base = smem + (size_t)buf * GB_STAGE;
#pragma unroll
    for (int q = 0; q < (GB_ROWS / 8) / (GB_NT / 64); ++q) {      // 8 rows x 8 chunks per instruction; rows >= T / >= len read as zero
      const int i8 = wave + (GB_NT / 64) * q;
      const int row = 8 * i8 + (lane >> 3), pos = lane & 7;
      const unsigned ch = (unsigned)((pos ^ gb_swz(row)) << 4);
      untracked_dma16(rdy, (unsigned)(t0 + row) * pitch_dy + ch, base + i8 * 1024);
      untracked_dma16(rg, (unsigned)(t0 + row) * pitch_g + ch, base + GB_TILE + i8 * 1024);
    }
  };

  // transposed-fragment offsets (the k-step advances rows by 16, which keeps the swizzle)
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3, thh = tg >> 1;
  const int ra = 8 * thh + tq, rb = ra + 4;
  const int col_a = 32 * wa + 16 * (tg & 1) + 4 * tp, col_b = 32 * wc + 16 * (tg & 1) + 4 * tp;
  const int offa0 = ra * 128 + (((col_a >> 3) ^ gb_swz(ra)) << 4) + (col_a & 7) * 2;
  const int offa1 = rb * 128 + (((col_a >> 3) ^ gb_swz(rb)) << 4) + (col_a & 7) * 2;
  const int offb0 = GB_TILE + ra * 128 + (((col_b >> 3) ^ gb_swz(ra)) << 4) + (col_b & 7) * 2;
  const int offb1 = GB_TILE + rb * 128 + (((col_b >> 3) ^ gb_swz(rb)) << 4) + (col_b & 7) * 2;

  if (tile_begin < tile_end) stage(tile_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first tile (and the weights); later tiles: counted wait at the END
#pragma unroll
  for (int kk = 0; kk < GB_C / 16; ++kk) asm volatile("" : "+v"(wfrag[kk]));
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    int b, t0;
    decode(tile, b, t0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave's part of this tile landed; the other buffer is free again
    if (tile + 1 < tile_end) stage(tile + 1, buf ^ 1);
    const unsigned char* base = smem + (size_t)buf * GB_STAGE;

    // ---- data gradient, transposed: D^T[ci][row] = sum_co Wt[ci][co] * dout[row][co]
    {
      const int row = 32 * ri + r;
      const int t = t0 + row;
      const int len = p.lens ? scalar_load_i32(p.lens + b) : 0x7fffffff;
      const float keep_row = (t >= len) ? 0.f : 1.f;
      f32x16 accd;
#pragma unroll
      for (int e = 0; e < 16; ++e) accd[e] = 0.f;
      const int swz_r = gb_swz(row);
#pragma unroll
      for (int kk = 0; kk < GB_C / 16; ++kk) {
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(base + row * 128 + (((2 * kk + hh) ^ swz_r) << 4));
        accd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[kk], bv, accd, 0, 0, 0);
      }
      unsigned yp[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (float)(T)accd[4 * g + k] * keep_row;
        yp[2 * g] = pack_bf16x2(o[0], o[1]);
        yp[2 * g + 1] = pack_bf16x2(o[2], o[3]);
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          auto sw = __builtin_amdgcn_permlane32_swap(yp[4 * h2 + d], yp[4 * h2 + 2 + d], false, false);
          yp[4 * h2 + d] = sw[0]; yp[4 * h2 + 2 + d] = sw[1];
        }
      {                                                 // rows >= T: out of range, dropped -- but ISSUED
        const __amdgpu_buffer_rsrc_t rdx = ws_rsrc(p.dx, (long long)b * p.dx_bs * 2, (unsigned)p.T * pitch_dx);
        const unsigned vo = (unsigned)t * pitch_dx + (unsigned)(32 * dc + 8 * hh) * 2u;
        __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[0], (int)yp[1], (int)yp[2], (int)yp[3]}, rdx, (int)vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[4], (int)yp[5], (int)yp[6], (int)yp[7]}, rdx, (int)(vo + 32u), 0, 0);
      }
    }

    // ---- weight gradient: dW[co][ci] += sum_rows dout[row][co] * g[row][ci] over this wave's half of the rows
#pragma unroll
    for (int ks = 0; ks < GB_ROWS / 32; ++ks) {
      const int ko = (4 * kh + ks) * 16 * 128;
      const bf16x8 afr = gb_tr2(base + offa0 + ko, base + offa1 + ko);
      const bf16x8 bfr = gb_tr2(base + offb0 + ko, base + offb1 + ko);
      accw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, accw, 0, 0, 0);
      if (p.with_bias && wc == 0) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, ones, accb, 0, 0, 0);
    }
    // the next tile's DMA was issued before this tile's 2 dx stores
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }

  // ---- partial dW / db -> slab[chunk = 2 wg + kh][plane][64 co][64 ci]
  float* out = p.slab + ((size_t)wg * 2 + kh) * 2 * 64 * GB_C;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = 32 * wa + (e & 3) + 8 * (e >> 2) + 4 * hh;
    out[(size_t)row * GB_C + 32 * wc + r] = accw[e];
    if (wc == 0) out[(size_t)(64 + row) * GB_C + r] = accb[e];
  }
}

static int gate_bwd_nwg(int batch, int t) {
  const long long ntiles = (long long)((t + GB_ROWS - 1) / GB_ROWS) * batch;
  long long nwg = std::min<long long>(512, std::max<long long>(8, (ntiles + fused_min_tpw() - 1) / fused_min_tpw()));   // two workgroups per CU (64 KiB of LDS)
  return (int)((nwg + 7) / 8 * 8);
}

}  // namespace smt

using namespace smt;

extern "C" size_t smt_conv_gate_bwd_workspace_bytes(int batch, int t) {
  return (size_t)gate_bwd_nwg(batch, t) * 2 * 2 * 64 * GB_C * sizeof(float);
}

extern "C" int smt_conv_gate_bwd(const void* dy, int64_t bs_dy, int ld_dy, const void* g, int64_t bs_g, int ld_g,
                                 const void* w_packed_bwd, void* dx, int64_t bs_dx, int ld_dx, const int* lens,
                                 int batch, int t, const void* zero_page, float* dweight, int64_t stride_out,
                                 int64_t stride_in, float* dbias, void* workspace, size_t workspace_bytes,
                                 smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(dy && g && w_packed_bwd && dx && zero_page && dweight && workspace, "smt_conv_gate_bwd: null pointer");
  SMT_CHECK_ARG(ld_dy % 8 == 0 && ld_g % 8 == 0 && ld_dx % 8 == 0, "smt_conv_gate_bwd: row pitches must keep 16-byte alignment");
  SMT_CHECK_ARG(workspace_bytes >= smt_conv_gate_bwd_workspace_bytes(batch, t), "smt_conv_gate_bwd: workspace too small");
  const int nwg = gate_bwd_nwg(batch, t);
  if (batch > 0 && t > 0) {
    GateBwdArgs a;
    a.dy = dy; a.g = g; a.w = w_packed_bwd; a.dx = dx; a.slab = (float*)workspace; a.lens = lens;
    a.dy_bs = bs_dy; a.g_bs = bs_g; a.dx_bs = bs_dx;
    a.lddy = ld_dy; a.ldg = ld_g; a.lddx = ld_dx;
    a.B = batch; a.T = t;
    a.tiles_per_batch = (t + GB_ROWS - 1) / GB_ROWS;
    const long long ntiles = (long long)a.tiles_per_batch * batch;
    a.tiles_per_wg = (int)((ntiles + nwg - 1) / nwg);
    a.with_bias = dbias ? 1 : 0;
    (void)hipFuncSetAttribute((const void*)conv_gate_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    conv_gate_bwd_kernel<<<nwg, GB_NT, 2 * GB_STAGE, stream>>>(a, (const __bf16*)zero_page);
    SMT_CHECK_LAUNCH("conv_gate_bwd");
  }
  const int jmap[1] = {0};
  const int n_chunks = (batch > 0 && t > 0) ? 2 * nwg : 0;
  return launch_wgrad_reduce((const float*)workspace, dweight, dbias, n_chunks, 1, 1, 1, GB_C, GB_C, 64, stride_out,
                             stride_in, 0, jmap, stream);
}
