// Fused backward of a 1x1 convolution whose input went through relu+dropout (K3 of GatedHiFiBlock, reference
// models/vqvae/resnet.py:218-227: `Conv1d(2w, 2w, 1)` after `ReLU, Dropout`): ONE pass over the rows produces
//
//   dx[t, ci]  = (sum_co dy[t, co] * W[co][ci]) * scale * [u[t, ci] != 0]          (data gradient + activation mask)
//   dW[co][ci] = sum_t dy[t, co] * u[t, ci],   db[co] = sum_t dy[t, co]            (weight / bias gradient)
//
// where u = relu(dropout(h)) is the saved forward input.  Both results need exactly the same two operand
// rows (dy and u), and the layer is HBM-bound (768 B of traffic per row against 64 KFLOP), so computing them
// separately -- a data-gradient conv (dy, u in, dx out) and a weight-gradient pass (dy, u in again) -- moves
// twice the bytes.  Here a persistent workgroup streams 128-row tiles of dy and u through an LDS double
// buffer by LDS-DMA once:
//   * data gradient : transposed MFMA tile (A = W^T slice held in registers, B = dy rows), masked with u read
//                     back from the SAME LDS tile, paired to 16-byte pieces with v_permlane32_swap and stored
//                     from registers;
//   * weight gradient: dy^T and u fragments come from the same two LDS tiles through ds_read_b64_tr_b16 and
//                     accumulate into a 128 x 128 fp32 block that stays in registers for the whole run; the
//                     bias gradient is one more MFMA against a constant-one operand.
// One swizzle serves both access patterns: the 16-byte chunk c of row r sits at c ^ swz(r),
// swz(r) = ((r & 3) << 2) | ((r >> 2) & 3): 16 consecutive rows hit 16 different chunks (row fragments, b128) and
// 4 consecutive rows are 64 B apart (transposed fragments, 4 rows x 32 B per 16-lane group).
// Each workgroup leaves its partial dW / db in a slab; conv_wgrad_reduce_kernel sums the slabs in fixed order.
//
// Round 3: the tile loop owns its vector-memory waits (see conv_k3gate.hip): tiles and stores go through range-checked
// V#s (always issued), lens[b] is a scalar load, and ONE counted wait per tile -- vmcnt(4 stores) -- leaves the dx stores
// draining under the next tile while the prefetched tile is known to have landed.  Before, the vector load of lens[b]
// made the compiler wait vmcnt(0) right after the prefetch was issued: every tile waited for its own prefetch.
#include <algorithm>

#include "conv_common.h"

namespace smt {

struct Bwd1x1Args {
  const void* dy; const void* u; const void* w; void* dx; float* slab;
  const int* lens_out;
  long long dy_bs, u_bs, dx_bs;
  int lddy, ldu, lddx;
  int B, T, tiles_per_batch, tiles_per_wg, with_bias;
  float scale;
};

constexpr int FB_ROWS = 128, FB_C = 128, FB_ROWB = FB_C * 2, FB_TILE = FB_ROWS * FB_ROWB, FB_NT = 512;

__device__ __forceinline__ int fb_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ bf16x8 fb_tr2(const unsigned char* pa, const unsigned char* pb) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pa);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pb);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(FB_NT) void conv1x1_bwd_kernel(Bwd1x1Args p) {
  typedef __bf16 T;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // 2 x [dy tile | u tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;          // data gradient: rows 64 wm.., ci 32 wn..; weight gradient: co 64 wm.., ci 32 wn..
  const int r = lane & 31, hh = lane >> 5;

  const int ntiles = p.tiles_per_batch * p.B;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = min(ntiles, wg * p.tiles_per_wg);
  const int tile_end = min(ntiles, tile_begin + p.tiles_per_wg);

  // W^T slice for the data gradient: operand A rows = ci (packed "bwd" layout [ci][co], chunk c of row ci at c ^ (ci & 15))
  bf16x8 wfrag[FB_C / 16];
  {
    const int ci = wn * 32 + r;
    const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.w) + (size_t)ci * FB_ROWB;
#pragma unroll
    for (int kk = 0; kk < FB_C / 16; ++kk)
      wfrag[kk] = *reinterpret_cast<const bf16x8*>(wrow + (((2 * kk + hh) ^ (ci & 15)) << 4));
  }

  // bias gradient: a dy^T fragment is 8 rows of ONE output channel per lane, so db is a running sum per lane; the four waves
  // that hold the same fragment (wn = 0..3) take every fourth k-step each and leave their partial sums in columns 0, 32, 64,
  // 96 of the bias plane (the reducer adds them).  Round 2 ran it as MFMAs against a constant-one operand on the wn = 0
  // waves only: 1.5 x the matrix work of the other waves on SIMD 0, and every tile ends at a workgroup barrier.
  float bsum[2] = {0.f, 0.f};
  f32x16 accw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) accw[j][e] = 0.f;

  auto decode = [&](int tile, int& b, int& t0) {           // scalars: the V#s below must live in SGPRs
    b = __builtin_amdgcn_readfirstlane(tile / p.tiles_per_batch);
    t0 = __builtin_amdgcn_readfirstlane((tile - b * p.tiles_per_batch) * FB_ROWS);
  };
  const unsigned pitch_dy = (unsigned)p.lddy * 2u, pitch_u = (unsigned)p.ldu * 2u, pitch_dx = (unsigned)p.lddx * 2u;
  // group g = wave + 8 q covers rows 4 g + (lane >> 4) = 4 wave + (lane >> 4) + 32 q: fb_swz(row) does not depend on q
  const int srow = 4 * wave + (lane >> 4);
  const unsigned schunk = (unsigned)(((lane & 15) ^ fb_swz(srow)) << 4);
  auto stage = [&](int tile, int buf) {
    int b, t0;
    decode(tile, b, t0);
    const UntrackedRsrc rdy = untracked_rsrc(p.dy, (long long)b * p.dy_bs * 2, (unsigned)p.T * pitch_dy);
    const UntrackedRsrc ru = untracked_rsrc(p.u, (long long)b * p.u_bs * 2, (unsigned)p.T * pitch_u);
    unsigned char* base = smem + (size_t)buf * 2 * FB_TILE + wave * 1024;
    unsigned vdy = (unsigned)(t0 + srow) * pitch_dy + schunk, vu = (unsigned)(t0 + srow) * pitch_u + schunk;
#pragma unroll
    for (int q = 0; q < (FB_ROWS / 4) / (FB_NT / 64); ++q) {       // rows >= T: out of range, read as zero
      untracked_dma16(rdy, vdy, base + q * (FB_NT / 64) * 1024);
      untracked_dma16(ru, vu, base + FB_TILE + q * (FB_NT / 64) * 1024);
      vdy += 32u * pitch_dy; vu += 32u * pitch_u;
    }
  };

  // per-lane offsets of the transposed fragments (the k-step advances rows by 16, which keeps swz)
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3, thh = tg >> 1;
  const int ra = 8 * thh + tq, rb = ra + 4;
  const int col_a = wm * 64 + 16 * (tg & 1) + 4 * tp;          // dy^T fragment of co tile j: + 32 j  (byte offset ^ 64)
  const int col_b = wn * 32 + 16 * (tg & 1) + 4 * tp;          // u fragment
  const int offa0 = ra * FB_ROWB + (((col_a >> 3) ^ fb_swz(ra)) << 4) + (col_a & 7) * 2;
  const int offa1 = rb * FB_ROWB + (((col_a >> 3) ^ fb_swz(rb)) << 4) + (col_a & 7) * 2;
  const int offb0 = ra * FB_ROWB + (((col_b >> 3) ^ fb_swz(ra)) << 4) + (col_b & 7) * 2;
  const int offb1 = rb * FB_ROWB + (((col_b >> 3) ^ fb_swz(rb)) << 4) + (col_b & 7) * 2;
  const int swz_r = fb_swz(r);                                   // rows 64 wm + 32 i + r share it

  if (tile_begin < tile_end) stage(tile_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first tile (and the weights); later tiles: counted wait at the END
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    int b, t0;
    decode(tile, b, t0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave's part of this tile landed; the other buffer is free again
    if (tile + 1 < tile_end) stage(tile + 1, buf ^ 1);
    const unsigned char* dyt = smem + (size_t)buf * 2 * FB_TILE;
    const unsigned char* ut = dyt + FB_TILE;

    // The two halves of a tile's work only READ the tile, so their order is free per wave.  Waves w and w + 4 share a
    // SIMD: one runs data gradient -> weight gradient, the other the reverse, so that on every SIMD the LDS-heavy transposed
    // reads and MFMAs of one overlap the VALU epilogue and stores of the other (in lockstep the phases add up: per tile
    // ~4.3 k cycles of LDS reads + 3 k of MFMA + 2 k of VALU = the 11.4 k cycles measured).
    auto data_gradient = [&]() {
    // ---- data gradient, transposed: D^T[ci][row] = sum_co Wt[ci][co] * dy[row][co]
      f32x16 accd[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) accd[i][e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < FB_C / 16; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wm * 64 + 32 * i + r;
          bf16x8 bv = *reinterpret_cast<const bf16x8*>(dyt + row * FB_ROWB + (((2 * kk + hh) ^ swz_r) << 4));
          accd[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[kk], bv, accd[i], 0, 0, 0);
        }
      }
      {
        const __amdgpu_buffer_rsrc_t rdx = ws_rsrc(p.dx, (long long)b * p.dx_bs * 2, (unsigned)p.T * pitch_dx);
        const int len_out = p.lens_out ? scalar_load_i32(p.lens_out + b) : 0x7fffffff;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wm * 64 + 32 * i + r;
          const int t = t0 + row;
          const float keep_row = (t >= len_out) ? 0.f : 1.f;
          unsigned yp[8];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            // element 4g + k of this lane = input channel wn*32 + 8g + 4hh + k
            const bf16x4 uv = *reinterpret_cast<const bf16x4*>(ut + row * FB_ROWB + (((wn * 4 + g) ^ swz_r) << 4) + 8 * hh);
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float v = (float)(T)accd[i][4 * g + k];
              o[k] = (((float)uv[k] != 0.f) ? v * p.scale : 0.f) * keep_row;
            }
            yp[2 * g] = pack_bf16x2(o[0], o[1]);
            yp[2 * g + 1] = pack_bf16x2(o[2], o[3]);
          }
          // lanes r / r + 32 hold channels {0-3, 8-11, ..} / {4-7, 12-15, ..} of one row -> 16-byte pieces
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              auto sw = __builtin_amdgcn_permlane32_swap(yp[4 * h2 + d], yp[4 * h2 + 2 + d], false, false);
              yp[4 * h2 + d] = sw[0]; yp[4 * h2 + 2 + d] = sw[1];
            }
          {                                               // rows >= T: out of range, dropped -- but ISSUED (the wait below counts them)
            const unsigned vo = (unsigned)t * pitch_dx + (unsigned)(wn * 32 + 8 * hh) * 2u;
            __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[0], (int)yp[1], (int)yp[2], (int)yp[3]}, rdx, (int)vo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[4], (int)yp[5], (int)yp[6], (int)yp[7]}, rdx, (int)(vo + 32u), 0, 0);
          }
        }
      }

    };
    auto weight_gradient = [&]() {
    // ---- weight gradient: dW[co][ci] += sum_rows dy[row][co] * u[row][ci]  (rows beyond T are zero in LDS)
#pragma unroll
      for (int k0 = 0; k0 < FB_ROWS / 16; ++k0) {
        const int ko = k0 * 16 * FB_ROWB;
        const bf16x8 bfr = fb_tr2(ut + offb0 + ko, ut + offb1 + ko);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 afr = fb_tr2(dyt + ((offa0 + ko) ^ (64 * j)), dyt + ((offa1 + ko) ^ (64 * j)));
          accw[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, accw[j], 0, 0, 0);
          if (p.with_bias && (k0 & 3) == wn) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 aw = __builtin_bit_cast(u32x4, afr);
            float s8 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) s8 += __uint_as_float(aw[q] << 16) + __uint_as_float(aw[q] & 0xffff0000u);
            bsum[j] += s8;
          }
        }
      }
    };
    if (wm == 0) { data_gradient(); weight_gradient(); }
    else { weight_gradient(); data_gradient(); }
    // the next tile's 8 DMA instructions were issued before this tile's 4 dx stores: all but those stores are done
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }

  // ---- partial dW / db of this workgroup -> slab[wg][blk = wm][plane][64][128]
  float* out = p.slab + ((size_t)wg * 2 + wm) * 2 * 64 * FB_C;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = 32 * j + (e & 3) + 8 * (e >> 2) + 4 * hh;
      out[(size_t)row * FB_C + wn * 32 + r] = accw[j][e];
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {                         // lane l: output channel 32 j + (l & 31), rows of its k-half
    const float other = __shfl_xor(bsum[j], 32);
    if (hh == 0) out[(size_t)(64 + 32 * j + r) * FB_C + 32 * wn] = bsum[j] + other;
  }
}

static int bwd1x1_nwg(const smt_conv_desc* d) {
  const long long ntiles = (long long)((d->t_out + FB_ROWS - 1) / FB_ROWS) * d->batch;
  // one workgroup per CU (128 KiB of LDS); at least two tiles per workgroup, whole XCD octets
  long long nwg = std::min<long long>(256, std::max<long long>(8, (ntiles + fused_min_tpw() - 1) / fused_min_tpw()));
  return (int)((nwg + 7) / 8 * 8);
}

}  // namespace smt

using namespace smt;

static bool bwd1x1_ok(const smt_conv_desc* d) {
  return d->dtype == SMT_BF16 && d->taps == 1 && d->c_in == FB_C && d->c_out == FB_C && d->stride == 1 &&
         d->dilation == 1 && d->padding == 0 && d->out_stride == 1 && d->out_offset == 0 && d->t_in == d->t_out &&
         d->t_y == d->t_out && d->w_swizzled && d->zero_page && d->act_grad && d->act_grad_src && !d->res &&
         !d->act_out && !d->bias && !d->lens_in;
}

extern "C" size_t smt_conv1x1_bwd_workspace_bytes(const smt_conv_desc* d) {
  if (!d) return 0;
  return (size_t)bwd1x1_nwg(d) * 2 * 2 * 64 * FB_C * sizeof(float);
}

extern "C" int smt_conv1x1_bwd(const smt_conv_desc* d, float* dweight, int64_t stride_out, int64_t stride_in,
                               float* dbias, void* workspace, size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(d && d->x && d->y && d->w && dweight && workspace, "smt_conv1x1_bwd: null pointer");
  SMT_CHECK_ARG(bwd1x1_ok(d),
                "smt_conv1x1_bwd: needs the bf16 1x1 data-gradient descriptor of a 128->128 layer (swizzled weights, "
                "zero_page, act_grad set, no bias/res/act_out/lens_in)");
  SMT_CHECK_ARG(d->ld_x % 8 == 0 && d->ld_y % 8 == 0 && d->ld_act % 8 == 0, "smt_conv1x1_bwd: row pitches must keep 16-byte alignment");
  SMT_CHECK_ARG(workspace_bytes >= smt_conv1x1_bwd_workspace_bytes(d), "smt_conv1x1_bwd: workspace too small");
  const int nwg = bwd1x1_nwg(d);
  if (d->batch > 0 && d->t_out > 0) {
    Bwd1x1Args a;
    a.dy = d->x; a.u = d->act_grad_src; a.w = d->w; a.dx = d->y; a.slab = (float*)workspace;
    a.lens_out = d->lens_out;
    a.dy_bs = d->bs_x; a.u_bs = d->bs_act; a.dx_bs = d->bs_y;
    a.lddy = d->ld_x; a.ldu = d->ld_act; a.lddx = d->ld_y;
    a.B = d->batch; a.T = d->t_out;
    a.tiles_per_batch = (d->t_out + FB_ROWS - 1) / FB_ROWS;
    const long long ntiles = (long long)a.tiles_per_batch * d->batch;
    a.tiles_per_wg = (int)((ntiles + nwg - 1) / nwg);
    a.with_bias = dbias ? 1 : 0;
    a.scale = d->drop_scale;
    (void)hipFuncSetAttribute((const void*)conv1x1_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    conv1x1_bwd_kernel<<<nwg, FB_NT, 4 * FB_TILE, stream>>>(a);
    SMT_CHECK_LAUNCH("conv1x1_bwd");
  }
  const int jmap[1] = {0};
  const int n_chunks = (d->batch > 0 && d->t_out > 0) ? nwg : 0;
  // weight gradient of the FORWARD layer: rows = forward output channels (the channels of dy = desc c_in),
  // columns = forward input channels (the channels of u = desc c_out)
  return launch_wgrad_reduce((const float*)workspace, dweight, dbias, n_chunks, 2, 1, 1, FB_C, FB_C, FB_C, stride_out,
                             stride_in, 0, jmap, stream, /*bias_cols=*/4);
}
