// Fused backward of K1 of GatedHiFiBlock for all branches at once (reference models/vqvae/resnet.py:205-216: the
// `Conv1d(w, 2w, 1)` at the head of every branch; here the four branches are one 64 -> 512 layer):
//
//   dx[t, ci]   = keep(t) * sum_co dh[t, co] * W[co][ci] + res[t, ci]          (data gradient + block residual)
//   dW[co][ci]  = sum_t dh[t, co] * x[t, ci],   db[co] = sum_t dh[t, co]       (x rows >= lens read as 0)
//
// Both need the 1 KiB row of dh, which is 3/4 of all bytes this layer moves; computed separately it is read twice.
// A persistent workgroup streams 64-row tiles of dh (64 KiB) and x (8 KiB) through an LDS-DMA double buffer once:
//   * data gradient: four 32 x 32 tiles of dx, each computed by TWO waves that split the 512 output channels (the
//     contraction) in halves and keep their W^T fragments in registers (fetching them through L1 per tile left the
//     kernel latency-bound: 1170 vs 695 us at the top level, tools/ablate_k1bwd.sh); the upper half hands its
//     partial tile to the lower one through LDS (fixed order: bitwise reproducible), which adds the residual in
//     registers, pairs lanes with v_permlane32_swap and stores 16-byte pieces;
//   * weight gradient: wave w owns output channels 64 w .. 64 w + 63 (4 accumulator tiles + 2 bias tiles kept in
//     registers for the whole run), fragments of dh^T and x through ds_read_b64_tr_b16.
// Swizzles: dh rows are 1 KiB -- chunk c of row r sits at c ^ (((r & 3) << 2) | ((r >> 2) & 3)) (16 rows -> 16
// different chunks for row fragments, 4 consecutive rows 64 B apart for transposed fragments); x rows are 128 B --
// chunk c of row r at c ^ (((r >> 1) & 3) << 1).
// Each workgroup leaves its partial dW / db in a slab; conv_wgrad_reduce_kernel sums the slabs in fixed order.
//
// Round 3: the tile loop owns its vector-memory waits (see conv_k3gate.hip / conv_common.h).  Vector memory completes in
// issue order, so everything this tile needs from global memory -- the residual rows -- is issued BEFORE the next tile's
// prefetch, the prefetch and those loads are untracked (no compiler-placed waits), lens[b] is a scalar load, the two
// workgroup barriers are raw s_barrier + lgkmcnt(0) (a __syncthreads() also drains vector memory), and the stores stay in
// flight across the tile boundary: per tile the storing waves wait vmcnt(9) for the residual rows and vmcnt(2) at the end.
#include <algorithm>

#include "conv_common.h"

namespace smt {

struct K1BwdArgs {
  const void* dh; const void* x; const void* w; const void* res; void* dx; float* slab;
  const int* lens;
  long long dh_bs, x_bs, res_bs, dx_bs;
  int lddh, ldx, ldres, lddx;
  int B, T, tiles_per_batch, tiles_per_wg, with_bias;
};

constexpr int KB_ROWS = 64, KB_CO = 512, KB_CI = 64, KB_NT = 512;
constexpr int KB_DH = KB_ROWS * KB_CO * 2, KB_X = KB_ROWS * KB_CI * 2, KB_STAGE = KB_DH + KB_X;   // 64 KiB + 8 KiB
constexpr int KB_RED = 4 * 32 * 32 * 4;   // partial data-gradient tiles of waves 4..7 (behind the two stages)

__device__ __forceinline__ int kb_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int kb_swz_x(int row) { return ((row >> 1) & 3) << 1; }

__device__ __forceinline__ bf16x8 kb_tr2(const unsigned char* pa, const unsigned char* pb) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pa);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pb);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(KB_NT) void conv_k1_bwd_kernel(K1BwdArgs p) {
  typedef __bf16 T;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // 2 x [dh tile | x tile], then the reduction scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;

  const int ntiles = p.tiles_per_batch * p.B;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = min(ntiles, wg * p.tiles_per_wg);
  const int tile_end = min(ntiles, tile_begin + p.tiles_per_wg);

  // bias gradient: the dh^T fragment of a lane is 8 rows of ONE output channel, so db is a running sum per lane (round 3;
  // the MFMA against a constant-one operand it replaces held 32 accumulator + 4 operand registers and the kernel spilled:
  // a scratch reload inside the tile loop is a tracked load, i.e. an s_waitcnt vmcnt(0) that drains the prefetch)
  float bsum[2] = {0.f, 0.f};
  f32x16 accw[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) accw[a][c][e] = 0.f;
  }

  auto decode = [&](int tile, int& b, int& t0) {           // scalars: the V#s below must live in SGPRs
    b = __builtin_amdgcn_readfirstlane(tile / p.tiles_per_batch);
    t0 = __builtin_amdgcn_readfirstlane((tile - b * p.tiles_per_batch) * KB_ROWS);
  };
  const unsigned pitch_dh = (unsigned)p.lddh * 2u, pitch_x = (unsigned)p.ldx * 2u, pitch_dx = (unsigned)p.lddx * 2u;
  const int xrow = 8 * wave + (lane >> 3);
  const unsigned xoff0 = (unsigned)xrow * pitch_x + (unsigned)(((lane & 7) ^ kb_swz_x(xrow)) << 4);
  constexpr int KB_NDMA = KB_ROWS / (KB_NT / 64) + 1;       // LDS-DMA instructions per wave and tile
  auto stage = [&](int tile, int buf) {
    int b, t0;
    decode(tile, b, t0);
    const int len = p.lens ? min(scalar_load_i32(p.lens + b), p.T) : p.T;
    const UntrackedRsrc rdh = untracked_rsrc(p.dh, (long long)b * p.dh_bs * 2, (unsigned)p.T * pitch_dh);
    const UntrackedRsrc rx = untracked_rsrc(p.x, (long long)b * p.x_bs * 2, (unsigned)len * pitch_x);
    unsigned char* base = smem + (size_t)buf * KB_STAGE;
    int ln = lane;
    asm volatile("" : "+v"(ln));                                  // the eight swizzled lane offsets are formed here, per tile: hoisted
#pragma unroll                                                    // out of the tile loop they cost eight registers and the kernel spills
    for (int q = 0; q < KB_ROWS / (KB_NT / 64); ++q) {            // dh: one 1 KiB row per instruction (rows >= T read as zero)
      const int row = wave + (KB_NT / 64) * q;
      untracked_dma16(rdh, (unsigned)(t0 + row) * pitch_dh + (unsigned)((ln ^ kb_swz(row)) << 4), base + row * 1024);
    }
    untracked_dma16(rx, xoff0 + (unsigned)t0 * pitch_x, base + KB_DH + wave * 1024);   // x: 8 rows x 8 chunks (rows >= len: zero)
  };

  // weight-gradient fragment offsets (the k-step advances rows by 16, which keeps both swizzles)
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3, thh = tg >> 1;
  const int ra = 8 * thh + tq, rb = ra + 4;
  const int col_a = wave * 64 + 16 * (tg & 1) + 4 * tp;           // dh^T fragment of co tile a: + 32 a (chunk + 4: bit 2 is free)
  const int col_b = 16 * (tg & 1) + 4 * tp;                       // x fragment of ci tile c: + 32 c   (chunk + 4: bit 2 is free)
  const int offa0 = ra * 1024 + (((col_a >> 3) ^ kb_swz(ra)) << 4) + (col_a & 7) * 2;
  const int offa1 = rb * 1024 + (((col_a >> 3) ^ kb_swz(rb)) << 4) + (col_a & 7) * 2;
  const int offb0 = KB_DH + ra * 128 + (((col_b >> 3) ^ kb_swz_x(ra)) << 4) + (col_b & 7) * 2;
  const int offb1 = KB_DH + rb * 128 + (((col_b >> 3) ^ kb_swz_x(rb)) << 4) + (col_b & 7) * 2;

  // data-gradient tile of waves w and w + 4: rows 32 di.., input channels 32 dc..; wave w + 4 takes the upper half of
  // the contraction (output channels 256..511).  W^T fragments (packed [ci][co], plain) live in registers.
  const int di = (wave >> 1) & 1, dc = wave & 1, kh = wave >> 2;
  bf16x8 wfrag[KB_CO / 32];
  {
    const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.w) + (size_t)(32 * dc + r) * (KB_CO * 2);
#pragma unroll
    for (int kk = 0; kk < KB_CO / 32; ++kk) wfrag[kk] = *reinterpret_cast<const bf16x8*>(wrow + ((2 * (kk + 16 * kh) + hh) << 4));
  }
  float* red = reinterpret_cast<float*>(smem + 2 * KB_STAGE) + (wave & 3) * 1024;   // [16 e][64 lanes] of tile (di, dc)

  if (tile_begin < tile_end) stage(tile_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first tile (and the weights); later tiles: counted waits
#pragma unroll
  for (int kk = 0; kk < KB_CO / 32; ++kk) asm volatile("" : "+v"(wfrag[kk]));   // tell the compiler the weight loads are done: it
                                                       // would otherwise re-wait for them (vmcnt(18) .. vmcnt(3)) inside the loop
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    int b, t0;
    decode(tile, b, t0);
    const bool more = tile + 1 < tile_end;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave's part of this tile landed; the other buffer is free again
    const unsigned char* base = smem + (size_t)buf * KB_STAGE;

    {
      // ---- data gradient, transposed: D^T[ci][row] = sum_co Wt[ci][co] * dh[row][co], contraction split over two waves
      const int row = 32 * di + r;
      const int t = t0 + row;
      f32x16 accd;
#pragma unroll
      for (int e = 0; e < 16; ++e) accd[e] = 0.f;
      const int swz_r = kb_swz(row);
#pragma unroll
      for (int kk = 0; kk < KB_CO / 32; ++kk) {
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(base + row * 1024 + (((2 * (kk + 16 * kh) + hh) ^ swz_r) << 4));
        accd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[kk], bv, accd, 0, 0, 0);
      }
      // residual rows of THIS tile (waves 0..3: channels 32 dc + 8 g + 4 hh + 0..3 of row 32 di + r), THEN the prefetch:
      // vector memory completes in issue order, so what this tile still needs must be older than the next tile's DMA.
      // (Issued here, after the MFMAs, the eight residual registers are not live across them: before them the kernel spills.)
      unsigned rvw[4][2];
      if (!kh) {
        const unsigned char* rbase = reinterpret_cast<const unsigned char*>(p.res) + ((long long)b * p.res_bs) * 2;
        asm volatile("" : "+s"(rbase));
        const unsigned ro = (unsigned)min(t, p.T - 1) * (unsigned)p.ldres * 2u + (unsigned)(32 * dc + 4 * hh) * 2u;
#pragma unroll
        for (int g = 0; g < 4; ++g) untracked_load8(rbase, ro + 16u * g, rvw[g][0], rvw[g][1]);
      }
      if (more) stage(tile + 1, buf ^ 1);
      if (kh) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[e * 64 + lane] = accd[e];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // partial tiles of waves 4..7 are in LDS
      if (!kh) {
        const int len = p.lens ? scalar_load_i32(p.lens + b) : 0x7fffffff;
        const float keep_row = (t >= len) ? 0.f : 1.f;
        // the residual loads are older than the prefetch: all but its KB_NDMA instructions are done
        if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KB_NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bf16x4 rv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          asm volatile("" : "+v"(rvw[g][0]), "+v"(rvw[g][1]));          // values as of AFTER the wait
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          rv[g] = __builtin_bit_cast(bf16x4, u32x2{rvw[g][0], rvw[g][1]});
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) accd[e] += red[e * 64 + lane];      // lower half + upper half, always in this order
        unsigned yp[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = fmaf((float)(T)accd[4 * g + k], keep_row, (float)rv[g][k]);
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
        {                                               // rows >= T: out of range, dropped -- but ISSUED
          const __amdgpu_buffer_rsrc_t rdx = ws_rsrc(p.dx, (long long)b * p.dx_bs * 2, (unsigned)p.T * pitch_dx);
          const unsigned vo = (unsigned)t * pitch_dx + (unsigned)(32 * dc + 8 * hh) * 2u;
          __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[0], (int)yp[1], (int)yp[2], (int)yp[3]}, rdx, (int)vo, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[4], (int)yp[5], (int)yp[6], (int)yp[7]}, rdx, (int)(vo + 32u), 0, 0);
        }
      }
    }

    // ---- weight gradient: dW[co][ci] += sum_rows dh[row][co] * x[row][ci]   (rows beyond T / lens are zero in LDS)
#pragma unroll
    for (int k0 = 0; k0 < KB_ROWS / 16; ++k0) {
      bf16x8 bfr[2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
        bfr[c] = kb_tr2(base + ((offb0 + k0 * 16 * 128) ^ (64 * c)), base + ((offb1 + k0 * 16 * 128) ^ (64 * c)));
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const bf16x8 afr = kb_tr2(base + ((offa0 + k0 * 16 * 1024) ^ (64 * a)), base + ((offa1 + k0 * 16 * 1024) ^ (64 * a)));
#pragma unroll
        for (int c = 0; c < 2; ++c) accw[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr[c], accw[a][c], 0, 0, 0);
        if (p.with_bias) {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 aw = __builtin_bit_cast(u32x4, afr);
          float s8 = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) s8 += __uint_as_float(aw[j] << 16) + __uint_as_float(aw[j] & 0xffff0000u);
          bsum[a] += s8;
        }
      }
    }
    // the prefetch is older than this tile's two dx stores (waves 0..3) / is the youngest (waves 4..7)
    __builtin_amdgcn_sched_barrier(0);
    if (!kh) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  // ---- partial dW / db of this workgroup -> slab[wg][blk = wave][plane][64 co][64 ci]
  float* out = p.slab + ((size_t)wg * 8 + wave) * 2 * 64 * KB_CI;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = 32 * a + (e & 3) + 8 * (e >> 2) + 4 * hh;
#pragma unroll
      for (int c = 0; c < 2; ++c) out[(size_t)row * KB_CI + 32 * c + r] = accw[a][c][e];
    }
  // db: lane l summed output channel 32 a + (l & 31) over the rows of its k-half; halves added in a fixed order; the
  // reducer reads column 0 of the bias plane
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const float other = __shfl_xor(bsum[a], 32);
    if (hh == 0) out[(size_t)(64 + 32 * a + r) * KB_CI] = bsum[a] + other;
  }
}

static int k1_bwd_nwg(int batch, int t) {
  const long long ntiles = (long long)((t + KB_ROWS - 1) / KB_ROWS) * batch;
  long long nwg = std::min<long long>(256, std::max<long long>(8, (ntiles + fused_min_tpw() - 1) / fused_min_tpw()));   // one workgroup per CU (144 KiB of LDS)
  return (int)((nwg + 7) / 8 * 8);
}

}  // namespace smt

using namespace smt;

extern "C" size_t smt_conv_k1_bwd_workspace_bytes(int batch, int t) {
  return (size_t)k1_bwd_nwg(batch, t) * 8 * 2 * 64 * KB_CI * sizeof(float);
}

extern "C" int smt_conv_k1_bwd(const void* dh, int64_t bs_dh, int ld_dh, const void* x, int64_t bs_x, int ld_x,
                               const void* w_packed_bwd, const void* res, int64_t bs_res, int ld_res, void* dx,
                               int64_t bs_dx, int ld_dx, const int* lens, int batch, int t, const void* zero_page,
                               float* dweight, int64_t stride_out, int64_t stride_in, float* dbias, void* workspace,
                               size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(dh && x && w_packed_bwd && res && dx && zero_page && dweight && workspace, "smt_conv_k1_bwd: null pointer");
  SMT_CHECK_ARG(ld_dh % 8 == 0 && ld_x % 8 == 0 && ld_res % 4 == 0 && ld_dx % 8 == 0, "smt_conv_k1_bwd: row pitches must keep alignment");
  SMT_CHECK_ARG(workspace_bytes >= smt_conv_k1_bwd_workspace_bytes(batch, t), "smt_conv_k1_bwd: workspace too small");
  const int nwg = k1_bwd_nwg(batch, t);
  if (batch > 0 && t > 0) {
    K1BwdArgs a;
    a.dh = dh; a.x = x; a.w = w_packed_bwd; a.res = res; a.dx = dx; a.slab = (float*)workspace; a.lens = lens;
    a.dh_bs = bs_dh; a.x_bs = bs_x; a.res_bs = bs_res; a.dx_bs = bs_dx;
    a.lddh = ld_dh; a.ldx = ld_x; a.ldres = ld_res; a.lddx = ld_dx;
    a.B = batch; a.T = t;
    a.tiles_per_batch = (t + KB_ROWS - 1) / KB_ROWS;
    const long long ntiles = (long long)a.tiles_per_batch * batch;
    a.tiles_per_wg = (int)((ntiles + nwg - 1) / nwg);
    a.with_bias = dbias ? 1 : 0;
    (void)hipFuncSetAttribute((const void*)conv_k1_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    conv_k1_bwd_kernel<<<nwg, KB_NT, 2 * KB_STAGE + KB_RED, stream>>>(a);
    SMT_CHECK_LAUNCH("conv_k1_bwd");
  }
  const int jmap[1] = {0};
  const int n_chunks = (batch > 0 && t > 0) ? nwg : 0;
  return launch_wgrad_reduce((const float*)workspace, dweight, dbias, n_chunks, 8, 1, 1, KB_CI, KB_CO, 64, stride_out,
                             stride_in, 0, jmap, stream);
}
