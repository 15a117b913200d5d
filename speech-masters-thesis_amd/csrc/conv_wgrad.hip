// Weight / bias gradients of the channels-last convolutions (autograd of F.conv1d /
// F.conv_transpose1d in the reference, models/vqvae/conv.py, resnet.py) on the matrix cores.
//
//   dw[j][co][ci] = sum_{b,t} dy[b, t*os + oo, co] * pro(x)[b, t*stride + j*dil - pad, ci]
//
// The contraction runs over ROWS (time), which is the slow axis of both channels-last operands.
// Tiles are staged row-major into LDS exactly as in the forward kernel (same prologue) and the MFMA
// fragments are read TRANSPOSED:  bf16 with ds_read_b64_tr_b16 (4 rows x 16 channels per 16-lane
// group, delivered channel-major -- the hardware transpose), fp32 with plain ds_read_b32 (one scalar
// per lane per MFMA at the fp32 rate).  A row shift for tap j only moves the row index, so no
// alignment constraint arises from odd dilations.
//
// Decomposition: workgroup = 4 waves (2x2) = a 64(co) x 64(ci) block of dw for ALL taps (<= 9 taps x 16
// accumulator registers) over one chunk of rows; partial blocks go to a slab and a second kernel
// reduces the chunks in fixed order (deterministic) into the fp32 torch-layout gradient.  The bias
// gradient rides along as one extra "tap" whose x operand is the constant 1.
#include <algorithm>
#include <cstdlib>

#include "conv_common.h"

namespace smt {

struct WgradArgs {
  const void* x; const void* dy; float* slab;
  const int* lens_in;
  long long x_bs, dy_bs;
  int ldx, ldy;
  int B, Tin, Tout, Ty, Cin, Cout;
  int taps, stride, dil, pad, out_stride, out_offset;
  int rows_per_chunk, chunks_per_batch, nblk_ci, nblk_co, with_bias;
  int rs;   // dilation-class row stride (LDS-DMA kernel), 1 = off
  int win, xstride;   // window mode (LDS-DMA kernel): the Cin "channels" of output row t are the 64-channel rows
                      // xstride t - pad, xstride t - pad + 1, .. of x side by side (Cin / 64 taps of a 64-channel conv)
};

// R = rows per staged tile; CIB = input channels per workgroup (CIB/32 wave columns, 2 wave rows);
// pitches chosen so that the transposed reads are bank-conflict-free: for ds_read_b64_tr_b16 a
// 32-lane half touches 4 rows x 2 16-column blocks, which need row pitch == 16 dwords (mod 64).
template <typename T> struct WTr;
template <> struct WTr<__bf16> {
  static constexpr int R = 128;
  static constexpr int pitch(int ch) { return ch + 32; }   // 64 ch: 192 B == 48 dwords; 128 ch: 320 B == 16 (mod 64)
};
template <> struct WTr<float> {
  static constexpr int R = 64;
  static constexpr int pitch(int ch) { return ch + 4; }
};

// transposed fragment: element e of lane (n = lane&31, hh = lane>>5) = tile[row0 + 8*hh + e][col0 + n]
__device__ __forceinline__ bf16x8 frag_tr_bf16(const __bf16* tile, int pitch, int row0, int col0, int lane) {
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int hh = g >> 1;
  const __bf16* a0 = tile + (row0 + 8 * hh + q) * pitch + col0 + 16 * (g & 1) + 4 * pp;
  const __bf16* a1 = a0 + 4 * pitch;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// NT = accumulator planes (taps + bias plane); CIB = input channels per workgroup; STRIDED = stride > 1
template <typename T, int NT, int CIB, bool STRIDED>
__global__ __launch_bounds__(CIB * 4) void conv_wgrad_kernel(WgradArgs p) {
  constexpr int EPV = Tr<T>::EPV;
  constexpr int R = WTr<T>::R;          // rows per staged tile
  constexpr int CB = 64;                // output channels per block
  constexpr int WNC = CIB / 32;         // wave columns
  constexpr int NTHR = 128 * WNC;
  constexpr int PITCH_DY = WTr<T>::pitch(CB), PITCH_X = WTr<T>::pitch(CIB);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNC, wn = wave % WNC;
  const int r = lane & 31, hh = lane >> 5;

  // XCD-aware order: the co-blocks of one row chunk run back to back on the same XCD (ids == mod 8),
  // so the second reader of a chunk's rows hits L2
  const int nblk = p.nblk_ci * p.nblk_co;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int blk = q % nblk;
  const int cgl = (q / nblk) * 8 + xcd;          // global chunk index
  if (cgl >= p.B * p.chunks_per_batch) return;
  const int co0 = (blk / p.nblk_ci) * CB, ci0 = (blk % p.nblk_ci) * CIB;
  const int b = cgl / p.chunks_per_batch;
  const int chunk = cgl % p.chunks_per_batch;
  const int t_begin = chunk * p.rows_per_chunk;
  const int t_end = min(p.Tout, t_begin + p.rows_per_chunk);

  const int rows_x = (R - 1) * p.stride + (p.taps - 1) * p.dil + 1;
  T* lds_dy = reinterpret_cast<T*>(smem);                 // [R][PITCH_DY]
  T* lds_x = lds_dy + R * PITCH_DY;                       // [rows_x][PITCH_X]
  const T* xg = reinterpret_cast<const T*>(p.x) + (long long)b * p.x_bs;
  const T* dyg = reinterpret_cast<const T*>(p.dy) + (long long)b * p.dy_bs;
  const int len_in = p.lens_in ? min(p.lens_in[b], p.Tin) : p.Tin;
  const bool bias_plane = p.with_bias && (ci0 == 0);
  const int ntaps = p.taps;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  for (int t0 = t_begin; t0 < t_end; t0 += R) {
    __syncthreads();
    // dy tile (rows t0..t0+R, channels co0..co0+64) and haloed x tile: all loads of a batch are
    // issued before the first LDS store so that one memory latency covers UB vectors per thread
    constexpr int VPR_DY = CB / EPV, VPR_X = CIB / EPV;
    constexpr int UB = (NT > 6) ? 4 : 8;   // fewer in-flight staging registers when the accumulators are many
    const int tin0 = t0 * p.stride - p.pad;
    const int n_dy = R * VPR_DY, n_x = rows_x * VPR_X;
    for (int f0 = tid; f0 < n_dy + n_x; f0 += NTHR * UB) {
      Vec<T, EPV> v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int f = f0 + NTHR * u;
#pragma unroll
        for (int e = 0; e < EPV; ++e) v[u].v[e] = (T)0.f;
        if (f < n_dy) {
          const int row = f / VPR_DY, cv = f % VPR_DY;
          const int t = t0 + row;
          const int ty = t * p.out_stride + p.out_offset;
          if (t < t_end && ty < p.Ty && co0 + cv * EPV < p.Cout)
            v[u] = *reinterpret_cast<const Vec<T, EPV>*>(dyg + (long long)ty * p.ldy + co0 + cv * EPV);
        } else if (f < n_dy + n_x) {
          const int g = f - n_dy;
          const int row = g / VPR_X, cv = g % VPR_X;
          const int tin = tin0 + row;
          if (tin >= 0 && tin < len_in && ci0 + cv * EPV < p.Cin)
            v[u] = *reinterpret_cast<const Vec<T, EPV>*>(xg + (long long)tin * p.ldx + ci0 + cv * EPV);
        }
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int f = f0 + NTHR * u;
        if (f < n_dy) {
          *reinterpret_cast<Vec<T, EPV>*>(lds_dy + (f / VPR_DY) * PITCH_DY + (f % VPR_DY) * EPV) = v[u];
        } else if (f < n_dy + n_x) {
          const int g = f - n_dy;
          *reinterpret_cast<Vec<T, EPV>*>(lds_x + (g / VPR_X) * PITCH_X + (g % VPR_X) * EPV) = v[u];
        }
      }
    }
    __syncthreads();

    if constexpr (sizeof(T) == 2) {
      const __bf16* dyt = reinterpret_cast<const __bf16*>(lds_dy);
      const __bf16* xt = reinterpret_cast<const __bf16*>(lds_x);
      bf16x8 ones;
#pragma unroll
      for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
      // bound the unrolling: hoisting every transposed read of 8 k-steps x NT taps would spill
#pragma unroll 1
      for (int k0 = 0; k0 < R; k0 += 16) {
        bf16x8 a = frag_tr_bf16(dyt, PITCH_DY, k0, wm * 32, lane);   // A[co][k] = dy[k][co]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (j < ntaps) {
            // B[k][ci] = x[(k)*stride + j*dil][ci]   (stride 1 for every transposed read;
            // strided convs read rows k*stride: handled by row index arithmetic below)
            bf16x8 bfrag;
            if constexpr (!STRIDED) {
              bfrag = frag_tr_bf16(xt, PITCH_X, k0 + j * p.dil, wn * 32, lane);
            } else {
              // strided rows are not a dense 4-row block: gather element-wise
              typedef short s16x8 __attribute__((ext_vector_type(8)));
              s16x8 tmp;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const short* sp = reinterpret_cast<const short*>(xt) +
                                  ((k0 + 8 * hh + e) * p.stride + j * p.dil) * PITCH_X + wn * 32 + r;
                tmp[e] = *sp;
              }
              bfrag = __builtin_bit_cast(bf16x8, tmp);
            }
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag, acc[j], 0, 0, 0);
          } else if (j == ntaps && bias_plane) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, acc[j], 0, 0, 0);
          }
        }
      }
    } else {
      const float* dyt = reinterpret_cast<const float*>(lds_dy);
      const float* xt = reinterpret_cast<const float*>(lds_x);
#pragma unroll 2
      for (int k0 = 0; k0 < R; k0 += 2) {
        const float a = dyt[(k0 + hh) * PITCH_DY + wm * 32 + r];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (j < ntaps) {
            const float bv = xt[((k0 + hh) * p.stride + j * p.dil) * PITCH_X + wn * 32 + r];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[j], 0, 0, 0);
          } else if (j == ntaps && bias_plane) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, 1.0f, acc[j], 0, 0, 0);
          }
        }
      }
    }
  }
  // partial block -> slab[chunk_global][plane][co 64][ci 64] for this (co,ci) block
  const int planes = ntaps + 1;   // slab layout always carries the bias plane (zeros when not requested)
  float* out = p.slab + ((size_t)cgl * nblk + blk) * (size_t)planes * CB * CIB;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (j >= planes) break;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;  // co
      const int col = wn * 32 + r;                                 // ci
      out[((size_t)j * CB + row) * CIB + col] = acc[j][e];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (bf16, stride 1, C_in % 128 == 0, C_out % 64 == 0): the dy / x row tiles go
// HBM/L2 -> LDS with global_load_lds_dwordx4 into a DOUBLE buffer (tile i+1 lands while tile i is
// multiplied), rows unpadded.  The transposed reads stay conflict-free through an XOR swizzle of the
// 16-byte chunk index, applied on the source address while staging:
//   x  rows (256 B): chunk ^ ((row & 3) << 2)        dy rows (128 B): chunk ^ (((row >> 1) & 1) << 2)
// (a 32-lane half of ds_read_b64_tr_b16 touches 4 rows x 64 B; the swizzles put those on 4 distinct
// 64-byte slots of the 256-byte bank row).
__device__ __forceinline__ void wg_dma16(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_dst_wave_base, 16, 0, 0);
}

// element e of lane (n = lane&31, hh = lane>>5) = tile[row0 + 8*hh + e][col0 + n], rows of ROWB bytes, swizzled
template <int ROWB, bool IS_X>
__device__ __forceinline__ bf16x8 frag_tr_swz(const unsigned char* tile, int row0, int col0, int lane) {
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int hh = g >> 1;
  const int col = col0 + 16 * (g & 1) + 4 * pp;
  const int chunk = col >> 3, within = (col & 7) * 2;
  const int ra = row0 + 8 * hh + q, rb = ra + 4;
  const int fa = IS_X ? ((ra & 3) << 2) : (((ra >> 1) & 1) << 2);
  const int fb = IS_X ? ((rb & 3) << 2) : (((rb >> 1) & 1) << 2);
  const unsigned char* a0 = tile + ra * ROWB + ((chunk ^ fa) << 4) + within;
  const unsigned char* a1 = tile + rb * ROWB + ((chunk ^ fb) << 4) + within;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

#ifndef SMT_WABL
#define SMT_WABL 0   // ablation build switches (tools/ablate_wgrad.sh): 1 stage once, 2 no fragment reads, 4 no MFMA
#endif
constexpr int WABL = SMT_WABL;

template <int NT>
__global__ __launch_bounds__(512) void conv_wgrad_dma_kernel(WgradArgs p, const __bf16* __restrict__ zero_page) {
  typedef __bf16 T;
  constexpr int R = 128, CB = 64, CIB = 128, WNC = 4, NTHR = 512;
  constexpr int DYB = CB * 2, XB = CIB * 2;         // row bytes
  constexpr int DY_BYTES = R * DYB;                 // 16 KiB
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNC, wn = wave % WNC;
  const int r = lane & 31, hh = lane >> 5;

  const int nblk = p.nblk_ci * p.nblk_co;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int blk = q % nblk;
  const int cgl = (q / nblk) * 8 + xcd;
  if (cgl >= p.B * p.rs * p.chunks_per_batch) return;
  const int co0 = (blk / p.nblk_ci) * CB, ci0 = (blk % p.nblk_ci) * CIB;
  // dilation classes (see conv_gemm_dma_kernel): rs > 1 runs the dilated conv as rs dense convs over the
  // row classes t = cls (mod rs); a "batch item" here is one (batch, class) pair
  const int rs = p.rs;
  const int bb = cgl / p.chunks_per_batch;
  const int b = bb / rs, cls = bb - b * rs;
  const int chunk = cgl % p.chunks_per_batch;
  const int Tc = (p.Tout - cls + rs - 1) / rs;
  const int t_begin = chunk * p.rows_per_chunk;
  const int t_end = min(Tc, t_begin + p.rows_per_chunk);

  const int rows_x = (R - 1) + (p.taps - 1) * p.dil + 1;
  const int rows_x_pad = (rows_x + 3) & ~3;
  const size_t buf_bytes = (size_t)DY_BYTES + (size_t)rows_x_pad * XB;
  const T* xg = reinterpret_cast<const T*>(p.x) + (long long)b * p.x_bs + (long long)cls * p.ldx;
  const T* dyg = reinterpret_cast<const T*>(p.dy) + (long long)b * p.dy_bs + (long long)cls * p.ldy;
  const long long ldx = (long long)p.ldx * rs, ldy = (long long)p.ldy * rs;
  const int len_full = p.lens_in ? min(p.lens_in[b], p.Tin) : p.Tin;
  const int len_in = max(0, (len_full - cls + rs - 1) / rs);
  const bool bias_plane = p.with_bias && (ci0 == 0);
  const int ntaps = p.taps;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  // Round 3: untracked LDS-DMA through range-checked V#s (conv_common.h): the compiler used to put a wait for the NEXT tile's
  // prefetch in front of the first transposed read of the current tile, so nothing overlapped (it mattered little while the
  // kernel only ran 5- and 9-tap layers with 40+ MFMAs per tile; in window mode a tile is 8-16 MFMAs).  An offset of ~0 is
  // out of range of any descriptor: such lanes read zero (rows past the chunk, before the item, past the valid length).
  const UntrackedRsrc rdy = untracked_rsrc(dyg + co0, 0, (unsigned)min((long long)0xfffffff0ll, (long long)Tc * ldy * 2));
  const UntrackedRsrc rxx = untracked_rsrc(xg + (p.win ? 0 : ci0), 0, (unsigned)min((long long)0xfffffff0ll, (long long)len_in * ldx * 2));
  const unsigned pdy = (unsigned)(ldy * 2), pxx = (unsigned)(ldx * 2);
  auto stage = [&](int t0, int buf) {
    unsigned char* base = smem + (size_t)buf * buf_bytes;
    // dy: 128 rows x 8 chunks; one wave-instruction = 8 rows
    for (int g = wave; g < R / 8; g += NTHR / 64) {
      const int row = 8 * g + (lane >> 3), pos = lane & 7;
      const int t = t0 + row;
      const int ch = pos ^ (((row >> 1) & 1) << 2);
      untracked_dma16(rdy, t < t_end ? (unsigned)t * pdy + (unsigned)(ch << 4) : 0xffffff00u, base + g * 1024);
    }
    // x: rows_x_pad rows x 16 chunks; one wave-instruction = 4 rows
    const int tin0 = t0 - p.pad;
    for (int g = wave; g < rows_x_pad / 4; g += NTHR / 64) {
      const int row = 4 * g + (lane >> 4), pos = lane & 15;
      const int ch = pos ^ ((row & 3) << 2);
      int tin = tin0 + row, cch = ch;                     // input row and 16-byte chunk inside the 128-channel block
      if (p.win) { cch = (ci0 >> 3) + ch; tin = p.xstride * (t0 + row) - p.pad + (cch >> 3); cch &= 7; }
      const bool ok = (row < rows_x) && (tin >= 0) && (tin < len_in);
      untracked_dma16(rxx, ok ? (unsigned)tin * pxx + (unsigned)(cch << 4) : 0xffffff00u, base + DY_BYTES + g * 1024);
    }
  };

  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  // Per-lane fragment offsets, computed ONCE: the k-step advances rows by 16 (a multiple of 4), so the
  // swizzle term of a lane depends only on the tap.  Inside the loop every transposed read is then
  // "base + lane offset + compile-time immediate" -- no address arithmetic between the MFMAs.
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3, thh = tg >> 1;
  const int coly = wm * 32 + 16 * (tg & 1) + 4 * tp, colx = wn * 32 + 16 * (tg & 1) + 4 * tp;
  const int lrow = 8 * thh + tq;
  const int dyoff = lrow * DYB + (((coly >> 3) ^ (((lrow >> 1) & 1) << 2)) << 4) + (coly & 7) * 2;
  int xoff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int rj = lrow + j * p.dil;
    xoff[j] = rj * XB + (((colx >> 3) ^ ((rj & 3) << 2)) << 4) + (colx & 7) * 2;
  }
  auto tr2 = [&](const unsigned char* base, int imm, int step4) -> bf16x8 {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + imm));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + imm + step4));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  // Window mode runs THREE stage buffers (3 x 48 KiB): a tile is 8-16 MFMAs per wave, far shorter than the 2-3 us a tile takes
  // to arrive, so one tile of prefetch leaves the kernel waiting on latency (measured: 176 us per launch = 36 tiles x 2.5 us x
  // two rounds of workgroups).  Its DMA count per wave and tile is a constant (2 dy + 4 x pieces), so the counted wait can leave
  // the newest tile in flight.  The other modes keep two buffers (their tile size varies with taps and dilation).
  const int nbuf = p.win ? 3 : 2;
  stage(t_begin, 0);
  if (p.win && t_begin + R < t_end) { stage(t_begin + R, 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int it = 0;
  for (int t0 = t_begin; t0 < t_end; t0 += R, ++it) {
    const int buf = it % nbuf;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // tile `it` has landed for every wave (each waited at the end of the previous
                                           // iteration); the buffer read in the previous iteration is free
    const int ahead = nbuf - 1;
    const bool more = t0 + ahead * R < t_end && !((WABL & 1) && it > 0);
    if (more) stage(t0 + ahead * R, (it + ahead) % nbuf);
    const unsigned char* dyt = smem + (size_t)buf * buf_bytes + dyoff;
    const unsigned char* xbase = smem + (size_t)buf * buf_bytes + DY_BYTES;
    bf16x8 a0, b0;
    if (WABL & 2) { a0 = tr2(dyt, 0, 4 * DYB); b0 = tr2(xbase + xoff[0], 0, 4 * XB); }
#pragma unroll
    for (int k0 = 0; k0 < R; k0 += 16) {
      bf16x8 a = (WABL & 2) ? a0 : tr2(dyt, k0 * DYB, 4 * DYB);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (j < ntaps) {
          bf16x8 bfrag = (WABL & 2) ? b0 : tr2(xbase + xoff[j], k0 * XB, 4 * XB);
          if (WABL & 4) { asm volatile("" :: "v"(a), "v"(bfrag)); continue; }
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag, acc[j], 0, 0, 0);
        } else if (j == ntaps && bias_plane) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, acc[j], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next tile must have landed; in window mode the one after it (6 instructions per wave) may still be in flight
    if (p.win && more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const int planes = ntaps + 1;
  float* out = p.slab + ((size_t)cgl * nblk + blk) * (size_t)planes * CB * CIB;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (j >= planes) break;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      const int col = wn * 32 + r;
      out[((size_t)j * CB + row) * CIB + col] = acc[j][e];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Shifted-fragment variant for the dilated 128-channel convs (bf16, stride 1, <= 9 taps at distance 1 in the
// row domain -- natively, or after the dilation-class decomposition of conv_gemm_dma_kernel: rows t = cls (mod
// dilation) form `rs` independent dense problems).  The x operand of tap s is the x operand of tap 0 moved down
// s rows, and an MFMA B fragment holds 8 consecutive rows per lane, so all taps of a k-step are windows of ONE
// 16-row register window W = [P | Q] (P: rows 16 k0 + 8 hh + 0..7, Q: the next 8):
//     even s: B_s = dwords s/2 .. s/2+3 of W (register renaming, free)
//     odd  s: B_s[i] = v_alignbit(W[(s+1)/2 + i], W[(s-1)/2 + i], 16)
// Per k-step a wave reads 3 fragments from LDS (dy^T, P, Q) for up to 9 (+1 bias) MFMAs, where the per-tap
// kernel above reads one per MFMA -- that kernel is LDS-bandwidth bound.  All taps are handled in one launch
// (accumulators: 10 planes x 16 registers), and a workgroup walks a contiguous range of 128-row tiles across
// (batch, class) items, so the number of partial slabs does not grow with the number of classes.
struct ShiftArgs {
  const void* x; const void* dy; float* slab; const int* lens_in;
  long long x_bs, dy_bs;
  int ldx, ldy;
  int B, Tin, Tout, pad, rs;
  int tiles_per_item, tiles_per_wg, n_chunks, nblk_ci, nblk_co, with_bias;
};

constexpr int SH_R = 128, SH_XROWS = SH_R + 8, SH_DY = SH_R * 128, SH_X = SH_XROWS * 256, SH_STAGE = SH_DY + SH_X;

template <int NTAPS>
__global__ __launch_bounds__(512) void conv_wgrad_shift_kernel(ShiftArgs p, const __bf16* __restrict__ zero_page) {
  typedef __bf16 T;
  constexpr int CB = 64, CIB = 128, WNC = 4, NTHR = 512, DYB = CB * 2, XB = CIB * 2, PLANES = NTAPS + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNC, wn = wave % WNC;
  const int r = lane & 31, hh = lane >> 5;

  const int nblk = p.nblk_ci * p.nblk_co;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int blk = q % nblk;
  const int cgl = (q / nblk) * 8 + xcd;
  if (cgl >= p.n_chunks) return;
  const int co0 = (blk / p.nblk_ci) * CB, ci0 = (blk % p.nblk_ci) * CIB;
  const int ntiles = p.tiles_per_item * p.B * p.rs;
  const int tile_begin = cgl * p.tiles_per_wg;
  const int tile_end = min(ntiles, tile_begin + p.tiles_per_wg);
  const int rs = p.rs;
  const bool bias_plane = p.with_bias && (ci0 == 0);

  f32x16 acc[PLANES];
#pragma unroll
  for (int j = 0; j < PLANES; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  auto stage = [&](int tile, int buf) {
    const int item = tile / p.tiles_per_item;
    const int b = item / rs, cls = item - b * rs;
    const int t0 = (tile - item * p.tiles_per_item) * SH_R;
    const int Tc = (p.Tout - cls + rs - 1) / rs;
    const T* xg = reinterpret_cast<const T*>(p.x) + (long long)b * p.x_bs + (long long)cls * p.ldx + ci0;
    const T* dyg = reinterpret_cast<const T*>(p.dy) + (long long)b * p.dy_bs + (long long)cls * p.ldy + co0;
    const long long ldx = (long long)p.ldx * rs, ldy = (long long)p.ldy * rs;
    const int len_full = p.lens_in ? min(scalar_load_i32(p.lens_in + b), p.Tin) : p.Tin;
    const int len_in = max(0, (len_full - cls + rs - 1) / rs);
    unsigned char* base = smem + (size_t)buf * SH_STAGE;
    // dy: 128 rows x 8 chunks; one wave-instruction = 8 rows
#pragma unroll
    for (int g = wave; g < SH_R / 8; g += NTHR / 64) {
      const int row = 8 * g + (lane >> 3), pos = lane & 7;
      const int t = t0 + row;
      const int ch = pos ^ (((row >> 1) & 1) << 2);
      wg_dma16((t < Tc) ? dyg + (long long)t * ldy + ch * 8 : zero_page + pos * 8, base + g * 1024);
    }
    // x: 136 rows x 16 chunks; one wave-instruction = 4 rows
    const int tin0 = t0 - p.pad;
    for (int g = wave; g < SH_XROWS / 4; g += NTHR / 64) {
      const int row = 4 * g + (lane >> 4), pos = lane & 15;
      const int tin = tin0 + row;
      const bool ok = (row < SH_R + NTAPS - 1) && (tin >= 0) && (tin < len_in);
      const int ch = pos ^ ((row & 3) << 2);
      wg_dma16(ok ? xg + (long long)tin * ldx + ch * 8 : zero_page + pos * 8, base + SH_DY + g * 1024);
    }
  };

  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  // per-lane fragment offsets: the k-step advances rows by 16, which keeps both swizzle terms
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3, thh = tg >> 1;
  const int coly = wm * 32 + 16 * (tg & 1) + 4 * tp, colx = wn * 32 + 16 * (tg & 1) + 4 * tp;
  const int lrow = 8 * thh + tq;
  const int dyoff = lrow * DYB + (((coly >> 3) ^ (((lrow >> 1) & 1) << 2)) << 4) + (coly & 7) * 2;
  const int xoff = SH_DY + lrow * XB + (((colx >> 3) ^ ((lrow & 3) << 2)) << 4) + (colx & 7) * 2;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // The three fragments of k-step k0+1 are requested (asm, so that the request stays where it is written) before the
  // MFMAs of k-step k0; `s_waitcnt lgkmcnt(6)` then retires exactly the older six reads (LDS returns in order).  The
  // fragments are operands of the wait so that nothing that uses them can be scheduled above it.
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  struct Frags { u32x2 a0, a1, p0, p1, q0, q1; };
  auto request = [&](Frags& f, unsigned base, int k0) {
    const unsigned pa = base + dyoff + k0 * 16 * DYB, px = base + xoff + k0 * 16 * XB;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.a0) : "v"(pa));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.a1) : "v"(pa), "n"(4 * DYB));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.p0) : "v"(px));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.p1) : "v"(px), "n"(4 * XB));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.q0) : "v"(px), "n"(8 * XB));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.q1) : "v"(px), "n"(12 * XB));
  };

  if (tile_begin < tile_end) stage(tile_begin, 0);
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // this tile has landed for every wave; the other buffer is free
    if (tile + 1 < tile_end) stage(tile + 1, buf ^ 1);
    const unsigned base = lds0 + (unsigned)buf * SH_STAGE;
    Frags fr[2];
    request(fr[0], base, 0);
#pragma unroll
    for (int k0 = 0; k0 < SH_R / 16; ++k0) {
      Frags& f = fr[k0 & 1];
      if (k0 + 1 < SH_R / 16) {
        request(fr[(k0 + 1) & 1], base, k0 + 1);
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.p0), "+v"(f.p1), "+v"(f.q0), "+v"(f.q1) :: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.p0), "+v"(f.p1), "+v"(f.q0), "+v"(f.q1) :: "memory");
      }
      const u32x4 aw = {f.a0[0], f.a0[1], f.a1[0], f.a1[1]};
      const bf16x8 a = __builtin_bit_cast(bf16x8, aw);
      const unsigned w[8] = {f.p0[0], f.p0[1], f.p1[0], f.p1[1], f.q0[0], f.q0[1], f.q1[0], f.q1[1]};
#pragma unroll
      for (int s = 0; s < NTAPS; ++s) {
        const int m = s >> 1;
        u32x4 bw;
        if (s & 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) bw[i] = __builtin_amdgcn_alignbit(w[m + i + 1], w[m + i], 16);
        } else {
          bw = u32x4{w[m], w[m + 1], w[m + 2], w[m + 3]};
        }
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, bw), acc[s], 0, 0, 0);
      }
      // bias gradient: the four waves of a row share the k-steps (each column block of the bias plane then holds a
      // PARTIAL sum; the reduce kernel adds columns 0, 32, 64, 96)
      if (bias_plane && (k0 & 3) == wn) acc[NTAPS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, acc[NTAPS], 0, 0, 0);
    }
  }
  float* out = p.slab + ((size_t)cgl * nblk + blk) * (size_t)PLANES * CB * CIB;
#pragma unroll
  for (int j = 0; j < PLANES; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      out[((size_t)j * CB + row) * CIB + wn * 32 + r] = acc[j][e];
    }
}

// plan of the shifted-fragment variant; false = not applicable
struct ShiftPlan { int rs, pad, tiles_per_item, tiles_per_wg, n_chunks, nblk_co, nblk_ci; };
static bool wgrad_shift_plan(const smt_conv_desc* d, ShiftPlan* pl) {
  static const bool off = getenv("SMT_WGRAD_NO_SHIFT") != nullptr;
  if (off || d->dtype != SMT_BF16 || d->stride != 1 || d->out_stride != 1 || d->out_offset != 0 || !d->zero_page ||
      d->c_in % 128 != 0 || d->c_out % 64 != 0 || d->taps < 3 || d->taps > 9 || !(d->taps & 1) || d->t_in != d->t_out)
    return false;
  int rs = 1, pad = d->padding;
  if (d->dilation > 1) {
    static const int min_rows = getenv("SMT_CLASS_MIN_ROWS") ? atoi(getenv("SMT_CLASS_MIN_ROWS")) : 128;
    if (d->padding % d->dilation != 0 || d->t_out / d->dilation < min_rows) return false;
    rs = d->dilation; pad = d->padding / d->dilation;
  }
  if (pad < 0 || pad > d->taps - 1) return false;
  const long long tc_max = (d->t_out + rs - 1) / rs;
  const int tpi = (int)((tc_max + SH_R - 1) / SH_R);
  const long long ntiles = (long long)tpi * d->batch * rs;
  static const int min_tiles = getenv("SMT_SHIFT_MIN_TILES") ? atoi(getenv("SMT_SHIFT_MIN_TILES")) : 256;
  if (ntiles < min_tiles) return false;                 // small levels: the per-tap kernel wastes less
  const int nco = d->c_out / 64, nci = d->c_in / 128;
  // one workgroup per CU, but at least `min_tpw` tiles per workgroup: every workgroup leaves a (taps + 1) x 32 KiB slab
  // that the reduce kernel reads back, which at the small levels would rival the operand traffic
  static const int min_tpw = getenv("SMT_SHIFT_MIN_TPW") ? atoi(getenv("SMT_SHIFT_MIN_TPW")) : 2;   // measured: 66.9 -> 66.3 ms/step
  const long long chunks_target =
      std::max<long long>(8, std::min<long long>(256 / (nco * nci), ntiles / std::max(1, min_tpw)));
  const int tpw = (int)((ntiles + chunks_target - 1) / chunks_target);
  pl->rs = rs; pl->pad = pad; pl->tiles_per_item = tpi; pl->tiles_per_wg = tpw;
  pl->n_chunks = (int)((ntiles + tpw - 1) / tpw); pl->nblk_co = nco; pl->nblk_ci = nci;
  return true;
}

template <int NTAPS>
static void launch_shift(const ShiftArgs& a, const void* zero_page, dim3 grid, hipStream_t stream) {
  (void)hipFuncSetAttribute((const void*)conv_wgrad_shift_kernel<NTAPS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  conv_wgrad_shift_kernel<NTAPS><<<grid, 512, 2 * SH_STAGE, stream>>>(a, (const __bf16*)zero_page);
}

// ---- fixed-order reduction of the partial slabs -------------------------------------------------------------------
// One job = one weight (+ bias) gradient: slab[chunk][blk][plane][64 co][cib ci] -> dw (torch layout), db.  Jobs travel BY
// VALUE in the kernel arguments (no table upload, capturable in a hipGraph); up to WR_MAXJ jobs share one launch, so a
// GatedHiFi block's ten weight gradients cost one reduce launch instead of ten (smt_wgrad_reduce_defer / _flush).
struct WreduceJob {
  const float* slab; float* dw; float* db;
  long long so, si, sj;
  int n_chunks, nblk, nblk_ci, planes, taps, Cin, Cout, cib, bias_cols;
  int vsplit;                      // > 0: window mode -- column ci of the slab is input channel ci % vsplit of tap ci / vsplit
  int block0;                      // first workgroup of this job inside the launch
  int wblocks;                     // workgroups of the weight part (the bias part follows)
  signed char jmap[16];
};
constexpr int WR_MAXJ = 16;
struct WreduceBatch { int n_jobs, total_blocks; WreduceJob job[WR_MAXJ]; };

// Weight part: a workgroup owns 64 float4 outputs (256 consecutive ci of one (plane, co) row ... up to row ends) x 4 chunk
// slices: thread (o, s) sums chunks s, s+4, ... in index order (four independent 16-byte loads in flight; eight measured the same), the four slice
// sums are then added in slice order -- a fixed summation tree: bitwise reproducible.  (Round 3: a wave now reads ONE
// 1 KiB run of one slab per load instruction; the first layout -- 32 outputs x 8 slices, two 512-byte runs of two slabs per
// instruction -- read the slabs at 2.5-3 TB/s.)  Bias part: one thread per output channel and slice (32 x 8), columns 0, 32, ..
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(WreduceBatch bt) {
  __shared__ f32x4 part[4][64];
  int j = 0;
#pragma unroll 1
  for (int k = 1; k < bt.n_jobs; ++k)
    if ((int)blockIdx.x >= bt.job[k].block0) j = k;
  const WreduceJob& p = bt.job[j];
  const int lb = blockIdx.x - p.block0;
  const size_t blk_elems = (size_t)p.planes * 64 * p.cib;
  const size_t cstride = (size_t)p.nblk * blk_elems;
  if (lb < p.wblocks) {
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int cin4 = p.Cin >> 2;
    const long long total4 = (long long)p.taps * p.Cout * cin4;
    const long long e4 = (long long)lb * 64 + o;
    const bool live = e4 < total4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int ci = 0, co = 0, plane = 0;
    if (live) {
      ci = (int)(e4 % cin4) * 4; co = (int)((e4 / cin4) % p.Cout); plane = (int)(e4 / ((long long)cin4 * p.Cout));
      const int blk = (co / 64) * p.nblk_ci + (ci / p.cib);
      const float* src = p.slab + (size_t)blk * blk_elems + ((size_t)plane * 64 + (co % 64)) * p.cib + (ci % p.cib);
      int c = sl;
      for (; c + 12 < p.n_chunks; c += 16) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (size_t)c * cstride);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (size_t)(c + 4) * cstride);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (size_t)(c + 8) * cstride);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + (size_t)(c + 12) * cstride);
        s += v0; s += v1; s += v2; s += v3;
      }
      for (; c < p.n_chunks; c += 4) s += *reinterpret_cast<const f32x4*>(src + (size_t)c * cstride);
    }
    part[sl][o] = s;
    __syncthreads();
    if (sl == 0 && live) {
      f32x4 t = part[0][o];
#pragma unroll
      for (int k = 1; k < 4; ++k) t += part[k][o];
      float* dst = p.vsplit > 0 ? p.dw + co * p.so + (ci % p.vsplit) * p.si + p.jmap[ci / p.vsplit] * p.sj
                                : p.dw + co * p.so + ci * p.si + p.jmap[plane] * p.sj;
      dst[0] = t[0]; dst[p.si] = t[1]; dst[2 * p.si] = t[2]; dst[3 * p.si] = t[3];
    }
  } else {
    float* part1 = reinterpret_cast<float*>(&part[0][0]);      // [8][32]
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int co = (lb - p.wblocks) * 32 + o;
    const bool live = co < p.Cout && p.db != nullptr;
    float s = 0.f;
    if (live) {
      // bias plane: column 0 of every ci block holds db[co] (or partial sums in columns 0, 32, .. of the first block)
      const int blk = (co / 64) * p.nblk_ci;
      const float* src = p.slab + (size_t)blk * blk_elems + ((size_t)p.taps * 64 + (co % 64)) * p.cib;
      for (int c = sl; c < p.n_chunks; c += 8) {
        const float* q = src + (size_t)c * cstride;
        float part_sum = q[0];
        for (int k = 1; k < p.bias_cols; ++k) part_sum += q[32 * k];
        s += part_sum;
      }
    }
    part1[sl * 32 + o] = s;
    __syncthreads();
    if (sl == 0 && live) {
      float t = part1[o];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += part1[k * 32 + o];
      p.db[co] = t;
    }
  }
}

// input channels per workgroup: 128 (8 waves, two per SIMD) while the accumulator planes fit in the
// 256-register budget of that occupancy, else 64 (4 waves, one per SIMD, 512 registers)
static int wgrad_cib(const smt_conv_desc* d) {
  if (d->dtype != SMT_BF16) return 64;
  return (d->c_in > 64 && d->stride == 1) ? 128 : 64;
}

// dilation-class decomposition applies to the LDS-DMA variant only (bf16, 128-channel blocks, stride 1)
static int wgrad_rs(const smt_conv_desc* d) {
  const bool dma_shape = d->dtype == SMT_BF16 && d->stride == 1 && d->c_in % 128 == 0 && d->c_out % 64 == 0 &&
                         d->out_stride == 1 && d->out_offset == 0 && d->zero_page != nullptr;
  // Measured: for the weight gradient the strided class rows cost more than the smaller halo saves
  // (2.04 vs 1.79 ms at k = 9, dilation 27, T/2 level), so the decomposition stays switched off here.
  constexpr bool kUseClasses = false;
  return (kUseClasses && dma_shape && d->dilation >= 8 && d->taps > 1 && d->padding % d->dilation == 0 &&
          d->t_in == d->t_out) ? d->dilation : 1;
}

static void wgrad_plan(const smt_conv_desc* d, int* rows_per_chunk, int* chunks_per_batch, int* nblk_co,
                       int* nblk_ci, int* planes) {
  const int cib = wgrad_cib(d);
  const int rs = wgrad_rs(d);
  *nblk_co = (d->c_out + 63) / 64;
  *nblk_ci = (d->c_in + cib - 1) / cib;
  const int R = d->dtype == SMT_BF16 ? 128 : 64;
  const long long tc = (d->t_out + rs - 1) / rs;                 // rows per (batch, class) item
  long long total_rows = (long long)d->batch * rs * tc;
  long long target_wgs = 512;   // two rounds of one workgroup per CU: keeps the partial slabs small
  long long rows = (total_rows * (*nblk_co) * (*nblk_ci) + target_wgs - 1) / target_wgs;
  rows = std::max<long long>(R, (rows + R - 1) / R * R);
  rows = std::min<long long>(rows, (tc + R - 1) / R * R);
  *rows_per_chunk = (int)rows;
  *chunks_per_batch = (int)((tc + rows - 1) / rows);
  *planes = d->taps + 1;
}

static thread_local bool g_reduce_defer = false;
static thread_local WreduceBatch g_reduce_batch = {0, 0, {}};

static int reduce_flush(hipStream_t stream) {
  if (g_reduce_batch.n_jobs > 0 && g_reduce_batch.total_blocks > 0) {
    conv_wgrad_reduce_kernel<<<(unsigned)g_reduce_batch.total_blocks, 256, 0, stream>>>(g_reduce_batch);
    g_reduce_batch.n_jobs = 0; g_reduce_batch.total_blocks = 0;
    SMT_CHECK_LAUNCH("conv_wgrad_reduce");
  }
  g_reduce_batch.n_jobs = 0; g_reduce_batch.total_blocks = 0;
  return 0;
}

int launch_wgrad_reduce(const float* slab, float* dw, float* db, int n_chunks, int nblk_co, int nblk_ci, int taps,
                        int c_in, int c_out, int cib, long long so, long long si, long long sj, const int* jmap,
                        hipStream_t stream, int bias_cols, int vsplit) {
  SMT_CHECK_ARG(c_in % 4 == 0 && cib % 4 == 0 && taps <= 16, "conv_wgrad_reduce: c_in and the ci block must be multiples of 4");
  if (g_reduce_batch.n_jobs == WR_MAXJ) {
    int rc = reduce_flush(stream);
    if (rc) return rc;
  }
  WreduceJob& r = g_reduce_batch.job[g_reduce_batch.n_jobs];
  r.bias_cols = bias_cols;
  r.vsplit = vsplit;
  r.slab = slab; r.dw = dw; r.db = db;
  r.n_chunks = n_chunks; r.nblk = nblk_co * nblk_ci; r.nblk_ci = nblk_ci; r.planes = taps + 1; r.taps = taps;
  r.Cin = c_in; r.Cout = c_out; r.cib = cib;
  r.so = so; r.si = si; r.sj = sj;
  const int n_map = vsplit > 0 ? c_in / vsplit : taps;      // window mode: one entry per real tap
  for (int t = 0; t < 16; ++t) r.jmap[t] = (signed char)(t < n_map ? jmap[t] : 0);
  const long long total4 = (long long)taps * c_out * (c_in / 4);
  r.wblocks = (int)((total4 + 63) / 64);
  r.block0 = g_reduce_batch.total_blocks;
  g_reduce_batch.total_blocks += r.wblocks + (db ? (c_out + 31) / 32 : 0);
  g_reduce_batch.n_jobs += 1;
  return g_reduce_defer ? 0 : reduce_flush(stream);
}

}  // namespace smt

using namespace smt;

extern "C" int smt_wgrad_reduce_defer(int on, smt_stream_t stream_) {
  // on = 1: the weight-gradient calls that follow on this thread (smt_conv1d_wgrad, smt_conv1x1_bwd, smt_conv_k1_bwd,
  // smt_conv_gate_bwd) queue their slab reductions instead of launching them; every call must then be given a workspace
  // of its own, which has to stay untouched until the flush.  on = 0: launch what is queued (one launch per 16 jobs).
  g_reduce_defer = on != 0;
  return on ? 0 : reduce_flush((hipStream_t)stream_);
}

static bool wgrad_window_desc(const smt_conv_desc* d, smt_conv_desc* v);

static size_t wgrad_group_ws(const smt_conv_desc* d) {
  int rpc, cpb, nco, nci, planes;
  wgrad_plan(d, &rpc, &cpb, &nco, &nci, &planes);
  return (size_t)d->batch * wgrad_rs(d) * cpb * nco * nci * planes * 64 * wgrad_cib(d) * sizeof(float);
}

// Tap groups: more than 5 taps would need more accumulator registers than two waves per SIMD allow,
// so wide kernels are processed as consecutive groups of <= 5 taps (a group of taps j0.. is the same
// convolution with padding reduced by j0*dilation).
constexpr int WG_GROUP = 5;
static smt_conv_desc wgrad_group_desc(const smt_conv_desc* d, int j0, int n) {
  smt_conv_desc g = *d;
  g.taps = n;
  g.padding = d->padding - j0 * d->dilation;
  return g;
}

extern "C" size_t smt_conv1d_wgrad_workspace_bytes(const smt_conv_desc* d) {
  size_t best = 0;
  ShiftPlan pl;
  if (wgrad_shift_plan(d, &pl))
    best = (size_t)pl.n_chunks * pl.nblk_co * pl.nblk_ci * (d->taps + 1) * 64 * 128 * sizeof(float);
  smt_conv_desc wv;
  if (wgrad_window_desc(d, &wv)) return std::max(best, align_up(wgrad_group_ws(&wv), 256));
  // tap groups get consecutive regions (their reductions may be deferred: smt_wgrad_reduce_defer)
  size_t groups = 0;
  for (int j0 = 0; j0 < d->taps; j0 += WG_GROUP) {
    smt_conv_desc g = wgrad_group_desc(d, j0, std::min(WG_GROUP, d->taps - j0));
    groups += align_up(wgrad_group_ws(&g), 256);
  }
  return std::max(best, groups);
}

template <typename T, int CIB, bool STRIDED>
static int launch_wgrad(const WgradArgs& a, dim3 grid, size_t lds, int planes, hipStream_t stream) {
#define SMT_WG_CASE(NT)                                                                                  \
  case NT:                                                                                               \
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<T, NT, CIB, STRIDED>,                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                   \
    conv_wgrad_kernel<T, NT, CIB, STRIDED><<<grid, CIB * 4, lds, stream>>>(a);                           \
    break;
  switch (planes) {
    SMT_WG_CASE(2) SMT_WG_CASE(3) SMT_WG_CASE(4) SMT_WG_CASE(5) SMT_WG_CASE(6)
    default:
      set_error("conv_wgrad: unsupported tap count %d", planes - 1);
      return 1;
  }
#undef SMT_WG_CASE
  SMT_CHECK_LAUNCH("conv_wgrad");
  return 0;
}

static int wgrad_group(const smt_conv_desc* d, float* dweight, int64_t stride_out, int64_t stride_in,
                       int64_t stride_tap, const int* tap_map, float* dbias, void* workspace,
                       size_t workspace_bytes, hipStream_t stream) {
  SMT_CHECK_ARG(d && d->x && d->y && dweight && tap_map && workspace, "smt_conv1d_wgrad: null pointer");
  SMT_CHECK_ARG(d->dtype == SMT_BF16 || d->dtype == SMT_F32, "smt_conv1d_wgrad: bad dtype");
  const int epv = d->dtype == SMT_BF16 ? 8 : 4;
  SMT_CHECK_ARG(d->c_in % epv == 0 && d->c_out % epv == 0 && d->ld_x % epv == 0 && d->ld_y % epv == 0,
                "smt_conv1d_wgrad: channels / pitches must keep 16-byte alignment");
  SMT_CHECK_ARG(d->taps >= 1 && d->taps <= WG_GROUP, "smt_conv1d_wgrad: internal tap group too large");
  SMT_CHECK_ARG(workspace_bytes >= wgrad_group_ws(d), "smt_conv1d_wgrad: workspace too small");
  int rpc, cpb, nco, nci, planes;
  wgrad_plan(d, &rpc, &cpb, &nco, &nci, &planes);
  const int cib = wgrad_cib(d);
  WgradArgs a;
  a.x = d->x; a.dy = d->y; a.slab = (float*)workspace; a.lens_in = d->lens_in;
  a.x_bs = d->bs_x; a.dy_bs = d->bs_y; a.ldx = d->ld_x; a.ldy = d->ld_y;
  a.B = d->batch; a.Tin = d->t_in; a.Tout = d->t_out; a.Ty = d->t_y; a.Cin = d->c_in; a.Cout = d->c_out;
  a.taps = d->taps; a.stride = d->stride; a.dil = d->dilation; a.pad = d->padding;
  a.out_stride = d->out_stride; a.out_offset = d->out_offset;
  a.rows_per_chunk = rpc; a.chunks_per_batch = cpb; a.nblk_ci = nci; a.nblk_co = nco; a.with_bias = dbias ? 1 : 0;
  const int rs = wgrad_rs(d);
  a.rs = rs; a.win = 0; a.xstride = 1;
  if (rs > 1) { a.pad = d->padding / rs; a.dil = 1; }
  if (d->batch > 0 && d->t_out > 0) {
    const int n_chunks = d->batch * rs * cpb;
    dim3 grid((unsigned)(8 * ((n_chunks + 7) / 8) * nco * nci));
    const bool bf = d->dtype == SMT_BF16;
    const int R = bf ? 128 : 64;
    const int rows_x = (R - 1) * d->stride + (d->taps - 1) * a.dil + 1;
    const size_t lds = bf ? ((size_t)R * WTr<__bf16>::pitch(64) + (size_t)rows_x * WTr<__bf16>::pitch(cib)) * 2
                          : ((size_t)R * WTr<float>::pitch(64) + (size_t)rows_x * WTr<float>::pitch(cib)) * 4;
    SMT_CHECK_ARG(lds <= 160 * 1024, "conv_wgrad: tile needs %zu B of LDS", lds);
    int rc;
    const bool strided = d->stride > 1;
    const int rows_xp = (rows_x + 3) & ~3;
    const size_t lds_dma = 2 * ((size_t)128 * 128 + (size_t)rows_xp * 256);
    const bool dma = bf && cib == 128 && !strided && d->zero_page && d->c_in % 128 == 0 && d->c_out % 64 == 0 &&
                     d->out_stride == 1 && d->out_offset == 0 && lds_dma <= 160 * 1024;
    SMT_CHECK_ARG(rs == 1 || dma, "conv_wgrad: dilation classes need the LDS-DMA variant");
    if (dma) {
#define SMT_WGD_CASE(NT)                                                                                 \
  case NT:                                                                                               \
    (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<NT>,                                    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                   \
    conv_wgrad_dma_kernel<NT><<<grid, 512, lds_dma, stream>>>(a, (const __bf16*)d->zero_page);           \
    break;
      switch (planes) {
        SMT_WGD_CASE(2) SMT_WGD_CASE(3) SMT_WGD_CASE(4) SMT_WGD_CASE(5) SMT_WGD_CASE(6)
        default: set_error("conv_wgrad_dma: unsupported tap count"); return 1;
      }
#undef SMT_WGD_CASE
      SMT_CHECK_LAUNCH("conv_wgrad_dma");
      rc = 0;
    } else
    if (bf && cib == 128) rc = launch_wgrad<__bf16, 128, false>(a, grid, lds, planes, stream);
    else if (bf) rc = strided ? launch_wgrad<__bf16, 64, true>(a, grid, lds, planes, stream)
                              : launch_wgrad<__bf16, 64, false>(a, grid, lds, planes, stream);
    else rc = strided ? launch_wgrad<float, 64, true>(a, grid, lds, planes, stream)
                      : launch_wgrad<float, 64, false>(a, grid, lds, planes, stream);
    if (rc) return rc;
  }
  const int n_chunks = (d->batch == 0 || d->t_out == 0) ? 0 : d->batch * rs * cpb;
  return launch_wgrad_reduce((const float*)workspace, dweight, dbias, n_chunks, nco, nci, d->taps, d->c_in, d->c_out,
                             cib, stride_out, stride_in, stride_tap, tap_map, stream);
}

// Window mode of the LDS-DMA kernel (round 3): the weight gradients of the k = 4 / stride-2 resampling convs of a
// 64-channel level and of the two-tap phases of their transposes.  dW[co][ci][j] = sum_t dy[t][co] x[s t - pad + j][ci] is a
// 1 x 1 weight gradient over 64 * taps "virtual" channels whose 64-wide groups are consecutive ROWS of x; the kernel gathers
// them chunk by chunk while staging, the reducer scatters virtual columns back to (ci, tap).  MEASURED SLOWER than the
// register-staged generic kernel on the layers it was built for (k2/out_stride 2, 64 -> 128: 221 vs 176 us per launch;
// 64 -> 64: 89 vs 73; k4/stride 2: 138 vs 116 -- both forms already move their bytes at 5-6 TB/s: each phase of a transpose
// re-reads x, each 64-channel output block re-reads it again), so it is OFF unless SMT_WGRAD_WINDOW=1 (read per call).
static bool wgrad_window_desc(const smt_conv_desc* d, smt_conv_desc* v) {
  const char* env = getenv("SMT_WGRAD_WINDOW");
  const bool off = !env || atoi(env) == 0;
  const bool shape = (d->stride == 2 && d->taps == 4 && d->out_stride == 1 && d->out_offset == 0) ||
                     (d->stride == 1 && d->taps == 2 && d->out_stride >= 1);
  if (off || d->dtype != SMT_BF16 || d->c_in != 64 || d->c_out % 64 != 0 || !d->zero_page || d->dilation != 1 || !shape ||
      d->ld_x % 8 != 0 || d->ld_y % 8 != 0 || d->t_out < 1)
    return false;
  *v = *d;
  v->c_in = 64 * d->taps; v->taps = 1; v->stride = 1; v->dilation = 1; v->padding = 0;
  v->out_stride = 1; v->out_offset = 0;                   // folded into the dy pointer and pitch by the launcher
  return true;
}

static int wgrad_window(const smt_conv_desc* d, const smt_conv_desc* v, float* dweight, int64_t stride_out, int64_t stride_in,
                        int64_t stride_tap, const int* tap_map, float* dbias, void* workspace, size_t workspace_bytes,
                        hipStream_t stream) {
  SMT_CHECK_ARG(d->x && d->y && dweight && workspace, "smt_conv1d_wgrad: null pointer");
  SMT_CHECK_ARG(workspace_bytes >= wgrad_group_ws(v), "smt_conv1d_wgrad: workspace too small");
  SMT_CHECK_ARG((long long)(d->t_out - 1) * d->out_stride + d->out_offset < d->t_y, "smt_conv1d_wgrad: output rows out of range");
  int rpc, cpb, nco, nci, planes;
  wgrad_plan(v, &rpc, &cpb, &nco, &nci, &planes);
  WgradArgs a;
  a.x = d->x; a.slab = (float*)workspace; a.lens_in = d->lens_in;
  a.dy = reinterpret_cast<const __bf16*>(d->y) + (long long)d->out_offset * d->ld_y;
  a.x_bs = d->bs_x; a.dy_bs = d->bs_y; a.ldx = d->ld_x; a.ldy = d->ld_y * d->out_stride;
  a.B = d->batch; a.Tin = d->t_in; a.Tout = d->t_out; a.Ty = d->t_y; a.Cin = v->c_in; a.Cout = d->c_out;
  a.taps = 1; a.stride = 1; a.dil = 1; a.pad = d->padding; a.out_stride = 1; a.out_offset = 0;
  a.rows_per_chunk = rpc; a.chunks_per_batch = cpb; a.nblk_ci = nci; a.nblk_co = nco; a.with_bias = dbias ? 1 : 0;
  a.rs = 1; a.win = 1; a.xstride = d->stride;
  const int n_chunks = d->batch * cpb;
  if (d->batch > 0) {
    dim3 grid((unsigned)(8 * ((n_chunks + 7) / 8) * nco * nci));
    const size_t lds_dma = 3 * ((size_t)128 * 128 + (size_t)128 * 256);      // three stage buffers in window mode
    (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    conv_wgrad_dma_kernel<2><<<grid, 512, lds_dma, stream>>>(a, (const __bf16*)d->zero_page);
    SMT_CHECK_LAUNCH("conv_wgrad_dma");
  }
  return launch_wgrad_reduce((const float*)workspace, dweight, dbias, d->batch > 0 ? n_chunks : 0, nco, nci, 1, v->c_in,
                             d->c_out, 128, stride_out, stride_in, stride_tap, tap_map, stream, 1, /*vsplit=*/64);
}

extern "C" const char* smt_conv1d_wgrad_kernel_name(const smt_conv_desc* d) {
  if (!d) return "";
  ShiftPlan pl;
  if (wgrad_shift_plan(d, &pl)) return "conv_wgrad_shift";
  smt_conv_desc wv;
  if (wgrad_window_desc(d, &wv)) return "conv_wgrad_dma";
  const smt_conv_desc g = wgrad_group_desc(d, 0, std::min(WG_GROUP, d->taps));
  const int rows_x = 127 * g.stride + (g.taps - 1) * (wgrad_rs(&g) > 1 ? 1 : g.dilation) + 1;
  const size_t lds_dma = 2 * ((size_t)128 * 128 + (size_t)((rows_x + 3) & ~3) * 256);
  const bool dma = g.dtype == SMT_BF16 && wgrad_cib(&g) == 128 && g.stride == 1 && g.zero_page && g.c_in % 128 == 0 &&
                   g.c_out % 64 == 0 && g.out_stride == 1 && g.out_offset == 0 && lds_dma <= 160 * 1024;
  return dma ? "conv_wgrad_dma" : "conv_wgrad";
}

extern "C" int smt_conv1d_wgrad(const smt_conv_desc* d, float* dweight, int64_t stride_out, int64_t stride_in,
                                int64_t stride_tap, const int* tap_map, float* dbias, void* workspace,
                                size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(d && tap_map, "smt_conv1d_wgrad: null pointer");
  SMT_CHECK_ARG(d->taps >= 1 && d->taps <= 16, "smt_conv1d_wgrad: taps must be in [1, 16]");
  ShiftPlan pl;
  if (wgrad_shift_plan(d, &pl)) {      // all taps in one launch, fragments shifted in registers
    SMT_CHECK_ARG(d->x && d->y && dweight && workspace, "smt_conv1d_wgrad: null pointer");
    SMT_CHECK_ARG(d->ld_x % 8 == 0 && d->ld_y % 8 == 0, "smt_conv1d_wgrad: pitches must keep 16-byte alignment");
    SMT_CHECK_ARG(workspace_bytes >= smt_conv1d_wgrad_workspace_bytes(d), "smt_conv1d_wgrad: workspace too small");
    ShiftArgs a;
    a.x = d->x; a.dy = d->y; a.slab = (float*)workspace; a.lens_in = d->lens_in;
    a.x_bs = d->bs_x; a.dy_bs = d->bs_y; a.ldx = d->ld_x; a.ldy = d->ld_y;
    a.B = d->batch; a.Tin = d->t_in; a.Tout = d->t_out; a.pad = pl.pad; a.rs = pl.rs;
    a.tiles_per_item = pl.tiles_per_item; a.tiles_per_wg = pl.tiles_per_wg; a.n_chunks = pl.n_chunks;
    a.nblk_ci = pl.nblk_ci; a.nblk_co = pl.nblk_co; a.with_bias = dbias ? 1 : 0;
    dim3 grid((unsigned)(8 * ((pl.n_chunks + 7) / 8) * pl.nblk_co * pl.nblk_ci));
    switch (d->taps) {
      case 3: launch_shift<3>(a, d->zero_page, grid, stream); break;
      case 5: launch_shift<5>(a, d->zero_page, grid, stream); break;
      case 7: launch_shift<7>(a, d->zero_page, grid, stream); break;
      default: launch_shift<9>(a, d->zero_page, grid, stream); break;
    }
    SMT_CHECK_LAUNCH("conv_wgrad_shift");
    return launch_wgrad_reduce((const float*)workspace, dweight, dbias, pl.n_chunks, pl.nblk_co, pl.nblk_ci, d->taps,
                               d->c_in, d->c_out, 128, stride_out, stride_in, stride_tap, tap_map, stream, 4);
  }
  SMT_CHECK_ARG(workspace_bytes >= smt_conv1d_wgrad_workspace_bytes(d), "smt_conv1d_wgrad: workspace too small");
  smt_conv_desc wv;
  if (wgrad_window_desc(d, &wv))
    return wgrad_window(d, &wv, dweight, stride_out, stride_in, stride_tap, tap_map, dbias, workspace, workspace_bytes, stream);
  size_t off = 0;
  for (int j0 = 0; j0 < d->taps; j0 += WG_GROUP) {
    smt_conv_desc g = wgrad_group_desc(d, j0, std::min(WG_GROUP, d->taps - j0));
    const size_t need = align_up(wgrad_group_ws(&g), 256);
    int rc = wgrad_group(&g, dweight, stride_out, stride_in, stride_tap, tap_map + j0, j0 == 0 ? dbias : nullptr,
                         (char*)workspace + off, need, stream);
    if (rc) return rc;
    off += need;
  }
  return 0;
}
