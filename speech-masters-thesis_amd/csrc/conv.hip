// Channels-last 1-D convolution as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces F.conv1d / F.conv_transpose1d under MaskedConv1d, MaskedConvTranspose1d, ResLayer and
// GatedHiFiBlock of the reference (models/vqvae/conv.py:5-18, resnet.py:16-36, 184-241) -- forward
// and data-gradient (the data-gradient of a convolution is a convolution with repacked weights).
//
//   y[b, t*os + oo, co] = epi( sum_j sum_ci pro(x)[b, t*stride + j*dil - pad, ci] * w[j][co][ci] )
//
// Activations are [B, T, C] with an explicit row pitch, so channel slices of a wider tensor are
// operands without copies.  One workgroup (4 waves, 2x2) computes BM output rows x BN output
// channels:  the haloed input tile is staged ONCE into LDS (the prologue -- row mask, ReLU,
// counter-based dropout -- is applied while staging, once per element, not once per tap) and every
// tap reads it at a row offset; weight chunks [BN x KC] stream through a double buffer.
//   bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate.   fp32: v_mfma_f32_32x32x2_f32 (exact fp32 fma
//   chains; the parity path).  Both read 16-byte fragments with ds_read_b128 from rows padded by
//   16 B (conflict-free, cdna guide G4).
// The accumulator tile goes back through LDS so that the epilogue (bias, activation gradient, row
// mask, residual) and the stores are fully coalesced 16-byte accesses.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "smt_common.h"
#include "conv_common.h"

namespace smt {

struct ConvArgs {
  const void* x; const void* w; const float* bias; void* y; const void* res; const void* gate_h; void* y_act;
  const int* lens_in; const int* lens_out;
  long long x_bs, y_bs, res_bs, gh_bs, ya_bs;  // batch strides (elements)
  int ldx, ldy, ldr, ldgh, ldya;               // row pitches (elements)
  int B, Tin, Tout, Cin, Cout;          // Tout = output rows PER LAUNCH INDEX t (before os/oo)
  int taps, stride, dil, pad;
  int out_stride, out_offset, Ty;       // output row = t*out_stride + out_offset, Ty rows in y per batch
  int act_out, epi_act;                 // relu+dropout of the OUTPUT (second store) / its derivative as epilogue
  unsigned drop_keys[8]; int site_width; unsigned drop_thresh16; float drop_scale;
  const unsigned* drop_keys_dev; int drop_keys_dev_stride;   // keys in device memory (graph replay): override drop_keys
  int tiles_per_batch;
  int rs;   // row stride of the dilation-class decomposition (LDS-DMA kernel), 1 = off
  const void* x2; const void* w2; const float* bias2; const int* lens_in2; long long x2_bs; int ldx2;   // folded second 1x1 term (conv1x1_fold)
  int dbg;  // ablation switches (SMT_CONV_DBG): 1 no A loads, 2 no W loads, 4 no MFMA, 8 no stores
};

// dropout key of site s of this launch: by value, or from device memory when the caller keeps its keys there
__device__ __forceinline__ unsigned site_key(const ConvArgs& p, int site) {
  return p.drop_keys_dev ? p.drop_keys_dev[(site & 7) * p.drop_keys_dev_stride] : p.drop_keys[site & 7];
}

template <typename T>
__device__ __forceinline__ void mma_step(const T* a, const T* b, f32x16& acc);
template <>
__device__ __forceinline__ void mma_step<__bf16>(const __bf16* a, const __bf16* b, f32x16& acc) {
  bf16x8 av = *reinterpret_cast<const bf16x8*>(a);
  bf16x8 bv = *reinterpret_cast<const bf16x8*>(b);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_step<float>(const float* a, const float* b, f32x16& acc) {
  f32x4 av = *reinterpret_cast<const f32x4*>(a);
  f32x4 bv = *reinterpret_cast<const f32x4*>(b);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
}

// Waves are arranged 2 (rows) x WN (cols): NT = 256 threads -> WN = 2, NT = 512 -> WN = 4 (two waves per
// SIMD, so one wave's staging / address arithmetic hides under the other's MFMAs).
template <typename T, int BN, int NT>
__global__ __launch_bounds__(NT) void conv_gemm_kernel(ConvArgs p) {
  constexpr int EPV = Tr<T>::EPV, CCH = Tr<T>::CCH, KC = Tr<T>::KC, BM = Tr<T>::BM;
  constexpr int MW = BM / 64;           // 32-row tiles per wave (bf16: 2, fp32: 1)
  constexpr int WN = NT / 128;          // wave columns
  constexpr int NW = BN / (32 * WN);    // 32-col tiles per wave
  constexpr int PITCH_W = KC + EPV;
  constexpr int PITCH_C = BN + EPV;
  constexpr int WVEC = BN * KC / EPV;   // 16-byte vectors per weight chunk
  constexpr int WST = WVEC / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, hh = lane >> 5;
  // XCD-aware mapping (cdna guide T1): workgroups b and b+8 share an XCD/L2, so give every XCD a
  // CONTIGUOUS run of row tiles -- neighbouring tiles share their halo rows and the weights in L2.
  const int ntiles = p.tiles_per_batch * p.B;
  const int per_xcd = (ntiles + 7) / 8;
  const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  if (p.dbg & 16) return;
  const int b = tile / p.tiles_per_batch;
  const int t0 = (tile % p.tiles_per_batch) * BM;
  const int n0 = blockIdx.y * BN;

  const int rows_in = (BM - 1) * p.stride + (p.taps - 1) * p.dil + 1;
  const int cch_max = min(p.Cin, CCH);
  const int pitch_a = cch_max + EPV;
  T* lds_a = reinterpret_cast<T*>(smem);
  const size_t a_bytes = align_up((size_t)max(rows_in * pitch_a, BM * PITCH_C) * sizeof(T), 16);
  T* lds_w = reinterpret_cast<T*>(smem + a_bytes);
  T* lds_c = lds_a;

  const T* xg = reinterpret_cast<const T*>(p.x) + (long long)b * p.x_bs;
  const T* wg = reinterpret_cast<const T*>(p.w);
  const int len_in = p.lens_in ? min(p.lens_in[b], p.Tin) : p.Tin;
  const int tin0 = t0 * p.stride - p.pad;

  f32x16 acc[MW][NW];
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int n = 0; n < NW; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][n][e] = 0.f;

  // weight chunk staging: two register sets so that chunk s+2 is in flight while chunk s is consumed
  Vec<T, EPV> wst0[WST], wst1[WST];
  auto w_load = [&](Vec<T, EPV>* wst, int j, int ci0) {  // chunk rows co = n0.., cols ci = ci0..ci0+KC
#pragma unroll
    for (int s = 0; s < WST; ++s) {
      int f = tid + NT * s;
      int co = f / (KC / EPV), cv = f % (KC / EPV);
      int ci = ci0 + cv * EPV;
      Vec<T, EPV> v;
#pragma unroll
      for (int e = 0; e < EPV; ++e) v.v[e] = (T)0.f;
      if (n0 + co < p.Cout && ci < p.Cin && !(p.dbg & 2))
        v = *reinterpret_cast<const Vec<T, EPV>*>(wg + ((size_t)j * p.Cout + n0 + co) * p.Cin + ci);
      wst[s] = v;
    }
  };
  auto w_store = [&](const Vec<T, EPV>* wst, int buf) {
#pragma unroll
    for (int s = 0; s < WST; ++s) {
      int f = tid + NT * s;
      int co = f / (KC / EPV), cv = f % (KC / EPV);
      if (!(p.dbg & 128)) *reinterpret_cast<Vec<T, EPV>*>(lds_w + (size_t)buf * BN * PITCH_W + co * PITCH_W + cv * EPV) = wst[s];
    }
  };

  for (int cc = 0; cc < p.Cin; cc += CCH) {
    const int cch = min(CCH, p.Cin - cc);
    __syncthreads();  // previous chunk's readers are done with lds_a / lds_w
    // ---- stage the haloed input tile (this channel chunk), prologue applied once per element
    const int vpr = cch / EPV;
    const int vshift = 31 - __builtin_clz(vpr);   // vpr is a power of two (checked on the host)
    {
      // batches of UB independent 16-byte loads per thread, THEN the LDS stores: a load->store loop
      // would expose one full memory latency per vector
      constexpr int UB = 8;
      const int total = rows_in * vpr;
      for (int f0 = tid; f0 < total; f0 += NT * UB) {
        Vec<T, EPV> v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int f = f0 + NT * u;
          const int row = f >> vshift, cv = f & (vpr - 1);
          const int tin = tin0 + row;
#pragma unroll
          for (int e = 0; e < EPV; ++e) v[u].v[e] = (T)0.f;
          if (f < total && tin >= 0 && tin < len_in && !(p.dbg & 1))
            v[u] = *reinterpret_cast<const Vec<T, EPV>*>(xg + (long long)tin * p.ldx + cc + cv * EPV);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int f = f0 + NT * u;
          const int row = f >> vshift, cv = f & (vpr - 1);
          if (f < total && !(p.dbg & 64)) *reinterpret_cast<Vec<T, EPV>*>(lds_a + row * pitch_a + cv * EPV) = v[u];
        }
      }
    }
    // ---- K loop: taps x K-chunks.  LDS weight buffers alternate; global loads run two chunks ahead
    const int nkc = (cch + KC - 1) / KC;
    const int nsteps = p.taps * nkc;
    auto compute = [&](int s) {
      const int j = s / nkc, kc = s % nkc;
      const int kvalid = min(KC, cch - kc * KC);
      const T* abase = lds_a + (wm * (BM / 2) * p.stride + j * p.dil + r * p.stride) * pitch_a + kc * KC + hh * EPV;
      const T* bbase = lds_w + (size_t)(s & 1) * BN * PITCH_W + (wn * (BN / WN) + r) * PITCH_W + hh * EPV;
      const int a_step = 32 * p.stride * pitch_a;
      if (p.dbg & 4) return;
      if (kvalid == KC) {  // full chunk: constant trip count so the LDS reads are scheduled ahead of the MFMAs
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2 * EPV) {
#pragma unroll
          for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int n = 0; n < NW; ++n)
              mma_step<T>(abase + i * a_step + kk, bbase + n * 32 * PITCH_W + kk, acc[i][n]);
        }
      } else {
        for (int kk = 0; kk < kvalid; kk += 2 * EPV) {
#pragma unroll
          for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int n = 0; n < NW; ++n)
              mma_step<T>(abase + i * a_step + kk, bbase + n * 32 * PITCH_W + kk, acc[i][n]);
        }
      }
    };
    w_load(wst0, 0, cc);
    if (nsteps > 1) w_load(wst1, 1 / nkc, cc + (1 % nkc) * KC);
    w_store(wst0, 0);
    __syncthreads();
    for (int s = 0; s < nsteps; s += 2) {
      // even step: chunk s is in LDS[0]; chunk s+1 sits in wst1; fetch chunk s+2 into wst0
      if (s + 2 < nsteps) w_load(wst0, (s + 2) / nkc, cc + ((s + 2) % nkc) * KC);
      compute(s);
      if (s + 1 < nsteps) w_store(wst1, 1);
      __syncthreads();
      if (s + 1 >= nsteps) break;
      // odd step: chunk s+1 is in LDS[1]; chunk s+2 sits in wst0; fetch chunk s+3 into wst1
      if (s + 3 < nsteps) w_load(wst1, (s + 3) / nkc, cc + ((s + 3) % nkc) * KC);
      compute(s + 1);
      if (s + 2 < nsteps) w_store(wst0, 0);
      __syncthreads();
    }
  }

  if (p.dbg & 32) return;
  // ---- accumulators -> LDS (element type T) -> coalesced epilogue
  float bvals[NW];
#pragma unroll
  for (int n = 0; n < NW; ++n) {
    const int col = n0 + wn * (BN / WN) + n * 32 + r;
    bvals[n] = (p.bias && col < p.Cout) ? p.bias[col] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int n = 0; n < NW; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        const int col = wn * (BN / WN) + n * 32 + r;
        // bias joins in fp32 before the (single, for residual-free layers) rounding to T
        lds_c[row * PITCH_C + col] = (T)(acc[i][n][e] + bvals[n]);
      }
  __syncthreads();
  T* yg = p.y ? reinterpret_cast<T*>(p.y) + (long long)b * p.y_bs : nullptr;
  const T* rg = p.res ? reinterpret_cast<const T*>(p.res) + (long long)b * p.res_bs : nullptr;
  const T* hg = p.epi_act ? reinterpret_cast<const T*>(p.gate_h) + (long long)b * p.gh_bs : nullptr;
  const int len_out = p.lens_out ? p.lens_out[b] : 0x7fffffff;
  constexpr int CV = BN / EPV;
  constexpr int NE = BM * CV / NT;    // vectors per thread
  // A full tile (the overwhelmingly common case) needs no per-vector bounds logic: every condition
  // below is then wave-uniform and the eight vectors of a thread are independent straight-line code.
  const bool full = (t0 + BM <= p.Tout) && (n0 + BN <= p.Cout) && ((t0 + BM - 1) * p.out_stride + p.out_offset < p.Ty);
  const int cv0 = tid % CV, row0 = tid / CV;      // this thread's column vector; rows row0 + it*(NT/CV)
  const int col = n0 + cv0 * EPV;
  Vec<T, EPV> rv[NE], uv[NE];
  bool okv[NE];
#pragma unroll
  for (int it = 0; it < NE; ++it) {
    const int t = t0 + row0 + it * (NT / CV);
    const int ty = t * p.out_stride + p.out_offset;
    okv[it] = full || ((t < p.Tout) && (col < p.Cout) && (ty < p.Ty));
#pragma unroll
    for (int e = 0; e < EPV; ++e) { rv[it].v[e] = (T)0.f; uv[it].v[e] = (T)0.f; }
    if (rg && okv[it]) rv[it] = *reinterpret_cast<const Vec<T, EPV>*>(rg + (long long)ty * p.ldr + col);
    if (p.epi_act && okv[it]) uv[it] = *reinterpret_cast<const Vec<T, EPV>*>(hg + (long long)ty * p.ldgh + col);
  }
  const int site = p.act_out ? col / p.site_width : 0;
  const unsigned key = site_key(p, site);
  const int cs = col - site * (p.act_out ? p.site_width : 0);
#pragma unroll
  for (int it = 0; it < NE; ++it) {
    const int row = row0 + it * (NT / CV);
    const int ty = (t0 + row) * p.out_stride + p.out_offset;
    Vec<T, EPV> c = *reinterpret_cast<const Vec<T, EPV>*>(lds_c + row * PITCH_C + cv0 * EPV);
    float o[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) o[e] = (float)c.v[e];
    if (p.epi_act) {  // d relu(dropout(h))/dh = scale * [u != 0] with u the stored activated tensor
#pragma unroll
      for (int e = 0; e < EPV; ++e) o[e] = ((float)uv[it].v[e] != 0.f) ? o[e] * p.drop_scale : 0.f;
    }
    const float keep_row = (ty >= len_out) ? 0.f : 1.f;
    if (rg) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) o[e] = fmaf(o[e], keep_row, (float)rv[it].v[e]);
    } else {
#pragma unroll
      for (int e = 0; e < EPV; ++e) o[e] *= keep_row;
    }
    if (yg && okv[it] && !(p.dbg & 8)) {
      Vec<T, EPV> out;
#pragma unroll
      for (int e = 0; e < EPV; ++e) out.v[e] = (T)o[e];
      *reinterpret_cast<Vec<T, EPV>*>(yg + (long long)ty * p.ldy + col) = out;
    }
    if (p.act_out && okv[it]) {  // u = relu(dropout(y)), counter-based mask, applied in fp32 before rounding
      const unsigned long long base = ((unsigned long long)b * p.Ty + ty) * p.site_width + cs;
      Vec<T, EPV> ua;
#pragma unroll
      for (int e = 0; e < EPV; e += 2) {   // base is even: elements (e, e+1) share one hash
        const unsigned h = fmix32((unsigned)((base + e) >> 1) * 0x9E3779B1u + key);
        const bool k0 = (h & 0xFFFFu) >= p.drop_thresh16, k1 = (h >> 16) >= p.drop_thresh16;
        ua.v[e] = (T)((k0 && o[e] > 0.f) ? o[e] * p.drop_scale : 0.f);
        ua.v[e + 1] = (T)((k1 && o[e + 1] > 0.f) ? o[e + 1] * p.drop_scale : 0.f);
      }
      *reinterpret_cast<Vec<T, EPV>*>(reinterpret_cast<T*>(p.y_act) + (long long)b * p.ya_bs + (long long)ty * p.ldya + col) = ua;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (bf16, C_in % 128 == 0, C_out % 128 == 0, stride 1): both operands go HBM/L2 -> LDS
// with global_load_lds_dwordx4 (no VGPR round trip, no ds_write, no staging arithmetic in the loop).
// A DMA wave-instruction writes 64 x 16 B linearly, so rows cannot be padded; bank conflicts are
// avoided by an XOR swizzle of the 16-byte chunk index with the row index instead:
//   activations: applied on the per-lane SOURCE address while staging, undone by the fragment reads;
//   weights    : pre-swizzled in global memory by smt_pack_weight(swizzle = 1), copied linearly.
// Rows outside [0, len) are fetched from a zero page.
constexpr int DMA_BM = 128, DMA_BN = 128, DMA_KC = 128, DMA_NT = 512;
#ifndef SMT_ABL
#define SMT_ABL 0   // ablation build switches for conv_gemm_dma_kernel (tools/ablate_dma.sh); 0 in the product
#endif
constexpr int ABL = SMT_ABL;

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_dst_wave_base, 16, 0, 0);
}

// MW = 32-row tiles per wave: BM = 64 * MW rows per workgroup.  MW = 4 (256 rows) halves the weight
// stream per output row -- the L2 -> LDS weight DMA (taps x 32 KiB per tile) is what bounds this kernel.
template <int MW>
__global__ __launch_bounds__(DMA_NT) void conv_gemm_dma_kernel(ConvArgs p, const __bf16* __restrict__ zero_page) {
  typedef __bf16 T;
  constexpr int EPV = 8, BM = 64 * MW, BN = DMA_BN, KC = DMA_KC, NT = DMA_NT;
  constexpr int WN = 4;                         // waves 2 (rows) x 4 (cols); wave tile (32 MW) x 32
  constexpr int ROWB = KC * 2;                  // 256 bytes per LDS row (128 channels)
  constexpr int PITCH_C = BN + EPV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, hh = lane >> 5;

  const int ntiles = p.tiles_per_batch * p.B * p.rs;
  const int per_xcd = (ntiles + 7) / 8;
  const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  // Dilation classes: a conv with dilation d >= 8 is run as d independent DENSE convs over the row
  // classes t = cls (mod d) (rs = d, taps at distance 1 in the class domain), so the halo of a 128-row
  // tile is (taps - 1) rows instead of (taps - 1) * d.  Each class row is still a whole 256-byte
  // segment for the DMA.  rs == 1 is the plain case.
  const int rs = p.rs;
  const int bb = tile / p.tiles_per_batch;
  const int b = bb / rs, cls = bb - b * rs;
  const int t0 = (tile % p.tiles_per_batch) * BM;
  const int n0 = blockIdx.y * BN;
  const int Tc = (p.Tout - cls + rs - 1) / rs;          // rows of this class
  if (t0 >= Tc) return;

  const int rows_in = (BM - 1) + (p.taps - 1) * p.dil + 1;
  const int rows_pad = (rows_in + 3) & ~3;      // DMA granularity: 4 rows (1 KiB) per wave-instruction
  unsigned char* lds_a = smem;
  const size_t a_bytes = align_up((size_t)max(rows_pad * ROWB, BM * PITCH_C * 2), 1024);
  unsigned char* lds_w = smem + a_bytes;        // 2 x [BN rows][256 B]
  T* lds_c = reinterpret_cast<T*>(smem);

  const T* xg = reinterpret_cast<const T*>(p.x) + (long long)b * p.x_bs + (long long)cls * p.ldx;
  const long long ldx = (long long)p.ldx * rs;
  const unsigned char* wg = reinterpret_cast<const unsigned char*>(p.w);
  const int len_full = p.lens_in ? min(p.lens_in[b], p.Tin) : p.Tin;
  const int len_in = max(0, (len_full - cls + rs - 1) / rs);
  const int tin0 = t0 - p.pad;
  const int ncc = p.Cin / KC;

  f32x16 acc[MW];
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // epilogue operands (residual, activation source) are requested NOW so that they arrive under the GEMM
  constexpr int CV = BN / EPV, NE = BM * CV / NT;
  const int cv0 = tid % CV, row0 = tid / CV;
  const int col = n0 + cv0 * EPV;
  const T* rg = p.res ? reinterpret_cast<const T*>(p.res) + (long long)b * p.res_bs : nullptr;
  const T* hg = p.epi_act ? reinterpret_cast<const T*>(p.gate_h) + (long long)b * p.gh_bs : nullptr;
  Vec<T, EPV> rv[NE], uv[NE];
  bool okv[NE];
#pragma unroll
  for (int it = 0; it < NE; ++it) {
    const int tc = t0 + row0 + it * (NT / CV);
    okv[it] = (tc < Tc);
    const long long t = cls + (long long)rs * tc;          // actual row
#pragma unroll
    for (int e = 0; e < EPV; ++e) { rv[it].v[e] = (T)0.f; uv[it].v[e] = (T)0.f; }
    if (rg && okv[it]) rv[it] = *reinterpret_cast<const Vec<T, EPV>*>(rg + t * p.ldr + col);
    if (p.epi_act && okv[it]) uv[it] = *reinterpret_cast<const Vec<T, EPV>*>(hg + t * p.ldgh + col);
  }

  const int lrow = lane >> 4, lch = lane & 15;  // this lane's (row within the 4-row group, chunk) of a DMA instruction
  auto stage_w = [&](int j, int cc, int buf) {  // 32 KiB = 32 wave-instructions, 4 per wave
#pragma unroll
    for (int q = 0; q < (BN / 4) / (NT / 64); ++q) {
      const int g = wave + (NT / 64) * q;       // 4-row group index 0..31
      const int co = n0 + 4 * g + lrow;
      const unsigned char* src = wg + (((size_t)j * p.Cout + co) * ncc + cc) * ROWB + lch * 16;
      dma16(src, lds_w + (size_t)buf * BN * ROWB + g * 1024);
    }
  };

  for (int cc = 0; cc < ncc; ++cc) {
    __syncthreads();  // previous chunk's readers are done with lds_a / lds_w
    if (!(ABL & 1))
    for (int g = wave; g < rows_pad / 4; g += NT / 64) {
      const int row = 4 * g + lrow;
      const int tin = tin0 + row;
      const bool ok = (row < rows_in) && (tin >= 0) && (tin < len_in);
      const T* src = ok ? xg + (long long)tin * ldx + cc * KC + ((lch ^ (row & 15)) * EPV) : zero_page + lch * EPV;
      dma16(src, lds_a + g * 1024);
    }
    stage_w(0, cc, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nsteps = p.taps;
    for (int s = 0; s < nsteps; ++s) {
      if (s + 1 < nsteps && !(ABL & 2)) stage_w(s + 1, cc, (s + 1) & 1);
      const unsigned char* wb = lds_w + (size_t)(s & 1) * BN * ROWB + (wn * 32 + r) * ROWB;
      const int bsw = (wn * 32 + r) & 15;
      const int arow0 = wm * (BM / 2) + s * p.dil + r;
      bf16x8 bv0, av0[MW];
      if (ABL & 16) bv0 = *reinterpret_cast<const bf16x8*>(wb + ((hh ^ bsw) << 4));
      if (ABL & 8) {
#pragma unroll
        for (int i = 0; i < MW; ++i) av0[i] = *reinterpret_cast<const bf16x8*>(lds_a + (arow0 + 32 * i) * ROWB + ((hh ^ ((arow0 + 32 * i) & 15)) << 4));
      }
#pragma unroll
      for (int kk = 0; kk < KC / 16; ++kk) {
        const int ch = 2 * kk + hh;
        bf16x8 bv = (ABL & 16) ? bv0 : *reinterpret_cast<const bf16x8*>(wb + ((ch ^ bsw) << 4));
#pragma unroll
        for (int i = 0; i < MW; ++i) {
          const int ar = arow0 + 32 * i;
          bf16x8 av = (ABL & 8) ? av0[i] : *reinterpret_cast<const bf16x8*>(lds_a + ar * ROWB + ((ch ^ (ar & 15)) << 4));
          if (ABL & 4) asm volatile("" ::"v"(av), "v"(bv));
          else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue: identical to conv_gemm_kernel<bf16, 128, 512>
  const float bval = (p.bias && n0 + wn * 32 + r < p.Cout) ? p.bias[n0 + wn * 32 + r] : 0.f;
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      lds_c[row * PITCH_C + wn * 32 + r] = (T)(acc[i][e] + bval);
    }
  __syncthreads();
  T* yg = p.y ? reinterpret_cast<T*>(p.y) + (long long)b * p.y_bs : nullptr;
  const int len_out = p.lens_out ? p.lens_out[b] : 0x7fffffff;
  const int site = p.act_out ? col / p.site_width : 0;
  const unsigned key = site_key(p, site);
  const int cs = col - site * (p.act_out ? p.site_width : 0);
#pragma unroll
  for (int it = 0; it < NE; ++it) {
    const int row = row0 + it * (NT / CV);
    const int ty = cls + rs * (t0 + row);                  // actual output row
    Vec<T, EPV> c = *reinterpret_cast<const Vec<T, EPV>*>(lds_c + row * PITCH_C + cv0 * EPV);
    float o[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) o[e] = (float)c.v[e];
    if (p.epi_act) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) o[e] = ((float)uv[it].v[e] != 0.f) ? o[e] * p.drop_scale : 0.f;
    }
    const float keep_row = (ty >= len_out) ? 0.f : 1.f;
    if (rg) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) o[e] = fmaf(o[e], keep_row, (float)rv[it].v[e]);
    } else {
#pragma unroll
      for (int e = 0; e < EPV; ++e) o[e] *= keep_row;
    }
    if (yg && okv[it]) {
      Vec<T, EPV> out;
#pragma unroll
      for (int e = 0; e < EPV; ++e) out.v[e] = (T)o[e];
      *reinterpret_cast<Vec<T, EPV>*>(yg + (long long)ty * p.ldy + col) = out;
    }
    if (p.act_out && okv[it]) {
      const unsigned long long base = ((unsigned long long)b * p.Ty + ty) * p.site_width + cs;
      Vec<T, EPV> ua;
#pragma unroll
      for (int e = 0; e < EPV; e += 2) {
        const unsigned h = fmix32((unsigned)((base + e) >> 1) * 0x9E3779B1u + key);
        const bool k0 = (h & 0xFFFFu) >= p.drop_thresh16, k1 = (h >> 16) >= p.drop_thresh16;
        ua.v[e] = (T)((k0 && o[e] > 0.f) ? o[e] * p.drop_scale : 0.f);
        ua.v[e + 1] = (T)((k1 && o[e + 1] > 0.f) ? o[e + 1] * p.drop_scale : 0.f);
      }
      *reinterpret_cast<Vec<T, EPV>*>(reinterpret_cast<T*>(p.y_act) + (long long)b * p.ya_bs + (long long)ty * p.ldya + col) = ua;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Persistent 1x1 variant (bf16, C_in == 128, C_out % 128 == 0): an HBM-bound layer.  Each workgroup keeps
// its 128 x 128 weight block in REGISTERS (32 VGPRs per lane) for its whole run of row tiles, streams the
// activation tiles through an LDS double buffer by LDS-DMA (tile i+1 is in flight while tile i is
// multiplied, staged and stored) and requests the epilogue operands of a tile before its MFMAs.
constexpr int P1_BUF = 35 * 1024;   // 128 rows x 256 B operand tile, reused as the 128 x 272 B output staging tile

__global__ __launch_bounds__(DMA_NT) void conv1x1_dma_kernel(ConvArgs p, const __bf16* __restrict__ zero_page,
                                                             int tiles_per_wg) {
  typedef __bf16 T;
  constexpr int EPV = 8, BM = DMA_BM, BN = DMA_BN, KC = DMA_KC, NT = DMA_NT;
  constexpr int WN = 4, MW = 2, ROWB = KC * 2, PITCH_C = BN + EPV;
  constexpr int CV = BN / EPV, NE = BM * CV / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, hh = lane >> 5;
  const int lrow = lane >> 4, lch = lane & 15;
  const int n0 = blockIdx.y * BN;

  // contiguous run of tiles per workgroup; runs of workgroups b, b+8, ... (one XCD) are adjacent
  const int ntiles = p.tiles_per_batch * p.B;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = wg * tiles_per_wg;
  const int tile_end = min(ntiles, tile_begin + tiles_per_wg);
  if (tile_begin >= tile_end) return;

  // weight fragments -> registers (weights are packed swizzled: chunk c of row co sits at c ^ (co & 15))
  bf16x8 wfrag[KC / 16];
  {
    const int co = n0 + wn * 32 + r;
    const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.w) + (size_t)co * ROWB;
#pragma unroll
    for (int kk = 0; kk < KC / 16; ++kk) wfrag[kk] = *reinterpret_cast<const bf16x8*>(wrow + (((2 * kk + hh) ^ (co & 15)) << 4));
  }
  const float bval = p.bias ? p.bias[n0 + wn * 32 + r] : 0.f;
  const int cv0 = tid % CV, row0 = tid / CV;
  const int col = n0 + cv0 * EPV;
  const int site = p.act_out ? col / p.site_width : 0;
  const unsigned key = site_key(p, site);
  const int cs = col - site * (p.act_out ? p.site_width : 0);

  auto stage_a = [&](int tile, int buf) {
    const int b = tile / p.tiles_per_batch;
    const int t0 = (tile % p.tiles_per_batch) * BM;
    const T* xg = reinterpret_cast<const T*>(p.x) + (long long)b * p.x_bs;
    const int len_in = p.lens_in ? min(p.lens_in[b], p.Tin) : p.Tin;
#pragma unroll
    for (int q = 0; q < (BM / 4) / (NT / 64); ++q) {
      const int g = wave + (NT / 64) * q;
      const int row = 4 * g + lrow;
      const int tin = t0 + row;
      const T* src = (tin < len_in) ? xg + (long long)tin * p.ldx + ((lch ^ (row & 15)) * EPV) : zero_page + lch * EPV;
      dma16(src, smem + (size_t)buf * P1_BUF + g * 1024);
    }
  };

  stage_a(tile_begin, 0);
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    const int b = tile / p.tiles_per_batch;
    const int t0 = (tile % p.tiles_per_batch) * BM;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's operand has landed (and older stores retired)
    __syncthreads();                                    // ... for every wave; the other buffer is free again
    if (tile + 1 < tile_end) stage_a(tile + 1, buf ^ 1);
    // epilogue operands of THIS tile: requested before the MFMAs
    const T* rg = p.res ? reinterpret_cast<const T*>(p.res) + (long long)b * p.res_bs : nullptr;
    const T* hg = p.epi_act ? reinterpret_cast<const T*>(p.gate_h) + (long long)b * p.gh_bs : nullptr;
    Vec<T, EPV> rv[NE], uv[NE];
    bool okv[NE];
#pragma unroll
    for (int it = 0; it < NE; ++it) {
      const int t = t0 + row0 + it * (NT / CV);
      okv[it] = (t < p.Tout);
#pragma unroll
      for (int e = 0; e < EPV; ++e) { rv[it].v[e] = (T)0.f; uv[it].v[e] = (T)0.f; }
      if (rg && okv[it]) rv[it] = *reinterpret_cast<const Vec<T, EPV>*>(rg + (long long)t * p.ldr + col);
      if (p.epi_act && okv[it]) uv[it] = *reinterpret_cast<const Vec<T, EPV>*>(hg + (long long)t * p.ldgh + col);
    }
    const unsigned char* lds_a = smem + (size_t)buf * P1_BUF;
    f32x16 acc[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KC / 16; ++kk) {
      const int ch = 2 * kk + hh;
#pragma unroll
      for (int i = 0; i < MW; ++i) {
        const int ar = wm * 64 + 32 * i + r;
        bf16x8 av = *reinterpret_cast<const bf16x8*>(lds_a + ar * ROWB + ((ch ^ (ar & 15)) << 4));
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wfrag[kk], acc[i], 0, 0, 0);
      }
    }
    __syncthreads();   // every wave is done reading the operand tile: reuse it as the output staging tile
    T* lds_c = reinterpret_cast<T*>(smem + (size_t)buf * P1_BUF);
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        lds_c[row * PITCH_C + wn * 32 + r] = (T)(acc[i][e] + bval);
      }
    __syncthreads();
    T* yg = p.y ? reinterpret_cast<T*>(p.y) + (long long)b * p.y_bs : nullptr;
    const int len_out = p.lens_out ? p.lens_out[b] : 0x7fffffff;
#pragma unroll
    for (int it = 0; it < NE; ++it) {
      const int row = row0 + it * (NT / CV);
      const int ty = t0 + row;
      Vec<T, EPV> c = *reinterpret_cast<const Vec<T, EPV>*>(lds_c + row * PITCH_C + cv0 * EPV);
      float o[EPV];
#pragma unroll
      for (int e = 0; e < EPV; ++e) o[e] = (float)c.v[e];
      if (p.epi_act) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) o[e] = ((float)uv[it].v[e] != 0.f) ? o[e] * p.drop_scale : 0.f;
      }
      const float keep_row = (ty >= len_out) ? 0.f : 1.f;
      if (rg) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) o[e] = fmaf(o[e], keep_row, (float)rv[it].v[e]);
      } else {
#pragma unroll
        for (int e = 0; e < EPV; ++e) o[e] *= keep_row;
      }
      if (yg && okv[it]) {
        Vec<T, EPV> out;
#pragma unroll
        for (int e = 0; e < EPV; ++e) out.v[e] = (T)o[e];
        *reinterpret_cast<Vec<T, EPV>*>(yg + (long long)ty * p.ldy + col) = out;
      }
      if (p.act_out && okv[it]) {
        const unsigned long long base = ((unsigned long long)b * p.Ty + ty) * p.site_width + cs;
        Vec<T, EPV> ua;
#pragma unroll
        for (int e = 0; e < EPV; e += 2) {
          const unsigned h = fmix32((unsigned)((base + e) >> 1) * 0x9E3779B1u + key);
          const bool k0 = (h & 0xFFFFu) >= p.drop_thresh16, k1 = (h >> 16) >= p.drop_thresh16;
          ua.v[e] = (T)((k0 && o[e] > 0.f) ? o[e] * p.drop_scale : 0.f);
          ua.v[e + 1] = (T)((k1 && o[e + 1] > 0.f) ? o[e + 1] * p.drop_scale : 0.f);
        }
        *reinterpret_cast<Vec<T, EPV>*>(reinterpret_cast<T*>(p.y_act) + (long long)b * p.ya_bs + (long long)ty * p.ldya + col) = ua;
      }
    }
    // the next iteration's top barrier orders these LDS reads before the buffer is overwritten by a DMA
  }
}

// ------------------------------------------------------------------------------------------------
// Weight-stationary variant (bf16, C_in == 128, 3..9 taps): the k x 128 x 128 weight block of a dilated conv
// is at most 288 KiB -- too large for LDS, but it fits the REGISTER FILE of one CU.  Four waves (one per
// SIMD, up to 512 VGPRs each) each own 32 output channels and keep that slice of every tap in registers
// for the whole run of a persistent workgroup, so nothing but activations moves through LDS:
//   * activation tiles (128 rows + halo) arrive by LDS-DMA into a double buffer: tile i+1 and the epilogue
//     operands of tile i are in flight while tile i is multiplied, and the stores of tile i-1 drain;
//   * ONE barrier per tile (the buffer swap); the tap loop has none;
//   * the MFMA computes the TRANSPOSED tile (A = weights, B = activations), so a lane ends up with 4
//     consecutive output channels of one row; a v_permlane32_swap pairs them to 16-byte pieces that are stored
//     straight from registers -- no LDS staging of the output, no epilogue barrier;
//   * each wave DMAs its own 64-byte column slice of the residual / activation-source rows to LDS and reads
//     only that back, so the epilogue operands need neither registers during the tap loop nor a barrier.
// Measured motivation (tools/ablate_dma.sh): the streaming kernel spends as long waiting for HBM (tile in,
// tile out) as it does in MFMAs, and with one workgroup per CU the two never overlap.
// Buffer addressing (raw V#, byte offsets): rows outside [0, valid rows) fall outside num_records and read as zero /
// are not stored -- the hardware's range check replaces the per-lane bounds tests and zero-page selects, and a per-lane
// 32-bit offset replaces the 64-bit address arithmetic (both were VALU work serial with the MFMAs: tools/ws_phases.py
// measured 1,100-2,000 cycles per tile for the issue of ~5 DMA instructions per wave).
typedef short s16x2v __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2v __attribute__((ext_vector_type(2)));
// relu on a packed bf16 pair: a negative bf16 is a negative int16 (v_pk_max_i16)
__device__ __forceinline__ unsigned pk_relu_bf16(unsigned w) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2v, w), s16x2v{0, 0}));
}
// 0xFFFF per 16-bit half of h that is >= thr (thr_m1 = thr - 1 in both halves, thr >= 1): saturating subtract, min 1, negate
__device__ __forceinline__ unsigned pk_keep_mask(unsigned h, unsigned thr_m1) {
  u16x2v d = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2v, h), __builtin_bit_cast(u16x2v, thr_m1));
  d = __builtin_elementwise_min(d, u16x2v{1, 1});
  return __builtin_bit_cast(unsigned, (u16x2v)(u16x2v{0, 0} - d));
}

#ifndef SMT_WS_STAMP
#define SMT_WS_STAMP 0   // diagnostic build (tools/ws_phases.sh): per-wave cycle sums of the phases of conv_ws2_kernel
#endif
#if SMT_WS_STAMP
__device__ unsigned long long ws_dbg[256 * 8 * 8];
#define WS_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define WS_ACC(k, a, b) do { if (lane == 0 && wg < 256) ws_dbg[(wg * 8 + wave) * 8 + (k)] += (b) - (a); } while (0)
#else
#define WS_T(var) do {} while (0)
#define WS_ACC(k, a, b) do {} while (0)
#endif
constexpr int WS_AGPR_TAPS = 8;   // taps whose weights are pinned to AccVGPRs (8 x 32 = all 256)
constexpr int WS_BM = 128, WS_NT = 256, WS_EPI = 128 * 256;   // rows per tile, threads, epilogue-operand tile bytes

// MODE fixes the epilogue at compile time (straight-line code instead of ~85 branches in the unrolled epilogue, which
// a single wave per SIMD cannot hide): 1 = activated output only (K2 forward), 2 = y with activation-gradient mask and
// residual (K2 data gradient), 0 = whatever the descriptor asks for.
template <int NTAPS, int MODE>
__global__ __launch_bounds__(WS_NT) void conv_ws_kernel(ConvArgs p, const __bf16* __restrict__ zero_page,
                                                        int tiles_per_wg, int buf_bytes) {
  const bool has_y = MODE == 0 ? (p.y != nullptr) : (MODE == 2);
  const bool has_act_out = MODE == 0 ? (p.act_out != 0) : (MODE == 1);
  const bool has_res = MODE == 0 ? (p.res != nullptr) : (MODE == 2);
  const bool has_epi_act = MODE == 0 ? (p.epi_act != 0) : (MODE == 2);
  typedef __bf16 T;
  constexpr int BM = WS_BM, BN = 128, KC = 128, NT = WS_NT, MW = BM / 32, ROWB = KC * 2;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int lrow = lane >> 4, lch = lane & 15;
  const int n0 = blockIdx.y * BN;
  const int rs = p.rs;

  const int ntiles = p.tiles_per_batch * p.B * rs;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = wg * tiles_per_wg;
  const int tile_end = min(ntiles, tile_begin + tiles_per_wg);
  if (tile_begin >= tile_end) return;

  // epilogue-operand tiles: per wave [128 rows][64 B] (its 32 output channels), 16-byte chunks XOR-swizzled
  unsigned char* lds_res = smem + 2 * (size_t)buf_bytes + wave * (WS_EPI / 4);
  unsigned char* lds_act = lds_res + WS_EPI;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // this wave's 32 output channels of every tap -> registers (packed swizzled: chunk c of row co at c ^ (co & 15))
  bf16x8 wfrag[NTAPS][KC / 16];
  {
    const int co = n0 + wave * 32 + r;
#pragma unroll
    for (int s = 0; s < NTAPS; ++s) {
      const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.w) + ((size_t)s * p.Cout + co) * ROWB;
#pragma unroll
      for (int kk = 0; kk < KC / 16; ++kk)
        wfrag[s][kk] = *reinterpret_cast<const bf16x8*>(wrow + (((2 * kk + hh) ^ (co & 15)) << 4));
    }
  }
  // accumulator element 4g + k of a lane = output channel col0 + 8g + k (this lane's row: 32 i + r)
  const int col0 = n0 + wave * 32 + 4 * hh;
  float bval[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bval[e] = p.bias ? p.bias[col0 + 8 * (e >> 2) + (e & 3)] : 0.f;
  const int rows_in = BM + (NTAPS - 1) * p.dil;
  const int rows_pad = (rows_in + 3) & ~3;

  // tile -> (batch, class, first class row)
  auto decode = [&](int tile, int& b, int& cls, int& t0) {
    const int bb = tile / p.tiles_per_batch;
    b = bb / rs; cls = bb - b * rs;
    t0 = (tile - bb * p.tiles_per_batch) * BM;
  };
  // buffer addressing (ws_rsrc / ws_dma16): per-lane byte offsets are tile-invariant up to a scalar; rows outside
  // [0, valid rows) are out of range of the V# and read as zero.  Group g = wave + 4 j covers rows 16 j + 4 wave + lrow,
  // so the swizzle term (row & 15) does not depend on j.
  const unsigned pitch_x = (unsigned)p.ldx * rs * 2u;
  const unsigned voff_a0 = (unsigned)(4 * wave + lrow) * pitch_x + (unsigned)((lch ^ ((4 * wave + lrow) & 15)) << 4);
  const int ngroups = rows_pad >> 2;
  auto stage_a = [&](int tile, int buf) {
    int b, cls, t0;
    decode(tile, b, cls, t0);
    const int len_full = p.lens_in ? min(scalar_load_i32(p.lens_in + b), p.Tin) : p.Tin;
    const int len_in = max(0, (len_full - cls + rs - 1) / rs);
    const __amdgpu_buffer_rsrc_t rx = ws_rsrc(p.x, ((long long)b * p.x_bs + (long long)cls * p.ldx) * 2, (unsigned)len_in * pitch_x);
    unsigned vo = voff_a0 + (unsigned)(t0 - p.pad) * pitch_x;      // rows before the item wrap to huge offsets: zero
    unsigned char* dst = smem + (size_t)buf * buf_bytes + wave * 1024;
    for (int g = wave; g < ngroups; g += NT / 64) {
      ws_dma16(rx, vo, dst);
      vo += 16u * pitch_x; dst += 4096;
    }
  };
  // this wave's slice of an epilogue operand: one DMA instruction = 16 rows x 64 B; slot c of row n holds the
  // 16-byte chunk c ^ ((n >> 2) & 3) of the slice (keeps the 8-byte fragment reads at <= 2-way bank conflicts)
  const unsigned echunk = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
  auto stage_epi = [&](const void* base, long long bs, int ld, int b, int cls, int t0, int Tc, unsigned char* dst) {
    const unsigned pitch = (unsigned)ld * rs * 2u;
    const __amdgpu_buffer_rsrc_t re = ws_rsrc(base, ((long long)b * bs + (long long)cls * ld + n0 + wave * 32) * 2, (unsigned)Tc * pitch);
    unsigned vo = (unsigned)(t0 + (lane >> 2)) * pitch + echunk;
#pragma unroll
    for (int q = 0; q < BM / 16; ++q) {
      ws_dma16(re, vo, dst + q * 1024);
      vo += 16u * pitch;
    }
  };

  stage_a(tile_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    int b, cls, t0;
    decode(tile, b, cls, t0);
    const int Tc = (p.Tout - cls + rs - 1) / rs;
    // tile `tile` is in LDS for every wave (each waited for its own DMAs before its previous epilogue) and
    // every wave is done reading the other buffer
    WS_T(c0);
    __syncthreads();
    WS_T(c1);
    if (tile + 1 < tile_end) stage_a(tile + 1, buf ^ 1);
    WS_T(c2);
    if (has_res) stage_epi(p.res, p.res_bs, p.ldr, b, cls, t0, Tc, lds_res);
    if (has_epi_act) stage_epi(p.gate_h, p.gh_bs, p.ldgh, b, cls, t0, Tc, lds_act);

    WS_T(c4);
    f32x16 acc[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    // Software pipeline over the NTAPS x 8 k-steps, written as asm so that it stays a pipeline: the four
    // fragments of step q+1 are requested before the four MFMAs of step q (one wave per SIMD -- nothing else
    // hides the LDS latency) and `s_waitcnt lgkmcnt(4)` retires exactly the older four (LDS returns in order;
    // a stray scalar load in flight only makes the wait more conservative).  The asm also pins the register
    // classes: the first WS_AGPR_TAPS taps of the weight block live in AccVGPRs and are read directly as an
    // operand, the rest and the accumulators in VGPRs.  Left to itself the compiler keeps all 288 weight
    // registers in the 256 VGPRs, spills to AccVGPRs, and serialises every ds_read behind the previous MFMA.
    // fragment address of step (s, kk) = tap_base[s] ^ (32 kk): buffers are 1 KiB-aligned, so the XOR swizzle of
    // the 16-byte chunk index (bits 4..7) can be applied after the buffer base has been added
    const unsigned abase = lds_base + (unsigned)buf * (unsigned)buf_bytes;
    unsigned tap_base[NTAPS];
#pragma unroll
    for (int s = 0; s < NTAPS; ++s) {
      const int ar = r + s * p.dil;                 // rows ar + 32 i share the swizzle term (ar & 15)
      tap_base[s] = abase + ar * ROWB + ((hh ^ (ar & 15)) << 4);
    }
    auto frag_addr = [&](int q) -> unsigned { return tap_base[q / (KC / 16)] ^ (32u * (q % (KC / 16))); };
    bf16x8 afr[2][MW];
    if (!(ABL & 4)) {
    {
      const unsigned ap = frag_addr(0);
#pragma unroll
      for (int i = 0; i < MW; ++i)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afr[0][i]) : "v"(ap), "n"(i * 32 * ROWB));
    }
#pragma unroll
    for (int q = 0; q < NTAPS * (KC / 16); ++q) {
      if (q + 1 < NTAPS * (KC / 16)) {
        const unsigned ap = frag_addr(q + 1);
        if (ABL & 8) {           // ablation: no fragment reads after the first step
#pragma unroll
          for (int i = 0; i < MW; ++i) afr[(q + 1) & 1][i] = afr[q & 1][i];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
#pragma unroll
        for (int i = 0; i < MW; ++i)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afr[(q + 1) & 1][i]) : "v"(ap), "n"(i * 32 * ROWB));
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        }
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int i = 0; i < MW; ++i) {
        // first use of an accumulator: the compiler may have initialised it with a VALU move in the instruction
        // just before, and it cannot see that an MFMA follows (VALU write -> MFMA SrcC needs wait states)
        if (q == 0) asm volatile("s_nop 4" : "+v"(acc[i]));
        // D^T = W * A^T: operand A = weights [32 co x 16 k], operand B = activations [16 k x 32 rows]
        if (q / (KC / 16) < WS_AGPR_TAPS)
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(wfrag[q / (KC / 16)][q % (KC / 16)]), "v"(afr[q & 1][i]));
        else
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(wfrag[q / (KC / 16)][q % (KC / 16)]), "v"(afr[q & 1][i]));
      }
    }
    // The hazard recogniser does not see MFMAs inside asm: let the last ones drain before VALU reads acc.  The
    // accumulators are operands of the nops so that no reader of acc can be scheduled above them.
    static_assert(MW == 4, "drain below names four accumulators");
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    }
    WS_T(c5);
    // next tile + this tile's epilogue operands have landed (issued a whole tap loop ago); older stores retired
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WS_T(c6);
    if (ABL & 32) {   // ablation: no epilogue
      asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
      continue;
    }

    // ---- epilogue straight from the accumulators; same arithmetic as the other kernels: bf16(acc + bias) first.
    // Stores go through range-checked buffer descriptors (rows >= Tc are dropped by the hardware, 32-bit offsets).
    const int len_out = p.lens_out ? scalar_load_i32(p.lens_out + b) : 0x7fffffff;
    const unsigned pitch_y = (unsigned)p.ldy * rs * 2u, pitch_u = (unsigned)p.ldya * rs * 2u;
    const __amdgpu_buffer_rsrc_t ry = ws_rsrc(has_y ? p.y : p.x, has_y ? ((long long)b * p.y_bs + (long long)cls * p.ldy + n0 + wave * 32) * 2 : 0,
                                              has_y ? (unsigned)Tc * pitch_y : 0u);
    const __amdgpu_buffer_rsrc_t ru = ws_rsrc(has_act_out ? p.y_act : p.x,
                                              has_act_out ? ((long long)b * p.ya_bs + (long long)cls * p.ldya + n0 + wave * 32) * 2 : 0,
                                              has_act_out ? (unsigned)Tc * pitch_u : 0u);
    if constexpr (MODE == 2) {
      // dx = y * scale * [u != 0] * row mask + residual, written for few VALU instructions (tools/ws_phases.py: at one wave
      // per SIMD the epilogue and the DMA issue are serial with the tap loop).  The eight LDS reads of a row block are
      // issued together, so their latency is paid once per row block.  (Tried and dropped in round 3: loading the two
      // operands straight into registers in the store layout after touching their lines before the tap loop -- the 4-byte
      // touches, 64 lines per instruction, slowed the tap loop by 20 %: 826 vs 775 us at 9 taps.)
#pragma unroll
      for (int i = 0; i < MW; ++i) {
        const int row = 32 * i + r;
        const int tc = t0 + row;
        const float srow = (cls + rs * tc >= len_out) ? 0.f : p.drop_scale;     // row mask and 1 / (1 - p) in one factor
        const int swz = (row >> 2) & 3;
        uint2 uv[4], rv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int eoff = row * 64 + ((g ^ swz) << 4) + 8 * hh;
          uv[g] = *reinterpret_cast<const uint2*>(lds_act + eoff);
          rv[g] = *reinterpret_cast<const uint2*>(lds_res + eoff);
        }
        unsigned yp[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const unsigned uvp[2] = {uv[g].x, uv[g].y}, rvp[2] = {rv[g].x, rv[g].y};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const unsigned ypk = pack_bf16x2(acc[i][4 * g + 2 * j] + bval[4 * g + 2 * j], acc[i][4 * g + 2 * j + 1] + bval[4 * g + 2 * j + 1]);
            const float o0 = __builtin_bit_cast(float, ypk << 16), o1 = __builtin_bit_cast(float, ypk & 0xffff0000u);
            const float v0 = (uvp[j] & 0x7fffu) ? o0 * srow : 0.f, v1 = (uvp[j] & 0x7fff0000u) ? o1 * srow : 0.f;
            yp[2 * g + j] = pack_bf16x2(v0 + __builtin_bit_cast(float, rvp[j] << 16), v1 + __builtin_bit_cast(float, rvp[j] & 0xffff0000u));
          }
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            auto sw = __builtin_amdgcn_permlane32_swap(yp[4 * h2 + d], yp[4 * h2 + 2 + d], false, false);
            yp[4 * h2 + d] = sw[0]; yp[4 * h2 + 2 + d] = sw[1];
          }
        const unsigned vo = (unsigned)tc * pitch_y + (unsigned)hh * 16u;
        __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[0], (int)yp[1], (int)yp[2], (int)yp[3]}, ry, (int)vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[4], (int)yp[5], (int)yp[6], (int)yp[7]}, ry, (int)(vo + 32u), 0, 0);
      }
    } else {
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const int row = 32 * i + r;
      const int tc = t0 + row;
      const int ty = cls + rs * tc;                          // actual output row
      const float keep_row = (ty >= len_out) ? 0.f : 1.f;
      const int swz = (row >> 2) & 3;
      unsigned yp[8], up[8];
      {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (float)(T)(acc[i][4 * g + k] + bval[4 * g + k]);
        const int eoff = row * 64 + ((g ^ swz) << 4) + 8 * hh;
        if (has_epi_act) {
          const bf16x4 uv = *reinterpret_cast<const bf16x4*>(lds_act + eoff);
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = ((float)uv[k] != 0.f) ? o[k] * p.drop_scale : 0.f;
        }
        if (has_res) {
          const bf16x4 rv = *reinterpret_cast<const bf16x4*>(lds_res + eoff);
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = fmaf(o[k], keep_row, (float)rv[k]);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] *= keep_row;
        }
        yp[2 * g] = pack_bf16x2(o[0], o[1]);
        yp[2 * g + 1] = pack_bf16x2(o[2], o[3]);
        if (has_act_out) {
          const int col = col0 + 8 * g;
          const int site = col / p.site_width;
          const unsigned key = site_key(p, site);
          const unsigned long long base = ((unsigned long long)b * p.Ty + ty) * p.site_width + (col - site * p.site_width);
#pragma unroll
          for (int k = 0; k < 4; k += 2) {
            const unsigned h = fmix32((unsigned)((base + k) >> 1) * 0x9E3779B1u + key);
            const bool k0 = (h & 0xFFFFu) >= p.drop_thresh16, k1 = (h >> 16) >= p.drop_thresh16;
            up[2 * g + (k >> 1)] = pack_bf16x2((k0 && o[k] > 0.f) ? o[k] * p.drop_scale : 0.f,
                                               (k1 && o[k + 1] > 0.f) ? o[k + 1] * p.drop_scale : 0.f);
          }
        }
      }
      }
      // lanes r and r + 32 hold channels {0-3, 8-11, 16-19, 24-27} and {4-7, 12-15, 20-23, 28-31} of the same
      // row: swap so that lane r owns 0-7 | 16-23 and lane r + 32 owns 8-15 | 24-31 (16-byte pieces)
      auto pair_up = [&](unsigned* v) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            auto sw = __builtin_amdgcn_permlane32_swap(v[4 * h2 + d], v[4 * h2 + 2 + d], false, false);
            v[4 * h2 + d] = sw[0]; v[4 * h2 + 2 + d] = sw[1];
          }
      };
      if (has_y) {
        pair_up(yp);
        const unsigned vo = (unsigned)tc * pitch_y + (unsigned)hh * 16u;
        __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[0], (int)yp[1], (int)yp[2], (int)yp[3]}, ry, (int)vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[4], (int)yp[5], (int)yp[6], (int)yp[7]}, ry, (int)(vo + 32u), 0, 0);
      }
      if (has_act_out) {
        pair_up(up);
        const unsigned vo = (unsigned)tc * pitch_u + (unsigned)hh * 16u;
        __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)up[0], (int)up[1], (int)up[2], (int)up[3]}, ru, (int)vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)up[4], (int)up[5], (int)up[6], (int)up[7]}, ru, (int)(vo + 32u), 0, 0);
      }
    }
    }
    WS_T(c7);
    WS_ACC(0, c0, c1); WS_ACC(1, c1, c2); WS_ACC(3, c2, c4); WS_ACC(4, c4, c5); WS_ACC(5, c5, c6); WS_ACC(6, c6, c7); WS_ACC(7, c0, c0 + 1);
  }
}

// ------------------------------------------------------------------------------------------------
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ------------------------------------------------------------------------------------------------
// Weight-stationary kernel with the epilogue of tile i-1 hidden in the MFMA gaps of tile i (K2 forward: only the
// activated output u = relu(dropout(y)) is written).  One wave per SIMD issues strictly in order, so the epilogue of
// conv_ws_kernel (~1000 VALU instructions per tile at 4 cycles each) runs while the matrix core idles; but an
// MFMA holds the vector issue port for only 8 of its 32 cycles, which leaves room for ~5 other instructions per MFMA.
// Here the finished accumulators are packed to bf16 (32 registers) and the rest of the epilogue -- dropout hash,
// ReLU, scaling, lane pairing, stores -- is cut into micro-steps of 2..10 instructions, one behind each MFMA of the
// next tile (sched_barrier keeps them where they are written).  Taps 0..7 of the weight block are pinned to
// AccVGPRs.  Same arithmetic as conv_ws_kernel: bit-identical outputs.
template <int NTAPS>
__global__ __launch_bounds__(WS_NT) void conv_ws_pipe_kernel(ConvArgs p, const __bf16* __restrict__ zero_page,
                                                             int tiles_per_wg, int buf_bytes) {
  constexpr int BM = WS_BM, BN = 128, KC = 128, NT = WS_NT, MW = BM / 32, ROWB = KC * 2;
  constexpr int NSTEP = NTAPS * (KC / 16), NGAP = NSTEP * MW;
  constexpr int PH_UNIT = 12, PH_ROW = 4 * PH_UNIT + 2, PH_TILE = MW * PH_ROW;
  constexpr int AGPR_TAPS = NTAPS < 8 ? NTAPS : 8;
  static_assert(MW == 4, "four accumulators");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int lrow = lane >> 4, lch = lane & 15;
  const int n0 = blockIdx.y * BN;
  const int rs = p.rs;

  const int ntiles = p.tiles_per_batch * p.B * rs;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = wg * tiles_per_wg;
  const int tile_end = min(ntiles, tile_begin + tiles_per_wg);
  if (tile_begin >= tile_end) return;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  bf16x8 wfrag[NTAPS][KC / 16];
  {
    const int co = n0 + wave * 32 + r;
#pragma unroll
    for (int s = 0; s < NTAPS; ++s) {
      const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.w) + ((size_t)s * p.Cout + co) * ROWB;
#pragma unroll
      for (int kk = 0; kk < KC / 16; ++kk)
        wfrag[s][kk] = *reinterpret_cast<const bf16x8*>(wrow + (((2 * kk + hh) ^ (co & 15)) << 4));
    }
  }
  // accumulator element 4g + k of a lane = output channel col0 + 8g + k (this lane's row: 32 i + r)
  const int col0 = n0 + wave * 32 + 4 * hh;
  float bval[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bval[e] = p.bias ? p.bias[col0 + 8 * (e >> 2) + (e & 3)] : 0.f;
  unsigned keyg[4], cshalf[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int col = col0 + 8 * g;
    const int site = col / p.site_width;
    keyg[g] = site_key(p, site);
    cshalf[g] = (unsigned)(col - site * p.site_width) >> 1;
  }
  const int rows_in = BM + (NTAPS - 1) * p.dil;
  const int rows_pad = (rows_in + 3) & ~3;

  auto decode = [&](int tile, int& b, int& cls, int& t0) {
    const int bb = tile / p.tiles_per_batch;
    b = bb / rs; cls = bb - b * rs;
    t0 = (tile - bb * p.tiles_per_batch) * BM;
  };
  // buffer addressing as in conv_ws_kernel: one per-lane offset, a scalar per tile, out-of-range rows read as zero
  const unsigned pitch_x = (unsigned)p.ldx * rs * 2u;
  const unsigned voff_a0 = (unsigned)(4 * wave + lrow) * pitch_x + (unsigned)((lch ^ ((4 * wave + lrow) & 15)) << 4);
  const int ngroups = rows_pad >> 2;
  auto stage_a = [&](int tile, int buf) {
    int b, cls, t0;
    decode(tile, b, cls, t0);
    const int len_full = p.lens_in ? min(scalar_load_i32(p.lens_in + b), p.Tin) : p.Tin;
    const int len_in = max(0, (len_full - cls + rs - 1) / rs);
    const __amdgpu_buffer_rsrc_t rx = ws_rsrc(p.x, ((long long)b * p.x_bs + (long long)cls * p.ldx) * 2, (unsigned)len_in * pitch_x);
    unsigned vo = voff_a0 + (unsigned)(t0 - p.pad) * pitch_x;
    unsigned char* dst = smem + (size_t)buf * buf_bytes + wave * 1024;
    for (int g = wave; g < ngroups; g += NT / 64) {
      ws_dma16(rx, vo, dst);
      vo += 16u * pitch_x; dst += 4096;
    }
  };

  // ---- state of the tile whose epilogue is pending
  unsigned pc[MW][8];                 // bf16 pairs of (acc + bias): pc[i][2g + h] = elements 4g + 2h, 4g + 2h + 1
  int p_b = 0, p_cls = 0, p_t0 = 0, p_Tc = 0, p_len = 0;
  __amdgpu_buffer_rsrc_t p_ru = ws_rsrc(p.y_act, 0, 0);      // output rows of the pending item (set at the hand-over)
  // ---- scratch of the micro-steps (live across MFMAs)
  float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f, u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f, keepf = 1.f;
  unsigned h0 = 0, h1 = 0, rh = 0, up[8];
  int ty = 0;
  auto fbits = [](unsigned v) { return __builtin_bit_cast(float, v); };
  auto epi = [&](auto M) {           // micro-step M of the pending epilogue (M is a compile-time constant)
    constexpr int m = decltype(M)::value;
    constexpr int i = m / PH_ROW, rem = m % PH_ROW;
    if constexpr (rem < 4 * PH_UNIT) {
      constexpr int g = rem / PH_UNIT, ph = rem % PH_UNIT;
      if constexpr (ph == 0) {
        if constexpr (g == 0) {
          const int tc = p_t0 + 32 * i + r;
          ty = (tc < p_Tc) ? p_cls + rs * tc : -1;
          keepf = (ty >= p_len) ? 0.f : 1.f;
          rh = (unsigned)((((unsigned long long)p_b * p.Ty + (unsigned)max(ty, 0)) * (unsigned)p.site_width) >> 1);
        }
      } else if constexpr (ph == 1) {
        o0 = fbits(pc[i][2 * g] << 16) * keepf; o1 = fbits(pc[i][2 * g] & 0xffff0000u) * keepf;
      } else if constexpr (ph == 2) {
        o2 = fbits(pc[i][2 * g + 1] << 16) * keepf; o3 = fbits(pc[i][2 * g + 1] & 0xffff0000u) * keepf;
      } else if constexpr (ph == 3) {
        h0 = (rh + cshalf[g]) * 0x9E3779B1u + keyg[g]; h1 = h0 + 0x9E3779B1u;
      } else if constexpr (ph == 4) {
        h0 ^= h0 >> 16; h0 *= 0x85EBCA6Bu; h1 ^= h1 >> 16; h1 *= 0x85EBCA6Bu;
      } else if constexpr (ph == 5) {
        h0 ^= h0 >> 13; h0 *= 0xC2B2AE35u; h1 ^= h1 >> 13; h1 *= 0xC2B2AE35u;
      } else if constexpr (ph == 6) {
        h0 ^= h0 >> 16; h1 ^= h1 >> 16;
      } else if constexpr (ph == 7) {
        u0 = ((h0 & 0xFFFFu) >= p.drop_thresh16 && o0 > 0.f) ? o0 * p.drop_scale : 0.f;
      } else if constexpr (ph == 8) {
        u1 = ((h0 >> 16) >= p.drop_thresh16 && o1 > 0.f) ? o1 * p.drop_scale : 0.f;
      } else if constexpr (ph == 9) {
        u2 = ((h1 & 0xFFFFu) >= p.drop_thresh16 && o2 > 0.f) ? o2 * p.drop_scale : 0.f;
      } else if constexpr (ph == 10) {
        u3 = ((h1 >> 16) >= p.drop_thresh16 && o3 > 0.f) ? o3 * p.drop_scale : 0.f;
      } else {
        up[2 * g] = pack_bf16x2(u0, u1); up[2 * g + 1] = pack_bf16x2(u2, u3);
      }
    } else if constexpr (rem == 4 * PH_UNIT) {
      // lanes r / r + 32 hold channels {0-3, 8-11, ..} / {4-7, 12-15, ..} of one row -> 16-byte pieces
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          auto sw = __builtin_amdgcn_permlane32_swap(up[4 * h2 + d], up[4 * h2 + 2 + d], false, false);
          up[4 * h2 + d] = sw[0]; up[4 * h2 + 2 + d] = sw[1];
        }
    } else {
      if (ty >= 0) {      // 32-bit offset from the pending item's base (a scalar); the V# covers the whole output tensor
        const unsigned vo = (unsigned)ty * ((unsigned)p.ldya * 2u) + (unsigned)hh * 16u;
        __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)up[0], (int)up[1], (int)up[2], (int)up[3]}, p_ru, (int)vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)up[4], (int)up[5], (int)up[6], (int)up[7]}, p_ru, (int)(vo + 32u), 0, 0);
      }
    }
  };

  stage_a(tile_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) pc[i][e] = 0;
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    int b, cls, t0;
    decode(tile, b, cls, t0);
    __syncthreads();                 // tile `tile` is in LDS for every wave; every wave is done with the other buffer
    if (tile + 1 < tile_end) stage_a(tile + 1, buf ^ 1);

    f32x16 acc[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const unsigned abase = lds_base + (unsigned)buf * (unsigned)buf_bytes;
    unsigned tap_base[NTAPS];
#pragma unroll
    for (int s = 0; s < NTAPS; ++s) {
      const int ar = r + s * p.dil;
      tap_base[s] = abase + ar * ROWB + ((hh ^ (ar & 15)) << 4);
    }
    auto frag_addr = [&](int q) -> unsigned { return tap_base[q / (KC / 16)] ^ (32u * (q % (KC / 16))); };
    bf16x8 afr[2][MW];

    // the MFMA pipeline of conv_ws_kernel with micro-step 4q + i of the pending epilogue behind MFMA (q, i).  (The
    // first tile runs the micro-steps on an empty pending tile -- p_Tc = 0 suppresses its stores -- so that there is
    // ONE copy of the loop: with two, the allocator no longer keeps the pinned weights in AccVGPRs.)
    {
      const unsigned ap0 = frag_addr(0);
#pragma unroll
      for (int i = 0; i < MW; ++i)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afr[0][i]) : "v"(ap0), "n"(i * 32 * ROWB));
    }
    static_for<0, NSTEP>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      if constexpr (q + 1 < NSTEP) {
        const unsigned ap = frag_addr(q + 1);
#pragma unroll
        for (int i = 0; i < MW; ++i)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afr[(q + 1) & 1][i]) : "v"(ap), "n"(i * 32 * ROWB));
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      static_for<0, MW>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (q == 0) asm volatile("s_nop 4" : "+v"(acc[i]));
        if constexpr (q / (KC / 16) < AGPR_TAPS)
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(wfrag[q / (KC / 16)][q % (KC / 16)]), "v"(afr[q & 1][i]));
        else
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(wfrag[q / (KC / 16)][q % (KC / 16)]), "v"(afr[q & 1][i]));
        if constexpr (4 * q + i < PH_TILE) {
          __builtin_amdgcn_sched_barrier(0);
          epi(std::integral_constant<int, 4 * q + i>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    });
    static_for<(NGAP < PH_TILE ? NGAP : PH_TILE), PH_TILE>(epi);   // micro-steps that did not fit behind this tile's MFMAs
    // the hazard recogniser does not see MFMAs inside asm: let the last ones drain before VALU reads acc
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile has landed; older stores retired
    // hand the tile over: y = bf16(acc + bias), packed
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int e = 0; e < 16; e += 2) pc[i][e >> 1] = pack_bf16x2(acc[i][e] + bval[e], acc[i][e + 1] + bval[e + 1]);
    p_b = b; p_cls = cls; p_t0 = t0;
    p_ru = ws_rsrc(p.y_act, ((long long)b * p.ya_bs + n0 + wave * 32) * 2, (unsigned)p.Ty * ((unsigned)p.ldya * 2u));
    p_Tc = (p.Tout - cls + rs - 1) / rs;
    p_len = p.lens_out ? scalar_load_i32(p.lens_out + b) : 0x7fffffff;
  }
  static_for<0, PH_TILE>(epi);                          // epilogue of the last tile
}

// ------------------------------------------------------------------------------------------------
// Weight-stationary kernel with TWO waves per SIMD (3 and 5 taps: 96 / 160 weight registers per wave leave room for a
// second wave in the 512-entry register file).  conv_ws_kernel runs one wave per SIMD, and a wave issues in order: its
// epilogue (VALU), its LDS-DMA issue and its barrier wait are all serial with its own MFMAs -- at 3 taps the matrix pipe
// is busy 30 % of the time (profiles/r02_pmc_mfma.txt).  Here a 512-thread workgroup splits the 128-row tile into two
// 64-row halves: waves w and w + 4 share a SIMD, own the SAME 32 output channels (both hold that weight slice) and
// different halves of the rows.  The two run the same program with one barrier per tile, but STAGGERED
// (MI355X_MICROARCH.md, "Two waves per SIMD", item 9): waves 0-3 multiply tile n and then write it out, waves 4-7 first
// write out tile n-1 (its sums stay in the accumulators across the barrier) and then multiply tile n -- so on every SIMD
// one wave's MFMA segment always sits beside its partner's epilogue segment (matrix beside VALU / memory, the pairing
// that nets).  Same arithmetic per output as conv_ws_kernel: bit-identical results.
// MODE 1 = activated output only (K2 forward), 2 = y with activation-gradient mask and residual (K2 data gradient).
constexpr int WS2_NT = 512;
// conv_ws2_kernel is dispatched up to this tap count.  Five taps compile (4 taps in AccVGPRs + 1 in VGPRs) but spill 7-24
// registers to scratch in the epilogue and run slower than the one-wave kernels (r03: 638 vs 490 us at the top level).
constexpr int WS2_MAX_TAPS = 3;

template <int NTAPS, int MODE>
__global__ __launch_bounds__(WS2_NT) void conv_ws2_kernel(ConvArgs p, const __bf16* __restrict__ zero_page,
                                                          int tiles_per_wg, int buf_bytes) {
  constexpr int BM = WS_BM, BN = 128, KC = 128, MW = 2, ROWB = KC * 2;
  constexpr int NSTEP = NTAPS * (KC / 16), NG = 5;          // NG: 4-row staging groups per wave (<= 160 rows per tile)
  static_assert(MODE == 1 || MODE == 2, "two epilogues");
  // Register budget at two waves per SIMD: 256 per lane, which the compiler splits 128 AccVGPRs / 128 VGPRs as soon as a
  // kernel names AccVGPRs (telling it to take 160 through an "a159" clobber leaves the VGPR side at 128 and the kernel
  // at one wave per SIMD): four taps (128 registers) are pinned there, a fifth lives in VGPRs, and what else is live
  // (32 accumulators, 24 fragment registers, offsets) has to fit beside it -- the bias values come from LDS for that.
  constexpr int AGPR_TAPS = NTAPS < 4 ? NTAPS : 4;
  static_assert(NTAPS <= 5, "weights of more than five taps do not fit two waves per SIMD");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave & 3, rhalf = wave >> 2;              // column group (32 output channels), row half = phase
  const int r = lane & 31, hh = lane >> 5;
  const int n0 = blockIdx.y * BN;
  const int rs = p.rs;

  const int ntiles = p.tiles_per_batch * p.B * rs;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = wg * tiles_per_wg;
  const int tile_end = min(ntiles, tile_begin + tiles_per_wg);
  if (tile_begin >= tile_end) return;

  // fp32 bias of the 128 output channels, then (MODE 2) the epilogue-operand slices: per wave [64 rows][64 B] (its 32
  // channels of its row half), 16-byte chunks swizzled
  float* lds_bias = reinterpret_cast<float*>(smem + 2 * (size_t)buf_bytes);
  unsigned char* lds_res = smem + 2 * (size_t)buf_bytes + 1024 + wave * (WS_EPI / 8);
  unsigned char* lds_act = lds_res + WS_EPI;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  if (tid < BN) lds_bias[tid] = p.bias ? p.bias[n0 + tid] : 0.f;       // visible after the first barrier of the tile loop

  bf16x8 wfrag[NTAPS][KC / 16];
  {
    const int co = n0 + cw * 32 + r;
#pragma unroll
    for (int s = 0; s < NTAPS; ++s) {
      const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.w) + ((size_t)s * p.Cout + co) * ROWB;
#pragma unroll
      for (int kk = 0; kk < KC / 16; ++kk)
        wfrag[s][kk] = *reinterpret_cast<const bf16x8*>(wrow + (((2 * kk + hh) ^ (co & 15)) << 4));
    }
  }
  // accumulator element 4g + k of a lane = output channel col0 + 8g + k (this lane's row: 64 rhalf + 32 i + r)
  const int col0 = n0 + cw * 32 + 4 * hh;
  // dropout (MODE 1): hash input of the pair (row, col0 + 8g + 2j ..+1) = rowh + kc[g] + j C, with
  // rowh = ((b Ty + ty) site_width / 2) C (mod 2^32) = scalar part + lane part, kc[g] = (column / 2) C + key
  const unsigned HC = 0x9E3779B1u;
  const unsigned swh = (unsigned)p.site_width >> 1;
  unsigned kc[4];
  unsigned rowh_lane = 0;
  const unsigned thr_m1 = (p.drop_thresh16 - 1u) * 0x00010001u;
  if (MODE == 1) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = col0 + 8 * g;
      const int site = col / p.site_width;
      kc[g] = ((unsigned)(col - site * p.site_width) >> 1) * HC + site_key(p, site);
    }
    rowh_lane = HC * swh * (unsigned)(rs * r);
  }
  const int rows_in = BM + (NTAPS - 1) * p.dil;
  const int ngroups = (rows_in + 3) >> 2;

  // ---- byte pitches of the class-domain rows and the tile-invariant per-lane offsets
  const unsigned pitch_x = (unsigned)p.ldx * rs * 2u;
  unsigned voff_a[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int row = 4 * (wave + 8 * j) + (lane >> 4);
    voff_a[j] = (unsigned)row * pitch_x + (unsigned)(((lane & 15) ^ (row & 15)) << 4);
  }
  const unsigned pitch_o = (MODE == 2 ? (unsigned)p.ldy : (unsigned)p.ldya) * rs * 2u;    // output rows
  const unsigned voff_o = (unsigned)(64 * rhalf + r) * pitch_o + (unsigned)hh * 16u;      // row block i: + 32 i pitch_o
  const unsigned pitch_r = (unsigned)p.ldr * rs * 2u, pitch_g = (unsigned)p.ldgh * rs * 2u;
  const int elr = 64 * rhalf + (lane >> 2);                                                 // + 16 q: staged row of the slices
  const unsigned echunk = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);                // chunk (lane & 3) ^ ((row >> 2) & 3)

  auto decode = [&](int tile, int& b, int& cls, int& t0) {
    const int bb = tile / p.tiles_per_batch;
    b = bb / rs; cls = bb - b * rs;
    t0 = (tile - bb * p.tiles_per_batch) * BM;
  };
  auto stage_a = [&](int tile, int buf) {
    int b, cls, t0;
    decode(tile, b, cls, t0);
    const int len_full = p.lens_in ? min(scalar_load_i32(p.lens_in + b), p.Tin) : p.Tin;
    const int len_in = max(0, (len_full - cls + rs - 1) / rs);
    const __amdgpu_buffer_rsrc_t rx = ws_rsrc(p.x, ((long long)b * p.x_bs + (long long)cls * p.ldx) * 2, (unsigned)len_in * pitch_x);
    const unsigned s0 = (unsigned)(t0 - p.pad) * pitch_x;       // rows before the item wrap to huge offsets: out of range, zero
    unsigned char* dst = smem + (size_t)buf * buf_bytes + wave * 1024;
#pragma unroll
    for (int j = 0; j < NG; ++j)
      if (wave + 8 * j < ngroups) ws_dma16(rx, voff_a[j] + s0, dst + j * 8192);
  };
  // this wave's slice of an epilogue operand: one DMA instruction = 16 rows x 64 B; slot c of local row n holds the
  // 16-byte chunk c ^ ((n >> 2) & 3) of the slice
  auto stage_epi = [&](const void* base, long long bs, int ld, unsigned pitch, int b, int cls, int t0, int Tc, unsigned char* dst) {
    const __amdgpu_buffer_rsrc_t re = ws_rsrc(base, ((long long)b * bs + (long long)cls * ld + n0 + cw * 32) * 2, (unsigned)Tc * pitch);
#pragma unroll
    for (int q = 0; q < 4; ++q) ws_dma16(re, (unsigned)(t0 + elr + 16 * q) * pitch + echunk, dst + q * 1024);
  };

  f32x16 acc[MW];
  // ---- epilogue of one tile half straight from the accumulators (the arithmetic of conv_ws_kernel, written for few VALU
  //      instructions: the epilogue, not the matrix pipe, bounds this kernel at 3 taps)
  auto epilogue = [&](int b, int cls, int t0, int Tc) {
    const int len_out = p.lens_out ? scalar_load_i32(p.lens_out + b) : 0x7fffffff;
    const __amdgpu_buffer_rsrc_t ry =
        MODE == 2 ? ws_rsrc(p.y, ((long long)b * p.y_bs + (long long)cls * p.ldy + n0 + cw * 32) * 2, (unsigned)Tc * pitch_o)
                  : ws_rsrc(p.y_act, ((long long)b * p.ya_bs + (long long)cls * p.ldya + n0 + cw * 32) * 2, (unsigned)Tc * pitch_o);
    const unsigned so = (unsigned)t0 * pitch_o;
    const unsigned rowh_s = MODE == 1 ? HC * swh * ((unsigned)b * (unsigned)p.Ty + (unsigned)cls + (unsigned)rs * (unsigned)(t0 + 64 * rhalf)) : 0u;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const int lr = 32 * i + r;                                // row inside this wave's half
      const int ty = cls + rs * (t0 + 64 * rhalf + lr);         // actual output row
      const float srow = (ty >= len_out) ? 0.f : p.drop_scale;  // row mask and 1 / (1 - p) in one factor
      const int swz = (lr >> 2) & 3;
      unsigned yp[8];
      const unsigned rowh = rowh_s + rowh_lane + HC * swh * (unsigned)(rs * 32 * i);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lds_bias + cw * 32 + 4 * hh + 8 * g);
        unsigned uvp[2] = {0, 0}, rvp[2] = {0, 0};
        if (MODE == 2) {
          const int eoff = lr * 64 + ((g ^ swz) << 4) + 8 * hh;
          const uint2 uv = *reinterpret_cast<const uint2*>(lds_act + eoff);
          const uint2 rv = *reinterpret_cast<const uint2*>(lds_res + eoff);
          uvp[0] = uv.x; uvp[1] = uv.y; rvp[0] = rv.x; rvp[1] = rv.y;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // y = bf16(acc + bias), two elements per conversion
          const unsigned ypk = pack_bf16x2(acc[i][4 * g + 2 * j] + bv[2 * j], acc[i][4 * g + 2 * j + 1] + bv[2 * j + 1]);
          const float o0 = __builtin_bit_cast(float, ypk << 16), o1 = __builtin_bit_cast(float, ypk & 0xffff0000u);
          if (MODE == 1) {
            // u = relu(dropout(y)): scale (row mask folded in), round, relu + keep mask on the packed pair
            unsigned w = pk_relu_bf16(pack_bf16x2(o0 * srow, o1 * srow));
            if (p.drop_thresh16) w &= pk_keep_mask(fmix32(rowh + kc[g] + (unsigned)j * HC), thr_m1);
            yp[2 * g + j] = w;
          } else {
            // dx = y * scale * [u != 0] * row mask + residual
            const unsigned u2 = uvp[j], r2 = rvp[j];
            const float v0 = (u2 & 0x7fffu) ? o0 * srow : 0.f, v1 = (u2 & 0x7fff0000u) ? o1 * srow : 0.f;
            yp[2 * g + j] = pack_bf16x2(v0 + __builtin_bit_cast(float, r2 << 16), v1 + __builtin_bit_cast(float, r2 & 0xffff0000u));
          }
        }
      }
      // lanes r and r + 32 hold channels {0-3, 8-11, 16-19, 24-27} and {4-7, 12-15, 20-23, 28-31} of the same
      // row: swap so that lane r owns 0-7 | 16-23 and lane r + 32 owns 8-15 | 24-31 (16-byte pieces)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          auto sw = __builtin_amdgcn_permlane32_swap(yp[4 * h2 + d], yp[4 * h2 + 2 + d], false, false);
          yp[4 * h2 + d] = sw[0]; yp[4 * h2 + 2 + d] = sw[1];
        }
      const unsigned vo = voff_o + so + (unsigned)(32 * i) * pitch_o;    // rows >= Tc are out of range: dropped by the hardware
      __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[0], (int)yp[1], (int)yp[2], (int)yp[3]}, ry, (int)vo, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[4], (int)yp[5], (int)yp[6], (int)yp[7]}, ry, (int)(vo + 32u), 0, 0);
    }
  };

  stage_a(tile_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int pb = 0, pcls = 0, pt0 = 0, pTc = 0;        // the tile whose sums waves 4-7 still hold
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    int b, cls, t0;
    decode(tile, b, cls, t0);
    const int Tc = (p.Tout - cls + rs - 1) / rs;
    // tile `tile` is in LDS for every wave (each waited for its own DMAs before arriving here) and every wave is done
    // reading the other buffer
    WS_T(c0);
    __syncthreads();
    WS_T(c1);
    if (tile + 1 < tile_end) stage_a(tile + 1, buf ^ 1);
    WS_T(c2);
    if (rhalf == 1 && tile > tile_begin) epilogue(pb, pcls, pt0, pTc);
    WS_T(c3);
    if (MODE == 2) {                         // this wave's slices are free: its previous epilogue is done
      stage_epi(p.res, p.res_bs, p.ldr, pitch_r, b, cls, t0, Tc, lds_res);
      stage_epi(p.gate_h, p.gh_bs, p.ldgh, pitch_g, b, cls, t0, Tc, lds_act);
    }
    WS_T(c4);
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    // MFMA pipeline over the NTAPS x 8 k-steps: the fragments of step q + 2 are requested before the MFMAs of step q
    // (two MFMAs = 64 cycles per step would not cover the LDS latency with a distance of one)
    const unsigned abase = lds_base + (unsigned)buf * (unsigned)buf_bytes + (unsigned)(64 * rhalf) * ROWB;
    unsigned tap_base[NTAPS];
#pragma unroll
    for (int s = 0; s < NTAPS; ++s) {
      const int ar = r + s * p.dil;                 // rows ar + 32 i (+ 64 rhalf) share the swizzle term (ar & 15)
      tap_base[s] = abase + ar * ROWB + ((hh ^ (ar & 15)) << 4);
    }
    auto frag_addr = [&](int q) -> unsigned { return tap_base[q / (KC / 16)] ^ (32u * (q % (KC / 16))); };
    bf16x8 afr[3][MW];
#pragma unroll
    for (int q0 = 0; q0 < 2; ++q0) {
      const unsigned ap = frag_addr(q0);
#pragma unroll
      for (int i = 0; i < MW; ++i)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afr[q0][i]) : "v"(ap), "n"(i * 32 * ROWB));
    }
    static_for<0, NSTEP>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      if constexpr (q + 2 < NSTEP) {
        const unsigned ap = frag_addr(q + 2);
#pragma unroll
        for (int i = 0; i < MW; ++i)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afr[(q + 2) % 3][i]) : "v"(ap), "n"(i * 32 * ROWB));
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      } else if constexpr (q + 1 < NSTEP) {
        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      // (clang's implicit capture in a generic lambda misses a variable that is only named in asm operands)
      constexpr int i0 = q - q, i1 = i0 + 1;
      (void)&acc; (void)&wfrag;       // odr-use outside asm: forces the capture
      // first use of an accumulator: VALU write -> MFMA SrcC needs wait states the compiler cannot see through asm
      if constexpr (q == 0) asm volatile("s_nop 4" : "+v"(acc[i0]), "+v"(acc[i1]));
      if constexpr (q / (KC / 16) < AGPR_TAPS) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i0]) : "a"(wfrag[q / (KC / 16)][q % (KC / 16)]), "v"(afr[q % 3][i0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i1]) : "a"(wfrag[q / (KC / 16)][q % (KC / 16)]), "v"(afr[q % 3][i1]));
      } else {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i0]) : "v"(wfrag[q / (KC / 16)][q % (KC / 16)]), "v"(afr[q % 3][i0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i1]) : "v"(wfrag[q / (KC / 16)][q % (KC / 16)]), "v"(afr[q % 3][i1]));
      }
    });
    static_assert(MW == 2, "two accumulators per wave");
    // the hazard recogniser does not see MFMAs inside asm: let the last ones drain before VALU reads acc
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
    WS_T(c5);
    // next tile + this tile's epilogue operands have landed; older stores retired
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WS_T(c6);
    if (rhalf == 0) epilogue(b, cls, t0, Tc);
    WS_T(c7);
    WS_ACC(0, c0, c1); WS_ACC(1, c1, c2); WS_ACC(2, c2, c3); WS_ACC(3, c3, c4); WS_ACC(4, c4, c5); WS_ACC(5, c5, c6);
    WS_ACC(6, c6, c7); WS_ACC(7, c0, c0 + 1);
    pb = b; pcls = cls; pt0 = t0; pTc = Tc;
  }
  if (rhalf == 1) epilogue(pb, pcls, pt0, pTc);
}

template <int NTAPS, int MODE>
static void launch_ws_mode(const ConvArgs& p, const void* zero_page, dim3 grid, size_t lds, int tpw, int buf_bytes,
                           hipStream_t stream) {
  (void)hipFuncSetAttribute((const void*)conv_ws_kernel<NTAPS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  conv_ws_kernel<NTAPS, MODE><<<grid, WS_NT, lds, stream>>>(p, (const __bf16*)zero_page, tpw, buf_bytes);
}

template <int NTAPS>
static void launch_ws(const ConvArgs& p, const void* zero_page, dim3 grid, size_t lds, int tpw, int buf_bytes,
                      hipStream_t stream) {
  const bool y = p.y != nullptr, ao = p.act_out != 0, res = p.res != nullptr, ea = p.epi_act != 0;
  static const bool no_pipe = getenv("SMT_CONV_NO_PIPE") != nullptr;
  if constexpr (NTAPS <= WS2_MAX_TAPS) {   // two waves per SIMD, staggered (conv_ws2_kernel)
    static const bool no_ws2 = getenv("SMT_CONV_NO_WS2") != nullptr;
    if (!no_ws2 && !y && ao && !res && !ea) {
      (void)hipFuncSetAttribute((const void*)conv_ws2_kernel<NTAPS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      conv_ws2_kernel<NTAPS, 1><<<grid, WS2_NT, 2 * (size_t)buf_bytes + 1024, stream>>>(p, (const __bf16*)zero_page, tpw, buf_bytes);
      return;
    }
    if (!no_ws2 && y && !ao && res && ea) {
      (void)hipFuncSetAttribute((const void*)conv_ws2_kernel<NTAPS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      conv_ws2_kernel<NTAPS, 2><<<grid, WS2_NT, lds + 1024, stream>>>(p, (const __bf16*)zero_page, tpw, buf_bytes);
      return;
    }
  }
  if (!y && ao && !res && !ea && !no_pipe) {
    (void)hipFuncSetAttribute((const void*)conv_ws_pipe_kernel<NTAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    conv_ws_pipe_kernel<NTAPS><<<grid, WS_NT, lds, stream>>>(p, (const __bf16*)zero_page, tpw, buf_bytes);
  } else if (!y && ao && !res && !ea) launch_ws_mode<NTAPS, 1>(p, zero_page, grid, lds, tpw, buf_bytes, stream);
  else if (y && !ao && res && ea) launch_ws_mode<NTAPS, 2>(p, zero_page, grid, lds, tpw, buf_bytes, stream);
  else launch_ws_mode<NTAPS, 0>(p, zero_page, grid, lds, tpw, buf_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// 1x1 conv with a folded second 1x1 term (bf16, C_in == 128, C_in2 == 64):
//   y = W u + b  +  W2 x2 + b2
// K3 of GatedHiFiBlock adds the branch input h1 = K1(x) + b1 (resnet.py:226).  K1 is linear, so the residual is
// recomputed here from the 64-channel block input instead of being written by K1 and read back as a 128-channel
// tensor: per row 384 B in / 256 B out instead of 512 B in / 256 B out here and 256 B less written by K1.
// Same structure as conv1x1_bwd_kernel: persistent workgroups, both operand tiles through an LDS-DMA double
// buffer, transposed MFMA tiles (A = weights in registers, B = rows) stored straight from registers.
constexpr int FO_ROWS = 128, FO_NT = 512, FO_U = FO_ROWS * 256, FO_X = FO_ROWS * 128, FO_STAGE = FO_U + FO_X;

__global__ __launch_bounds__(FO_NT) void conv1x1_fold_kernel(ConvArgs p, const __bf16* __restrict__ zero_page,
                                                              int tiles_per_wg) {
  typedef __bf16 T;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // 2 x [u tile 32 KiB | x2 tile 16 KiB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;          // rows 64 wm.., output channels n0 + 32 wn..
  const int r = lane & 31, hh = lane >> 5;
  const int n0 = blockIdx.y * 128;

  const int ntiles = p.tiles_per_batch * p.B;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = wg * tiles_per_wg;
  const int tile_end = min(ntiles, tile_begin + tiles_per_wg);
  if (tile_begin >= tile_end) return;

  bf16x8 wfrag[8], w2frag[4];
  {
    const int co = n0 + wn * 32 + r;
    const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.w) + (size_t)co * 256;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wfrag[kk] = *reinterpret_cast<const bf16x8*>(wrow + (((2 * kk + hh) ^ (co & 15)) << 4));
    const unsigned char* w2row = reinterpret_cast<const unsigned char*>(p.w2) + (size_t)co * 128;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) w2frag[kk] = *reinterpret_cast<const bf16x8*>(w2row + ((2 * kk + hh) << 4));
  }
  // accumulator element 4g + k of a lane = output channel col0 + 8g + k
  const int col0 = n0 + wn * 32 + 4 * hh;
  float bval[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int c = col0 + 8 * (e >> 2) + (e & 3);
    bval[e] = (p.bias ? p.bias[c] : 0.f) + (p.bias2 ? p.bias2[c] : 0.f);
  }

  auto stage = [&](int tile, int buf) {
    const int b = tile / p.tiles_per_batch;
    const int t0 = (tile - b * p.tiles_per_batch) * FO_ROWS;
    const T* ug = reinterpret_cast<const T*>(p.x) + (long long)b * p.x_bs;
    const T* xg = reinterpret_cast<const T*>(p.x2) + (long long)b * p.x2_bs;
    const int len_in = p.lens_in ? min(p.lens_in[b], p.Tin) : p.Tin;
    const int len_in2 = p.lens_in2 ? min(p.lens_in2[b], p.Tin) : p.Tin;
    unsigned char* base = smem + (size_t)buf * FO_STAGE;
#pragma unroll
    for (int q = 0; q < (FO_ROWS / 4) / (FO_NT / 64); ++q) {     // u: 4 rows x 16 chunks per instruction
      const int g = wave + (FO_NT / 64) * q;
      const int row = 4 * g + (lane >> 4), pos = lane & 15;
      const int t = t0 + row;
      lds_dma16(t < len_in ? ug + (long long)t * p.ldx + ((pos ^ (row & 15)) * 8) : zero_page + pos * 8, base + g * 1024);
    }
#pragma unroll
    for (int q = 0; q < (FO_ROWS / 8) / (FO_NT / 64); ++q) {     // x2: 8 rows x 8 chunks per instruction
      const int g = wave + (FO_NT / 64) * q;
      const int row = 8 * g + (lane >> 3), pos = lane & 7;
      const int t = t0 + row;
      // 128-byte rows: two rows share a 256-byte bank window, chunk c of row n sits at c ^ ((n >> 1) & 7)
      lds_dma16(t < len_in2 ? xg + (long long)t * p.ldx2 + ((pos ^ ((row >> 1) & 7)) * 8) : zero_page + pos * 8,
                base + FO_U + g * 1024);
    }
  };

  stage(tile_begin, 0);
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    const int b = tile / p.tiles_per_batch;
    const int t0 = (tile - b * p.tiles_per_batch) * FO_ROWS;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile has landed
    __syncthreads();                                    // ... for every wave; the other buffer is free again
    if (tile + 1 < tile_end) stage(tile + 1, buf ^ 1);
    const unsigned char* ut = smem + (size_t)buf * FO_STAGE;
    const unsigned char* xt = ut + FO_U;

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + 32 * i + r;
        bf16x8 bv = *reinterpret_cast<const bf16x8*>(ut + row * 256 + (((2 * kk + hh) ^ (row & 15)) << 4));
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[kk], bv, acc[i], 0, 0, 0);
      }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + 32 * i + r;
        bf16x8 bv = *reinterpret_cast<const bf16x8*>(xt + row * 128 + (((2 * kk + hh) ^ ((row >> 1) & 7)) << 4));
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2frag[kk], bv, acc[i], 0, 0, 0);
      }

    T* yg = reinterpret_cast<T*>(p.y) + (long long)b * p.y_bs;
    const int len_out = p.lens_out ? p.lens_out[b] : 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = t0 + wm * 64 + 32 * i + r;
      const float keep_row = (t >= len_out) ? 0.f : 1.f;
      unsigned yp[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (acc[i][4 * g + k] + bval[4 * g + k]) * keep_row;
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
      if (t < p.Tout) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        T* dst = yg + (long long)t * p.ldy + n0 + wn * 32 + 8 * hh;
        *reinterpret_cast<u32x4*>(dst) = u32x4{yp[0], yp[1], yp[2], yp[3]};
        *reinterpret_cast<u32x4*>(dst + 16) = u32x4{yp[4], yp[5], yp[6], yp[7]};
      }
    }
  }
}

static int launch_conv1x1_fold(ConvArgs p, const void* zero_page, hipStream_t stream) {
  p.tiles_per_batch = (p.Tout + FO_ROWS - 1) / FO_ROWS;
  const int ntiles = p.tiles_per_batch * p.B;
  int nwg = std::min(256, std::max(8, (ntiles + 1) / 2));     // one workgroup per CU (96 KiB of LDS)
  nwg = (nwg + 7) / 8 * 8;
  const int tpw = (ntiles + nwg - 1) / nwg;
  dim3 grid((unsigned)nwg, (unsigned)(p.Cout / 128));
  (void)hipFuncSetAttribute((const void*)conv1x1_fold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  conv1x1_fold_kernel<<<grid, FO_NT, 2 * FO_STAGE, stream>>>(p, (const __bf16*)zero_page, tpw);
  SMT_CHECK_LAUNCH("conv1x1_fold");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// K1 of GatedHiFiBlock for all branches at once when only the ACTIVATED output is wanted (bf16, C_in = 64,
// C_out = 512, y == NULL):  u = relu(dropout(W x + b)) -- 128 B in, 1 KiB out per row, an HBM-write-bound
// layer.  Persistent workgroups; the 512 x 64 weight block lives in registers (wave w owns output channels
// 64 w .. 64 w + 63); x tiles come through an LDS-DMA double buffer; transposed MFMA tiles, dropout hash and
// ReLU on the accumulators, v_permlane32_swap pairing, 16-byte stores straight from registers.
constexpr int K1_ROWS = 128, K1_NT = 512, K1_X = K1_ROWS * 128, K1_COUT = 512;

__global__ __launch_bounds__(K1_NT) void conv_k1act_kernel(ConvArgs p, const __bf16* __restrict__ zero_page,
                                                           int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // 2 x [128 rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;

  const int ntiles = p.tiles_per_batch * p.B;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = wg * tiles_per_wg;
  const int tile_end = min(ntiles, tile_begin + tiles_per_wg);
  if (tile_begin >= tile_end) return;

  bf16x8 wfrag[2][4];
  float bval[2][16];
  unsigned keys[2];
  int cs[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int co = wave * 64 + 32 * c + r;
    const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.w) + (size_t)co * 128;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wfrag[c][kk] = *reinterpret_cast<const bf16x8*>(wrow + ((2 * kk + hh) << 4));
    // accumulator element 4g + k of a lane = output channel col0 + 8g + k
    const int col0 = wave * 64 + 32 * c + 4 * hh;
#pragma unroll
    for (int e = 0; e < 16; ++e) bval[c][e] = p.bias ? p.bias[col0 + 8 * (e >> 2) + (e & 3)] : 0.f;
    const int site = col0 / p.site_width;            // a 32-channel tile never straddles a site (site_width % 32 == 0)
    keys[c] = site_key(p, site);
    cs[c] = col0 - site * p.site_width;
  }

  // Round 3: the tile loop owns its vector-memory waits (conv_common.h, conv_k3gate.hip): untracked x tiles through a V#,
  // scalar lens loads, stores through a V# (always issued), and ONE counted wait per tile -- vmcnt(16 stores) -- so the
  // 128 KiB a workgroup writes per tile drains under the next tile instead of in front of it.
  auto decode = [&](int tile, int& b, int& t0) {
    b = __builtin_amdgcn_readfirstlane(tile / p.tiles_per_batch);
    t0 = __builtin_amdgcn_readfirstlane((tile - b * p.tiles_per_batch) * K1_ROWS);
  };
  const unsigned pitch_x = (unsigned)p.ldx * 2u, pitch_u = (unsigned)p.ldya * 2u;
  const int xrow = 8 * wave + (lane >> 3);                        // group wave + 8 q: rows 8 wave + .. + 64 q, swizzle independent of q
  const unsigned xoff0 = (unsigned)xrow * pitch_x + (unsigned)(((lane & 7) ^ ((xrow >> 1) & 7)) << 4);
  auto stage = [&](int tile, int buf) {
    int b, t0;
    decode(tile, b, t0);
    const int len_in = p.lens_in ? min(scalar_load_i32(p.lens_in + b), p.Tin) : p.Tin;
    const UntrackedRsrc rx = untracked_rsrc(p.x, (long long)b * p.x_bs * 2, (unsigned)len_in * pitch_x);
    unsigned vo = xoff0 + (unsigned)t0 * pitch_x;
#pragma unroll
    for (int q = 0; q < (K1_ROWS / 8) / (K1_NT / 64); ++q) {     // 8 rows x 8 chunks per instruction; rows >= len_in read as zero
      // 128-byte rows: two rows share a 256-byte bank window, chunk c of row n sits at c ^ ((n >> 1) & 7)
      untracked_dma16(rx, vo, smem + (size_t)buf * K1_X + (wave + (K1_NT / 64) * q) * 1024);
      vo += 64u * pitch_x;
    }
  };

  stage(tile_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first tile, weights and biases; later tiles: counted wait at the END
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(wfrag[c][kk]));      // (so that the compiler does not re-wait for them in the loop)
#pragma unroll
    for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(bval[c][e]));
  }
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    int b, t0;
    decode(tile, b, t0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave's part of this tile landed; the other buffer is free again
    if (tile + 1 < tile_end) stage(tile + 1, buf ^ 1);
    const unsigned char* xt = smem + (size_t)buf * K1_X;
    const __amdgpu_buffer_rsrc_t ru = ws_rsrc(p.y_act, (long long)b * p.ya_bs * 2, (unsigned)p.Tout * pitch_u);
    const int len_out = p.lens_out ? scalar_load_i32(p.lens_out + b) : 0x7fffffff;
#pragma unroll 1
    for (int i = 0; i < K1_ROWS / 32; ++i) {
      const int row = 32 * i + r;
      f32x16 acc[2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 bv = *reinterpret_cast<const bf16x8*>(xt + row * 128 + (((2 * kk + hh) ^ ((row >> 1) & 7)) << 4));
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[c][kk], bv, acc[c], 0, 0, 0);
      }
      const int t = t0 + row;
      const float srow = (t >= len_out) ? 0.f : p.drop_scale;       // row mask folded into the dropout scale
      const unsigned thr_m1 = (unsigned)(p.drop_thresh16 - 1) * 0x10001u;
      // dropout hash input of element pair (g, k): ((rowbase + cs + 8 g + k) >> 1) * C + key.  rowbase, cs, 8 g and k are
      // all even (site_width % 32 == 0), so the shift distributes and the product is linear mod 2^32: ONE quarter-rate
      // multiply per (row, column block) instead of eight, the rest are adds of compile-time multiples of C
      const unsigned rowhalf = (unsigned)((((unsigned long long)b * p.Ty + t) * p.site_width) >> 1);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        unsigned up[8];
        const unsigned hbase = (rowhalf + (unsigned)(cs[c] >> 1)) * 0x9E3779B1u + keys[c];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            // h = bf16(acc + bias), two elements per conversion; u = relu(dropout(h)): scale, round, then ReLU and the keep
            // mask on the packed pair (same arithmetic, element by element, as the generic epilogue)
            const unsigned hp = pack_bf16x2(acc[c][4 * g + 2 * j] + bval[c][4 * g + 2 * j], acc[c][4 * g + 2 * j + 1] + bval[c][4 * g + 2 * j + 1]);
            unsigned w = pk_relu_bf16(pack_bf16x2(__builtin_bit_cast(float, hp << 16) * srow, __builtin_bit_cast(float, hp & 0xffff0000u) * srow));
            if (p.drop_thresh16) w &= pk_keep_mask(fmix32(hbase + (unsigned)(4 * g + j) * 0x9E3779B1u), thr_m1);
            up[2 * g + j] = w;
          }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            auto sw = __builtin_amdgcn_permlane32_swap(up[4 * h2 + d], up[4 * h2 + 2 + d], false, false);
            up[4 * h2 + d] = sw[0]; up[4 * h2 + 2 + d] = sw[1];
          }
        {                                               // rows >= Tout: out of range, dropped -- but ISSUED
          const unsigned vo = (unsigned)t * pitch_u + (unsigned)(wave * 64 + 32 * c + 8 * hh) * 2u;
          __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)up[0], (int)up[1], (int)up[2], (int)up[3]}, ru, (int)vo, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)up[4], (int)up[5], (int)up[6], (int)up[7]}, ru, (int)(vo + 32u), 0, 0);
        }
      }
    }
    // the next tile's DMA is older than this tile's 16 stores (4 row groups x 2 channel tiles x 2)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
}

static bool conv_k1act_eligible(const smt_conv_desc* d) {
  return d->dtype == SMT_BF16 && d->taps == 1 && d->c_in == 64 && d->c_out == K1_COUT && d->stride == 1 &&
         d->out_stride == 1 && d->out_offset == 0 && d->t_y == d->t_out && d->t_in == d->t_out && !d->y && d->act_out &&
         d->y_act && !d->res && !d->act_grad && !d->x2 && d->zero_page && !d->w_swizzled && d->site_width % 32 == 0;
}

static int launch_conv_k1act(ConvArgs p, const void* zero_page, hipStream_t stream) {
  p.tiles_per_batch = (p.Tout + K1_ROWS - 1) / K1_ROWS;
  const int ntiles = p.tiles_per_batch * p.B;
  int nwg = std::min(512, std::max(8, (ntiles + 1) / 2));     // two workgroups per CU (32 KiB of LDS each)
  nwg = (nwg + 7) / 8 * 8;
  const int tpw = (ntiles + nwg - 1) / nwg;
  conv_k1act_kernel<<<nwg, K1_NT, 2 * K1_X, stream>>>(p, (const __bf16*)zero_page, tpw);
  SMT_CHECK_LAUNCH("conv_k1act");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// The gate conv of GatedHiFiBlock, forward (reference models/vqvae/resnet.py:238-241): out = W g + b + x, a 64 -> 64 1 x 1
// conv with the block input as residual -- 128 B + 128 B in, 128 B out per row.  It ran on the generic register-staged
// kernel (1.07 ms/step over the 14 blocks; now 0.8 ms = 4.7 TB/s at the top level).  Same machine as conv_k1act: persistent
// workgroups (four waves, 64-row tiles, up to four per CU), the 64 x 64 weight block in registers (wave w: output channels
// 32 (w & 1).., rows 32 (w >> 1)..),
// g and x tiles through an untracked LDS-DMA double buffer, transposed MFMA tile, the generic epilogue's arithmetic element
// by element -- out = bf16(bf16(acc + b) * keep_row + x) -- v_permlane32_swap pairing, 16-byte stores through a V#, one
// counted wait per tile.
constexpr int C64_ROWS = 64, C64_NT = 256, C64_TILE = C64_ROWS * 128;   // four waves: 32 output channels x 32 rows each

__global__ __launch_bounds__(C64_NT) void conv1x1_c64_kernel(ConvArgs p, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // 2 x [g tile | x tile], 64 rows x 128 B each
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int ct = wave & 1, rg = wave >> 1;

  const int ntiles = p.tiles_per_batch * p.B;
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int tile_begin = wg * tiles_per_wg;
  const int tile_end = min(ntiles, tile_begin + tiles_per_wg);
  if (tile_begin >= tile_end) return;

  bf16x8 wfrag[4];
  float bval[16];
  {
    const int co = 32 * ct + r;
    const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.w) + (size_t)co * 128;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wfrag[kk] = *reinterpret_cast<const bf16x8*>(wrow + ((2 * kk + hh) << 4));
    // accumulator element 4g + k of a lane = output channel 32 ct + 4 hh + 8g + k
#pragma unroll
    for (int e = 0; e < 16; ++e) bval[e] = p.bias ? p.bias[32 * ct + 4 * hh + 8 * (e >> 2) + (e & 3)] : 0.f;
  }

  auto decode = [&](int tile, int& b, int& t0) {
    b = __builtin_amdgcn_readfirstlane(tile / p.tiles_per_batch);
    t0 = __builtin_amdgcn_readfirstlane((tile - b * p.tiles_per_batch) * C64_ROWS);
  };
  const unsigned pitch_x = (unsigned)p.ldx * 2u, pitch_r = (unsigned)p.ldr * 2u, pitch_y = (unsigned)p.ldy * 2u;
  const int srow = 8 * wave + (lane >> 3);                         // group wave + 8 q: rows 8 wave + .. + 64 q
  const unsigned schunk = (unsigned)(((lane & 7) ^ ((srow >> 1) & 7)) << 4);   // chunk c of row n sits at c ^ ((n >> 1) & 7)
  auto stage = [&](int tile, int buf) {
    int b, t0;
    decode(tile, b, t0);
    const int len_in = p.lens_in ? min(scalar_load_i32(p.lens_in + b), p.Tin) : p.Tin;
    const UntrackedRsrc rx = untracked_rsrc(p.x, (long long)b * p.x_bs * 2, (unsigned)len_in * pitch_x);
    const UntrackedRsrc rr = untracked_rsrc(p.res, (long long)b * p.res_bs * 2, (unsigned)p.Tout * pitch_r);
    unsigned char* base = smem + (size_t)buf * 2 * C64_TILE + wave * 1024;
#pragma unroll
    for (int q = 0; q < (C64_ROWS / 8) / (C64_NT / 64); ++q) {     // 8 rows x 8 chunks per instruction
      untracked_dma16(rx, (unsigned)(t0 + srow + 8 * (C64_NT / 64) * q) * pitch_x + schunk, base + q * (C64_NT / 64) * 1024);
      untracked_dma16(rr, (unsigned)(t0 + srow + 8 * (C64_NT / 64) * q) * pitch_r + schunk, base + C64_TILE + q * (C64_NT / 64) * 1024);
    }
  };

  stage(tile_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first tile, weights and biases; later tiles: counted wait at the END
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(wfrag[kk]));
#pragma unroll
  for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(bval[e]));
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int buf = (tile - tile_begin) & 1;
    int b, t0;
    decode(tile, b, t0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave's part of this tile landed; the other buffer is free again
    if (tile + 1 < tile_end) stage(tile + 1, buf ^ 1);
    const unsigned char* gt = smem + (size_t)buf * 2 * C64_TILE;
    const unsigned char* xt = gt + C64_TILE;
    const __amdgpu_buffer_rsrc_t ry = ws_rsrc(p.y, (long long)b * p.y_bs * 2, (unsigned)p.Tout * pitch_y);
    const int len_out = p.lens_out ? scalar_load_i32(p.lens_out + b) : 0x7fffffff;
    const int row = 32 * rg + r;
    const int swz = (row >> 1) & 7;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8 bv = *reinterpret_cast<const bf16x8*>(gt + row * 128 + (((2 * kk + hh) ^ swz) << 4));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[kk], bv, acc, 0, 0, 0);
    }
    const int t = t0 + row;
    const float keep_row = (t >= len_out) ? 0.f : 1.f;
    unsigned yp[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // residual channels 32 ct + 8 g + 4 hh + 0..3 of this row: half of the 16-byte chunk 4 ct + g
      const uint2 rv = *reinterpret_cast<const uint2*>(xt + row * 128 + (((4 * ct + g) ^ swz) << 4) + 8 * hh);
      const unsigned h01 = pack_bf16x2(acc[4 * g] + bval[4 * g], acc[4 * g + 1] + bval[4 * g + 1]);
      const unsigned h23 = pack_bf16x2(acc[4 * g + 2] + bval[4 * g + 2], acc[4 * g + 3] + bval[4 * g + 3]);
      yp[2 * g] = pack_bf16x2(fmaf(__builtin_bit_cast(float, h01 << 16), keep_row, __builtin_bit_cast(float, rv.x << 16)),
                              fmaf(__builtin_bit_cast(float, h01 & 0xffff0000u), keep_row, __builtin_bit_cast(float, rv.x & 0xffff0000u)));
      yp[2 * g + 1] = pack_bf16x2(fmaf(__builtin_bit_cast(float, h23 << 16), keep_row, __builtin_bit_cast(float, rv.y << 16)),
                                  fmaf(__builtin_bit_cast(float, h23 & 0xffff0000u), keep_row, __builtin_bit_cast(float, rv.y & 0xffff0000u)));
    }
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        auto sw = __builtin_amdgcn_permlane32_swap(yp[4 * h2 + d], yp[4 * h2 + 2 + d], false, false);
        yp[4 * h2 + d] = sw[0]; yp[4 * h2 + 2 + d] = sw[1];
      }
    {                                                   // rows >= Tout: out of range, dropped -- but ISSUED
      const unsigned vo = (unsigned)t * pitch_y + (unsigned)(32 * ct + 8 * hh) * 2u;
      __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[0], (int)yp[1], (int)yp[2], (int)yp[3]}, ry, (int)vo, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(i32x4v{(int)yp[4], (int)yp[5], (int)yp[6], (int)yp[7]}, ry, (int)(vo + 32u), 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");    // the next tile's DMA is older than this tile's two stores
  }
}

static bool conv1x1_c64_eligible(const smt_conv_desc* d) {
  static const bool off = getenv("SMT_NO_C64") != nullptr;
  return !off && d->dtype == SMT_BF16 && d->taps == 1 && d->c_in == 64 && d->c_out == 64 && d->stride == 1 && d->out_stride == 1 &&
         d->out_offset == 0 && d->t_y == d->t_out && d->t_in == d->t_out && d->y && d->res && !d->act_out && !d->act_grad &&
         !d->x2 && !d->w_swizzled && d->zero_page && d->ld_x % 8 == 0 && d->ld_res % 8 == 0 && d->ld_y % 8 == 0;
}

static int launch_conv1x1_c64(ConvArgs p, hipStream_t stream) {
  p.tiles_per_batch = (p.Tout + C64_ROWS - 1) / C64_ROWS;
  const int ntiles = p.tiles_per_batch * p.B;
  int nwg = std::min(1024, std::max(8, (ntiles + 1) / 2));    // four workgroups per CU (32 KiB of LDS, four waves each)
  nwg = (nwg + 7) / 8 * 8;
  const int tpw = (ntiles + nwg - 1) / nwg;
  (void)hipFuncSetAttribute((const void*)conv1x1_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * C64_TILE);
  conv1x1_c64_kernel<<<nwg, C64_NT, 4 * C64_TILE, stream>>>(p, tpw);
  SMT_CHECK_LAUNCH("conv1x1_c64");
  return 0;
}

static int launch_conv1x1_dma(ConvArgs p, const void* zero_page, hipStream_t stream) {
  p.tiles_per_batch = (p.Tout + DMA_BM - 1) / DMA_BM;
  const int ntiles = p.tiles_per_batch * p.B;
  // two workgroups per CU (70 KB of LDS each); at least 2 tiles per workgroup so that the pipeline pays
  int nwg = std::min(512, std::max(8, (ntiles + 1) / 2));
  nwg = (nwg + 7) / 8 * 8;
  const int tpw = (ntiles + nwg - 1) / nwg;
  dim3 grid((unsigned)nwg, (unsigned)(p.Cout / DMA_BN));
  const size_t lds = 2 * P1_BUF;
  (void)hipFuncSetAttribute((const void*)conv1x1_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  conv1x1_dma_kernel<<<grid, DMA_NT, lds, stream>>>(p, (const __bf16*)zero_page, tpw);
  SMT_CHECK_LAUNCH("conv1x1_dma");
  return 0;
}

static bool conv_dma_eligible(const smt_conv_desc* d) {
  return d->dtype == SMT_BF16 && d->w_swizzled && d->c_in % 128 == 0 && d->c_out % 128 == 0 && d->stride == 1 &&
         d->out_stride == 1 && d->out_offset == 0 && d->t_y == d->t_out && d->zero_page != nullptr;
}

// Plan of the LDS-DMA family for one conv: dilation classes, then weight-stationary / 256-row / 128-row tiles.
// rows a dilation class must have for the decomposition to pay (tile quantisation wastes ceil(Tc/128)*128 - Tc rows)
static int class_min_rows() {
  static const int v = getenv("SMT_CLASS_MIN_ROWS") ? atoi(getenv("SMT_CLASS_MIN_ROWS")) : 128;
  return v;
}
// tiles a launch must have for the persistent weight-stationary kernel (it loads the weight block once per workgroup)
static int ws_min_tiles() {
  static const int v = getenv("SMT_WS_MIN_TILES") ? atoi(getenv("SMT_WS_MIN_TILES")) : 512;
  return v;
}
struct DmaPlan { bool ws; bool big; int buf_bytes; size_t lds; int tiles_per_wg; dim3 grid; };
static bool plan_conv_dma(ConvArgs& p, DmaPlan& pl) {
  // Dilation classes (see conv_gemm_dma_kernel) for same-size convs whose padding is a multiple of a large
  // dilation, as long as a class still has enough rows to fill tiles.
  p.rs = 1;
  if (p.dil >= 8 && p.taps > 1 && p.pad % p.dil == 0 && p.Tin == p.Tout && p.Tout / p.dil >= class_min_rows()) {
    p.rs = p.dil; p.pad /= p.dil; p.dil = 1;
  }
  const int tc_max = (p.Tout + p.rs - 1) / p.rs;
  {  // weight-stationary persistent kernel: 128 input channels, odd tap counts 3..9, enough tiles per workgroup
    static const bool no_ws = getenv("SMT_CONV_NO_WS") != nullptr;
    const int rows_pad = (WS_BM + (p.taps - 1) * p.dil + 3) & ~3;
    const int buf_bytes = (int)align_up((size_t)std::max(rows_pad * 256, WS_BM * (DMA_BN + 8) * 2), 1024);
    const size_t lds = 2 * (size_t)buf_bytes + 2 * WS_EPI;
    const int tpb = (tc_max + WS_BM - 1) / WS_BM;
    const long long ntiles = (long long)tpb * p.B * p.rs;
    if (!no_ws && p.Cin == 128 && p.taps >= 3 && p.taps <= 9 && (p.taps & 1) && lds <= 160 * 1024 && ntiles >= ws_min_tiles()) {
      p.tiles_per_batch = tpb;
      const int nwg = 256;
      pl.ws = true; pl.big = false; pl.buf_bytes = buf_bytes; pl.lds = lds;
      pl.tiles_per_wg = (int)((ntiles + nwg - 1) / nwg);
      pl.grid = dim3((unsigned)nwg, (unsigned)(p.Cout / DMA_BN));
      return true;
    }
  }
  auto lds_for = [&](int bm) {
    const int rows_in = (bm - 1) + (p.taps - 1) * p.dil + 1;
    const int rows_pad = (rows_in + 3) & ~3;
    const size_t a_bytes = align_up((size_t)std::max(rows_pad * 256, bm * (DMA_BN + 8) * 2), 1024);
    return a_bytes + (size_t)(p.taps > 1 ? 2 : 1) * DMA_BN * 256;   // one weight buffer suffices for 1x1
  };
  // 256-row tiles when they fit in LDS and there are enough rows to keep every CU busy
  pl.ws = false;
  pl.big = p.taps > 1 && lds_for(256) <= 160 * 1024 && (long long)tc_max * p.B * p.rs >= 256LL * 256 * 2;
  const int bm = pl.big ? 256 : 128;
  pl.lds = lds_for(bm);
  if (pl.lds > 160 * 1024) return false;
  p.tiles_per_batch = (tc_max + bm - 1) / bm;
  const int ntiles = p.tiles_per_batch * p.B * p.rs;
  pl.grid = dim3((unsigned)(8 * ((ntiles + 7) / 8)), (unsigned)(p.Cout / DMA_BN));
  return true;
}

static int launch_conv_dma(ConvArgs p, const void* zero_page, hipStream_t stream) {
  DmaPlan pl;
  SMT_CHECK_ARG(plan_conv_dma(p, pl), "conv_gemm_dma: tile needs %zu B of LDS", pl.lds);
  if (pl.ws) {
    switch (p.taps) {
      case 3: launch_ws<3>(p, zero_page, pl.grid, pl.lds, pl.tiles_per_wg, pl.buf_bytes, stream); break;
      case 5: launch_ws<5>(p, zero_page, pl.grid, pl.lds, pl.tiles_per_wg, pl.buf_bytes, stream); break;
      case 7: launch_ws<7>(p, zero_page, pl.grid, pl.lds, pl.tiles_per_wg, pl.buf_bytes, stream); break;
      default: launch_ws<9>(p, zero_page, pl.grid, pl.lds, pl.tiles_per_wg, pl.buf_bytes, stream); break;
    }
    SMT_CHECK_LAUNCH("conv_ws");
    return 0;
  }
  if (pl.big) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_dma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    conv_gemm_dma_kernel<4><<<pl.grid, DMA_NT, pl.lds, stream>>>(p, (const __bf16*)zero_page);
  } else {
    (void)hipFuncSetAttribute((const void*)conv_gemm_dma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    conv_gemm_dma_kernel<2><<<pl.grid, DMA_NT, pl.lds, stream>>>(p, (const __bf16*)zero_page);
  }
  SMT_CHECK_LAUNCH("conv_gemm_dma");
  return 0;
}

template <typename T>
static size_t conv_gemm_lds_bytes(const ConvArgs& p, int BN) {
  constexpr int EPV = Tr<T>::EPV, CCH = Tr<T>::CCH, KC = Tr<T>::KC, BM = Tr<T>::BM;
  int rows_in = (BM - 1) * p.stride + (p.taps - 1) * p.dil + 1;
  int pitch_a = std::min(p.Cin, CCH) + EPV;
  size_t a = align_up((size_t)std::max(rows_in * pitch_a, BM * (BN + EPV)) * sizeof(T), 16);
  const int nsteps = p.taps * ((std::min(p.Cin, CCH) + KC - 1) / KC);
  return a + (size_t)(nsteps > 1 ? 2 : 1) * BN * (KC + EPV) * sizeof(T);
}

template <typename T>
static int launch_conv_gemm(ConvArgs p, hipStream_t stream) {
  constexpr int BM = Tr<T>::BM;
  const int BN = (p.Cout > 64) ? 128 : 64;
  p.tiles_per_batch = (p.Tout + BM - 1) / BM;
  const int ntiles = p.tiles_per_batch * p.B;
  dim3 grid((unsigned)(8 * ((ntiles + 7) / 8)), (unsigned)((p.Cout + BN - 1) / BN));
  size_t lds = conv_gemm_lds_bytes<T>(p, BN);
  SMT_CHECK_ARG(lds <= 160 * 1024, "conv_gemm: tile needs %zu B of LDS (taps=%d dil=%d stride=%d Cin=%d)", lds,
                p.taps, p.dil, p.stride, p.Cin);
  // two waves per SIMD (512 threads) for the 128-wide bf16 tiles; everything else keeps 256
  if (BN == 128) {
    constexpr int NT = sizeof(T) == 2 ? 512 : 256;
    (void)hipFuncSetAttribute((const void*)conv_gemm_kernel<T, 128, NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    conv_gemm_kernel<T, 128, NT><<<grid, NT, lds, stream>>>(p);
  } else {
    (void)hipFuncSetAttribute((const void*)conv_gemm_kernel<T, 64, 256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    conv_gemm_kernel<T, 64, 256><<<grid, 256, lds, stream>>>(p);
  }
  SMT_CHECK_LAUNCH("conv_gemm");
  return 0;
}

// ---- weight repacking: dst[tap][o][i] (act dtype) = src[o*so + i*si + jmap[tap]*sj] (fp32) ------------
struct PackArgs {
  const float* src; void* dst; int O, I, taps; long long so, si, sj; int jmap[16]; int swizzle;
};
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(PackArgs p) {
  const long long total = (long long)p.taps * p.O * p.I;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    int i = (int)(e % p.I);
    int o = (int)((e / p.I) % p.O);
    int tap = (int)(e / ((long long)p.I * p.O));
    long long dst = e;
    if (p.swizzle) {  // LDS-DMA layout: within every 128-channel group the 8-channel chunk index is XORed with (o & 15)
      const int chunk = (i >> 3) & 15;
      dst = e - ((long long)chunk << 3) + ((long long)(chunk ^ (o & 15)) << 3);
    }
    reinterpret_cast<T*>(p.dst)[dst] = (T)p.src[o * p.so + i * p.si + p.jmap[tap] * p.sj];
  }
}

// Table-driven repack of MANY weights in one launch (the table lives in device memory and is built once by the
// host; parameter storage is stable across optimiser steps).  block_entry[b] = table row of block b,
// block_local[b] = its index among that row's blocks; 1024 elements per block.
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const smt_pack_entry* __restrict__ table,
                                                                   const int* __restrict__ block_entry,
                                                                   const int* __restrict__ block_local) {
  const smt_pack_entry e = table[block_entry[blockIdx.x]];
  const long long total = (long long)e.taps * e.n_out * e.n_in;
  const long long base = (long long)block_local[blockIdx.x] * 1024;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long idx = base + u * 256 + threadIdx.x;
    if (idx >= total) break;
    const int i = (int)(idx % e.n_in);
    const int o = (int)((idx / e.n_in) % e.n_out);
    const int tap = (int)(idx / ((long long)e.n_in * e.n_out));
    int ipos = i;
    if (e.swizzle) ipos = (i & ~127) | ((((i >> 3) & 15) ^ (o & 15)) << 3) | (i & 7);
    const float v = e.src[o * e.stride_out + i * e.stride_in + e.tap_map[tap] * e.stride_tap];
    const long long d = e.dst_offset + (long long)tap * e.dst_tap_stride + (long long)o * e.dst_row_stride + ipos;
    if (e.dtype == SMT_BF16) reinterpret_cast<__bf16*>(e.dst)[d] = (__bf16)v;
    else reinterpret_cast<float*>(e.dst)[d] = v;
  }
}

}  // namespace smt

using namespace smt;

extern "C" int smt_pack_weights_batched(const smt_pack_entry* table_dev, const int* block_entry_dev,
                                        const int* block_local_dev, int n_blocks, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_blocks <= 0) return 0;
  SMT_CHECK_ARG(table_dev && block_entry_dev && block_local_dev, "smt_pack_weights_batched: null pointer");
  pack_weights_batched_kernel<<<(unsigned)n_blocks, 256, 0, stream>>>(table_dev, block_entry_dev, block_local_dev);
  SMT_CHECK_LAUNCH("pack_weights_batched");
  return 0;
}

extern "C" int smt_pack_weight(const float* src, void* dst, int dtype, int n_out, int n_in, int taps,
                               int64_t stride_out, int64_t stride_in, int64_t stride_tap, const int* tap_map,
                               int swizzle, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(src && dst && tap_map, "smt_pack_weight: null pointer");
  SMT_CHECK_ARG(taps >= 1 && taps <= 16, "smt_pack_weight: taps must be in [1, 16]");
  PackArgs p;
  p.src = src; p.dst = dst; p.O = n_out; p.I = n_in; p.taps = taps;
  p.so = stride_out; p.si = stride_in; p.sj = stride_tap;
  for (int t = 0; t < taps; ++t) p.jmap[t] = tap_map[t];
  SMT_CHECK_ARG(!swizzle || (n_in % 128 == 0 && dtype == SMT_BF16), "smt_pack_weight: swizzle needs bf16 and n_in %% 128 == 0");
  p.swizzle = swizzle;
  long long total = (long long)taps * n_out * n_in;
  unsigned grid = (unsigned)std::min<long long>(1024, (total + 255) / 256);
  if (dtype == SMT_BF16) pack_weight_kernel<__bf16><<<grid, 256, 0, stream>>>(p);
  else if (dtype == SMT_F32) pack_weight_kernel<float><<<grid, 256, 0, stream>>>(p);
  else SMT_CHECK_ARG(false, "smt_pack_weight: bad dtype %d", dtype);
  SMT_CHECK_LAUNCH("pack_weight");
  return 0;
}

static void conv_args_from_desc(const smt_conv_desc* d, ConvArgs& p) {
  p.x = d->x; p.w = d->w; p.bias = d->bias; p.y = d->y; p.res = d->res; p.gate_h = d->act_grad_src;
  p.lens_in = d->lens_in; p.lens_out = d->lens_out;
  p.x_bs = d->bs_x; p.y_bs = d->bs_y; p.res_bs = d->bs_res; p.gh_bs = d->bs_act;
  p.ldx = d->ld_x; p.ldy = d->ld_y; p.ldr = d->ld_res; p.ldgh = d->ld_act;
  p.B = d->batch; p.Tin = d->t_in; p.Tout = d->t_out; p.Cin = d->c_in; p.Cout = d->c_out;
  p.taps = d->taps; p.stride = d->stride; p.dil = d->dilation; p.pad = d->padding;
  p.out_stride = d->out_stride; p.out_offset = d->out_offset; p.Ty = d->t_y;
  p.y_act = d->y_act; p.ya_bs = d->bs_yact; p.ldya = d->ld_yact;
  p.act_out = d->act_out; p.epi_act = d->act_grad;
  for (int i = 0; i < 8; ++i) p.drop_keys[i] = d->drop_keys[i];
  p.drop_keys_dev = d->drop_keys_dev; p.drop_keys_dev_stride = d->drop_keys_dev_stride;
  p.site_width = d->site_width; p.drop_thresh16 = d->drop_thresh16; p.drop_scale = d->drop_scale;
  p.tiles_per_batch = 0;
  p.rs = 1;
  p.x2 = d->x2; p.w2 = d->w2; p.bias2 = d->bias2; p.lens_in2 = d->lens_in2; p.x2_bs = d->bs_x2; p.ldx2 = d->ld_x2;
  { static int dbg = getenv("SMT_CONV_DBG") ? atoi(getenv("SMT_CONV_DBG")) : 0; p.dbg = dbg; }
}

enum ConvKernelKind { K_GENERIC, K_1X1, K_DMA, K_WS, K_FOLD, K_K1ACT, K_C64 };
static bool conv_fold_eligible(const smt_conv_desc* d) {
  return d->x2 && d->w2 && d->c_in2 == 64 && d->taps == 1 && d->c_in == 128 && !d->res && !d->act_grad && !d->act_out &&
         d->y && d->ld_x2 % 8 == 0;
}
// the one dispatch rule (smt_conv1d_ntc and smt_conv1d_kernel_name both use it)
static ConvKernelKind pick_kernel(const smt_conv_desc* d, const ConvArgs& p0) {
  if (conv_k1act_eligible(d)) return K_K1ACT;
  if (conv1x1_c64_eligible(d)) return K_C64;
  if (!conv_dma_eligible(d)) return K_GENERIC;
  if (d->x2) return conv_fold_eligible(d) ? K_FOLD : K_GENERIC;
  if (d->taps == 1 && d->c_in == 128) return K_1X1;
  ConvArgs p = p0;
  DmaPlan pl;
  if (!plan_conv_dma(p, pl)) return K_GENERIC;
  return pl.ws ? K_WS : K_DMA;
}

#if SMT_WS_STAMP
extern "C" int smt_ws_debug_dump(unsigned long long* host, int n, int reset) {
  int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ws_dbg), sizeof(unsigned long long) * n);
  if (reset) { static unsigned long long z[256 * 8 * 8]; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(ws_dbg), z, sizeof(z)); }
  return rc;
}
#endif

extern "C" int smt_conv1d_ntc(const smt_conv_desc* d, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(d && d->x && d->w && (d->y || d->y_act), "smt_conv1d_ntc: null pointer");
  const int epv = d->dtype == SMT_BF16 ? 8 : 4;
  SMT_CHECK_ARG(d->dtype == SMT_BF16 || d->dtype == SMT_F32, "smt_conv1d_ntc: bad dtype");
  SMT_CHECK_ARG(d->c_in % (2 * epv) == 0, "smt_conv1d_ntc: c_in=%d must be a multiple of %d", d->c_in, 2 * epv);
  SMT_CHECK_ARG(d->c_out % epv == 0, "smt_conv1d_ntc: c_out=%d must be a multiple of %d", d->c_out, epv);
  {
    const int cch = d->dtype == SMT_BF16 ? 128 : 64;
    const int first = d->c_in < cch ? d->c_in : cch;
    SMT_CHECK_ARG((first & (first - 1)) == 0 && (d->c_in <= cch || d->c_in % cch == 0),
                  "smt_conv1d_ntc: c_in=%d must be a power of two up to %d or a multiple of it", d->c_in, cch);
    SMT_CHECK_ARG(!d->act_out || d->site_width % 2 == 0, "smt_conv1d_ntc: site_width must be even");
  }
  SMT_CHECK_ARG(d->ld_x % epv == 0 && d->ld_y % epv == 0 && (!d->res || d->ld_res % epv == 0) &&
                    (!d->act_grad_src || d->ld_act % epv == 0),
                "smt_conv1d_ntc: row pitches must keep 16-byte alignment");
  SMT_CHECK_ARG(d->taps >= 1 && d->stride >= 1 && d->dilation >= 1 && d->out_stride >= 1, "smt_conv1d_ntc: bad geometry");
  SMT_CHECK_ARG(!d->act_grad || d->act_grad_src, "smt_conv1d_ntc: act_grad needs act_grad_src");
  SMT_CHECK_ARG(!d->act_out || (d->y_act && d->site_width > 0 && d->site_width % epv == 0 && d->ld_yact % epv == 0 &&
                                (d->c_out + d->site_width - 1) / d->site_width <= 8),
                "smt_conv1d_ntc: act_out needs y_act and 1..8 sites of site_width channels");
  if (d->batch == 0 || d->t_out == 0) return 0;
  ConvArgs p;
  conv_args_from_desc(d, p);
  const ConvKernelKind kind = pick_kernel(d, p);
  SMT_CHECK_ARG(!d->x2 || kind == K_FOLD,
                "smt_conv1d_ntc: the folded second term needs the bf16 1x1 LDS-DMA path (c_in 128, c_in2 64, no other epilogue)");
  switch (kind) {
    case K_FOLD: return launch_conv1x1_fold(p, d->zero_page, stream);
    case K_K1ACT: return launch_conv_k1act(p, d->zero_page, stream);
    case K_C64: return launch_conv1x1_c64(p, stream);
    case K_1X1: return launch_conv1x1_dma(p, d->zero_page, stream);
    case K_DMA: case K_WS: return launch_conv_dma(p, d->zero_page, stream);
    default: break;
  }
  SMT_CHECK_ARG(!d->w_swizzled, "smt_conv1d_ntc: swizzled weights need the LDS-DMA path (bf16, stride 1, channels %% 128)");
  if (d->dtype == SMT_BF16) return launch_conv_gemm<__bf16>(p, stream);
  return launch_conv_gemm<float>(p, stream);
}

extern "C" const char* smt_conv1d_kernel_name(const smt_conv_desc* d) {
  if (!d) return "";
  ConvArgs p;
  conv_args_from_desc(d, p);
  switch (pick_kernel(d, p)) {
    case K_1X1: return "conv1x1_dma";
    case K_FOLD: return "conv1x1_fold";
    case K_K1ACT: return "conv_k1act";
    case K_C64: return "conv1x1_c64";
    case K_WS: {  // same rule as launch_ws: two waves per SIMD for <= 5 taps; else the pipelined variant for the activated-output epilogue
      const bool fwd = !d->y && d->act_out && !d->res && !d->act_grad, dgrad = d->y && !d->act_out && d->res && d->act_grad;
      if (d->taps <= WS2_MAX_TAPS && (fwd || dgrad) && !getenv("SMT_CONV_NO_WS2")) return "conv_ws2";
      return (fwd && !getenv("SMT_CONV_NO_PIPE")) ? "conv_ws_pipe" : "conv_ws";
    }
    case K_DMA: return "conv_gemm_dma";
    default: return "conv_gemm";
  }
}
