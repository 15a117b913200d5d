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

#include "conv_common.h"

namespace smt {

constexpr int WG_MAX_TAPS = 9;

struct WgradArgs {
  const void* x; const void* dy; float* slab;
  const int* lens_in;
  long long x_bs, dy_bs;
  int ldx, ldy;
  int B, Tin, Tout, Ty, Cin, Cout;
  int taps, stride, dil, pad, out_stride, out_offset;
  int rows_per_chunk, chunks_per_batch, nblk_ci, with_bias;
};

template <typename T> struct WTr;
template <> struct WTr<__bf16> { static constexpr int R = 128; };
template <> struct WTr<float> { static constexpr int R = 64; };

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

template <typename T, int NT>  // NT = number of accumulator planes (taps + optional bias plane)
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
  constexpr int EPV = Tr<T>::EPV;
  constexpr int R = WTr<T>::R;          // rows per staged tile
  constexpr int CB = 64;                // channels per block side
  constexpr int PITCH = CB + EPV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, hh = lane >> 5;

  const int blk = blockIdx.y;
  const int co0 = (blk / p.nblk_ci) * CB, ci0 = (blk % p.nblk_ci) * CB;
  const int b = blockIdx.x / p.chunks_per_batch;
  const int chunk = blockIdx.x % p.chunks_per_batch;
  const int t_begin = chunk * p.rows_per_chunk;
  const int t_end = min(p.Tout, t_begin + p.rows_per_chunk);

  const int rows_x = (R - 1) * p.stride + (p.taps - 1) * p.dil + 1;
  T* lds_dy = reinterpret_cast<T*>(smem);                 // [R][PITCH]
  T* lds_x = lds_dy + R * PITCH;                          // [rows_x][PITCH]
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
    constexpr int VPR = CB / EPV;
    constexpr int UB = 8;
    const int tin0 = t0 * p.stride - p.pad;
    const int n_dy = R * VPR, n_x = rows_x * VPR;
    for (int f0 = tid; f0 < n_dy + n_x; f0 += 256 * UB) {
      Vec<T, EPV> v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int f = f0 + 256 * u;
#pragma unroll
        for (int e = 0; e < EPV; ++e) v[u].v[e] = (T)0.f;
        if (f < n_dy) {
          const int row = f / VPR, cv = f % VPR;
          const int t = t0 + row;
          const int ty = t * p.out_stride + p.out_offset;
          if (t < t_end && ty < p.Ty && co0 + cv * EPV < p.Cout)
            v[u] = *reinterpret_cast<const Vec<T, EPV>*>(dyg + (long long)ty * p.ldy + co0 + cv * EPV);
        } else if (f < n_dy + n_x) {
          const int g = f - n_dy;
          const int row = g / VPR, cv = g % VPR;
          const int tin = tin0 + row;
          if (tin >= 0 && tin < len_in && ci0 + cv * EPV < p.Cin)
            v[u] = *reinterpret_cast<const Vec<T, EPV>*>(xg + (long long)tin * p.ldx + ci0 + cv * EPV);
        }
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int f = f0 + 256 * u;
        if (f < n_dy) {
          *reinterpret_cast<Vec<T, EPV>*>(lds_dy + (f / VPR) * PITCH + (f % VPR) * EPV) = v[u];
        } else if (f < n_dy + n_x) {
          const int g = f - n_dy;
          *reinterpret_cast<Vec<T, EPV>*>(lds_x + (g / VPR) * PITCH + (g % VPR) * EPV) = v[u];
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
      for (int k0 = 0; k0 < R; k0 += 16) {
        bf16x8 a = frag_tr_bf16(dyt, PITCH, k0, wm * 32, lane);   // A[co][k] = dy[k][co]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (j < ntaps) {
            // B[k][ci] = x[(k)*stride + j*dil][ci]   (stride 1 for every transposed read;
            // strided convs read rows k*stride: handled by row index arithmetic below)
            bf16x8 bfrag;
            if (p.stride == 1) {
              bfrag = frag_tr_bf16(xt, PITCH, k0 + j * p.dil, wn * 32, lane);
            } else {
              // strided rows are not a dense 4-row block: gather element-wise
              typedef short s16x8 __attribute__((ext_vector_type(8)));
              s16x8 tmp;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const short* sp = reinterpret_cast<const short*>(xt) +
                                  ((k0 + 8 * hh + e) * p.stride + j * p.dil) * PITCH + wn * 32 + r;
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
      for (int k0 = 0; k0 < R; k0 += 2) {
        const float a = dyt[(k0 + hh) * PITCH + wm * 32 + r];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (j < ntaps) {
            const float bv = xt[((k0 + hh) * p.stride + j * p.dil) * PITCH + wn * 32 + r];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[j], 0, 0, 0);
          } else if (j == ntaps && bias_plane) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, 1.0f, acc[j], 0, 0, 0);
          }
        }
      }
    }
  }
  // partial block -> slab[chunk_global][plane][co 64][ci 64] for this (co,ci) block
  const int planes = ntaps + (p.with_bias ? 1 : 0);
  float* out = p.slab + ((size_t)blockIdx.x * gridDim.y + blk) * (size_t)planes * CB * CB;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (j >= planes) break;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;  // co
      const int col = wn * 32 + r;                                 // ci
      out[((size_t)j * CB + row) * CB + col] = acc[j][e];
    }
  }
}

struct WreduceArgs {
  const float* slab; float* dw; float* db;
  int n_chunks, nblk, nblk_ci, planes, taps, Cin, Cout, with_bias;
  long long so, si, sj; int jmap[16];
};

// one thread per (plane, co, ci); chunks summed in index order
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(WreduceArgs p) {
  const long long total = (long long)p.planes * p.Cout * p.Cin;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int ci = (int)(e % p.Cin), co = (int)((e / p.Cin) % p.Cout), plane = (int)(e / ((long long)p.Cin * p.Cout));
  const int blk = (co / 64) * p.nblk_ci + (ci / 64);
  const size_t blk_elems = (size_t)p.planes * 64 * 64;
  const float* src = p.slab + (size_t)blk * blk_elems + ((size_t)plane * 64 + (co % 64)) * 64 + (ci % 64);
  if (plane == p.taps) {  // bias plane: every ci column holds db[co]; take column 0 of ci-block 0
    if (ci != 0) return;
  }
  float s = 0.f;
  for (int c = 0; c < p.n_chunks; ++c) s += src[(size_t)c * p.nblk * blk_elems];
  if (plane < p.taps) p.dw[co * p.so + ci * p.si + p.jmap[plane] * p.sj] = s;
  else if (p.db) p.db[co] = s;
}

static void wgrad_plan(const smt_conv_desc* d, int* rows_per_chunk, int* chunks_per_batch, int* nblk_co,
                       int* nblk_ci, int* planes) {
  *nblk_co = (d->c_out + 63) / 64;
  *nblk_ci = (d->c_in + 63) / 64;
  const int R = d->dtype == SMT_BF16 ? 128 : 64;
  long long total_rows = (long long)d->batch * d->t_out;
  long long target_wgs = 1024;
  long long rows = (total_rows * (*nblk_co) * (*nblk_ci) + target_wgs - 1) / target_wgs;
  rows = std::max<long long>(R, (rows + R - 1) / R * R);
  rows = std::min<long long>(rows, ((long long)d->t_out + R - 1) / R * R);
  *rows_per_chunk = (int)rows;
  *chunks_per_batch = (int)((d->t_out + rows - 1) / rows);
  *planes = d->taps + 1;
}

}  // namespace smt

using namespace smt;

extern "C" size_t smt_conv1d_wgrad_workspace_bytes(const smt_conv_desc* d) {
  int rpc, cpb, nco, nci, planes;
  wgrad_plan(d, &rpc, &cpb, &nco, &nci, &planes);
  return (size_t)d->batch * cpb * nco * nci * planes * 64 * 64 * sizeof(float);
}

template <typename T>
static int launch_wgrad(const WgradArgs& a, dim3 grid, size_t lds, int planes, hipStream_t stream) {
#define SMT_WG_CASE(NT)                                                                                       \
  case NT: {                                                                                                  \
    static bool attr = false;                                                                                 \
    if (!attr) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                160 * 1024);                                                                  \
      attr = true;                                                                                            \
    }                                                                                                         \
    conv_wgrad_kernel<T, NT><<<grid, 256, lds, stream>>>(a);                                                  \
  } break;
  switch (planes) {
    SMT_WG_CASE(2) SMT_WG_CASE(3) SMT_WG_CASE(4) SMT_WG_CASE(5) SMT_WG_CASE(6) SMT_WG_CASE(8) SMT_WG_CASE(10)
    default:
      set_error("conv_wgrad: unsupported tap count %d", planes - 1);
      return 1;
  }
#undef SMT_WG_CASE
  SMT_CHECK_LAUNCH("conv_wgrad");
  return 0;
}

extern "C" int smt_conv1d_wgrad(const smt_conv_desc* d, float* dweight, int64_t stride_out, int64_t stride_in,
                                int64_t stride_tap, const int* tap_map, float* dbias, void* workspace,
                                size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(d && d->x && d->y && dweight && tap_map && workspace, "smt_conv1d_wgrad: null pointer");
  SMT_CHECK_ARG(d->dtype == SMT_BF16 || d->dtype == SMT_F32, "smt_conv1d_wgrad: bad dtype");
  const int epv = d->dtype == SMT_BF16 ? 8 : 4;
  SMT_CHECK_ARG(d->c_in % epv == 0 && d->c_out % epv == 0 && d->ld_x % epv == 0 && d->ld_y % epv == 0,
                "smt_conv1d_wgrad: channels / pitches must keep 16-byte alignment");
  SMT_CHECK_ARG(d->taps >= 1 && d->taps <= WG_MAX_TAPS, "smt_conv1d_wgrad: taps must be in [1, %d]", WG_MAX_TAPS);
  SMT_CHECK_ARG(workspace_bytes >= smt_conv1d_wgrad_workspace_bytes(d), "smt_conv1d_wgrad: workspace too small");
  int rpc, cpb, nco, nci, planes;
  wgrad_plan(d, &rpc, &cpb, &nco, &nci, &planes);
  WgradArgs a;
  a.x = d->x; a.dy = d->y; a.slab = (float*)workspace; a.lens_in = d->lens_in;
  a.x_bs = d->bs_x; a.dy_bs = d->bs_y; a.ldx = d->ld_x; a.ldy = d->ld_y;
  a.B = d->batch; a.Tin = d->t_in; a.Tout = d->t_out; a.Ty = d->t_y; a.Cin = d->c_in; a.Cout = d->c_out;
  a.taps = d->taps; a.stride = d->stride; a.dil = d->dilation; a.pad = d->padding;
  a.out_stride = d->out_stride; a.out_offset = d->out_offset;
  a.rows_per_chunk = rpc; a.chunks_per_batch = cpb; a.nblk_ci = nci; a.with_bias = 1;
  if (d->batch > 0 && d->t_out > 0) {
    dim3 grid((unsigned)(d->batch * cpb), (unsigned)(nco * nci));
    const int R = d->dtype == SMT_BF16 ? 128 : 64;
    const int rows_x = (R - 1) * d->stride + (d->taps - 1) * d->dilation + 1;
    const size_t esz = d->dtype == SMT_BF16 ? 2 : 4;
    const size_t lds = (size_t)(R + rows_x) * (64 + epv) * esz;
    SMT_CHECK_ARG(lds <= 160 * 1024, "conv_wgrad: tile needs %zu B of LDS", lds);
    const int nt = planes;  // accumulator planes = taps + bias plane
    int rc = d->dtype == SMT_BF16 ? launch_wgrad<__bf16>(a, grid, lds, nt, stream)
                                  : launch_wgrad<float>(a, grid, lds, nt, stream);
    if (rc) return rc;
  }
  WreduceArgs r;
  r.slab = (const float*)workspace; r.dw = dweight; r.db = dbias;
  r.n_chunks = d->batch * cpb; r.nblk = nco * nci; r.nblk_ci = nci; r.planes = planes; r.taps = d->taps;
  r.Cin = d->c_in; r.Cout = d->c_out; r.with_bias = 1;
  r.so = stride_out; r.si = stride_in; r.sj = stride_tap;
  for (int t = 0; t < d->taps; ++t) r.jmap[t] = tap_map[t];
  if (d->batch == 0 || d->t_out == 0) r.n_chunks = 0;
  long long total = (long long)planes * d->c_out * d->c_in;
  conv_wgrad_reduce_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(r);
  SMT_CHECK_LAUNCH("conv_wgrad_reduce");
  return 0;
}
