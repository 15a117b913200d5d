// Windowed STFT on gfx950 as an in-LDS FFT (replaces the reference's DFT-as-conv1d formulation,
// datasets/transforms.py:86-123, and the multi-resolution spectral loss built on it,
// models/vqvae/losses.py:39-55).
//
// One workgroup (256 threads) owns one frame: the frame is read from HBM with the reflect padding
// resolved on the fly (coalesced: consecutive lanes read consecutive samples; overlapping frames hit
// L2), multiplied by the window while it is written to LDS, transformed by a radix-2 Stockham
// autosort FFT (log2 N passes over two LDS buffers, twiddles from a host-computed fp64->fp32 table),
// and reduced in place.  Magnitudes are only materialised for the stand-alone STFT / log-mel entry
// point; the loss kernels keep spectra on chip:
//   * forward packs the two real signals into ONE complex FFT (z = y + i*yh) and splits the spectra
//     by Hermitian symmetry, then emits per-frame partial sums of the linear and log terms;
//   * backward recomputes the spectra, forms dL/dYh on the one-sided spectrum, applies the adjoint
//     transform (a complex FFT with conjugate twiddles), windows it and overlap-adds into dyh.
#include <algorithm>

#include "smt_common.h"

namespace smt {

// SMT_HD: the transform core also compiles for the host, where tests/fft_core_host.cpp runs it lane by lane against a
// double-precision DFT (pytest -m "not gpu").
#define SMT_HD __host__ __device__ __forceinline__
struct cplx { float x, y; };
SMT_HD cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

#ifndef SMT_FFT_STAMP
#define SMT_FFT_STAMP 0   // diagnostic build (tools/fft_phases.sh): cycle sums of the phases of the one-wave frame kernels
#endif
#if SMT_FFT_STAMP
constexpr int FFT_DBG_WAVES = 8192;
__device__ unsigned long long fft_dbg[FFT_DBG_WAVES * 8];     // one row of phase slots per wave (plain stores)
#define FFT_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define FFT_ACC(k, a, b)                                                                                     \
  do {                                                                                                       \
    const unsigned wid_ = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);                    \
    if ((threadIdx.x & 63) == 0 && wid_ < FFT_DBG_WAVES) fft_dbg[wid_ * 8 + (k)] = (b) - (a);                \
  } while (0)
#else
#define FFT_T(var) do {} while (0)
#define FFT_ACC(k, a, b) do {} while (0)
#endif

// LDS index of element i of a frame.  The one-wave transform (NT = 64) writes its first pass at stride R (16 or 8 complex =
// 128 / 64 bytes) between lanes -- a 32- / 16-way bank conflict on a dense array -- so its frames carry one pad slot per R
// elements: the stride becomes R + 1 complex (34 / 18 dwords, conflict-free over a half-wave), and the second pass's four
// 16-lane groups land in four disjoint bank quarters.  The whole-workgroup form (NT = 256) keeps the dense layout.
template <int N, int NT>
SMT_HD constexpr int fidx(int i) { return NT == 64 ? i + (i >> (N == 512 ? 3 : 4)) : i; }
template <int N>
constexpr int frame_padded() { return N + (N >> (N == 512 ? 3 : 4)); }

// Stockham autosort FFT, radix-4 passes (plus one radix-2 pass when log2 N is odd): half the passes -- and barriers --
// of a radix-2 transform; this kernel is bound by the LDS round trip + barrier per pass, not by bandwidth.
// Input in `a`, result pointer returned (a or b).  tw[k] = exp(-2 pi i k / N), k < N/2; inverse => conjugate twiddles.
template <int N>
__device__ __forceinline__ cplx fft_twiddle(const cplx* __restrict__ tw, int idx, bool inverse) {
  // exp(-2 pi i idx / N) for idx < N from the half table: e^{-i(pi + x)} = -e^{-ix}
  cplx w = tw[idx & (N / 2 - 1)];
  if (idx >= N / 2) { w.x = -w.x; w.y = -w.y; }
  if (inverse) w.y = -w.y;
  return w;
}

// NT threads cooperate on one transform: 256 = the whole workgroup (a barrier per pass), 64 = ONE WAVE per frame -- a wave's
// LDS operations execute in order, so its passes need no barrier at all, and a workgroup carries four independent frames
// (round 3: the one-frame-per-workgroup form spent most of its time in the five or six barriers of a transform).
template <int NT>
__device__ __forceinline__ void fft_sync() {
  if constexpr (NT == 256) __syncthreads();
  else { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }
}

template <int N, bool INV>
__device__ __forceinline__ void fft_wave(cplx* buf, const cplx* __restrict__ tw, int lane);

template <int N, int NT>
__device__ __forceinline__ cplx* fft_lds(cplx* a, cplx* b, const cplx* __restrict__ tw, bool inverse, int tid) {
  if constexpr (NT == 64) {       // one wave per frame: in place, high radix (below); b is not used
    if (inverse) fft_wave<N, true>(a, tw, tid); else fft_wave<N, false>(a, tw, tid);
    return a;
  }
  cplx* in = a; cplx* out = b;
  int ns = 1;
#pragma unroll 1
  for (; ns * 4 <= N; ns <<= 2) {
    const int tw_stride = N / (4 * ns);
    for (int j = tid; j < N / 4; j += NT) {
      const int k = j & (ns - 1);
      const cplx u0 = in[j];
      const cplx u1 = cmul(in[j + N / 4], fft_twiddle<N>(tw, k * tw_stride, inverse));
      const cplx u2 = cmul(in[j + N / 2], fft_twiddle<N>(tw, 2 * k * tw_stride, inverse));
      const cplx u3 = cmul(in[j + 3 * (N / 4)], fft_twiddle<N>(tw, 3 * k * tw_stride, inverse));
      const cplx v0 = {u0.x + u2.x, u0.y + u2.y}, v1 = {u0.x - u2.x, u0.y - u2.y};
      const cplx v2 = {u1.x + u3.x, u1.y + u3.y};
      const cplx d = {u1.x - u3.x, u1.y - u3.y};
      // forward: times -i, inverse: times +i
      const cplx v3 = inverse ? cplx{-d.y, d.x} : cplx{d.y, -d.x};
      const int j0 = ((j - k) << 2) + k;
      out[j0] = {v0.x + v2.x, v0.y + v2.y};
      out[j0 + ns] = {v1.x + v3.x, v1.y + v3.y};
      out[j0 + 2 * ns] = {v0.x - v2.x, v0.y - v2.y};
      out[j0 + 3 * ns] = {v1.x - v3.x, v1.y - v3.y};
    }
    fft_sync<NT>();
    cplx* t = in; in = out; out = t;
  }
  if (ns < N) {   // one radix-2 pass left (log2 N odd)
    const int tw_stride = N / (2 * ns);
    for (int j = tid; j < N / 2; j += NT) {
      const int k = j & (ns - 1);
      const cplx w = fft_twiddle<N>(tw, k * tw_stride, inverse);
      const cplx u = in[j];
      const cplx v = cmul(in[j + N / 2], w);
      const int j0 = ((j - k) << 1) + k;
      out[j0] = {u.x + v.x, u.y + v.y};
      out[j0 + ns] = {u.x - v.x, u.y - v.y};
    }
    fft_sync<NT>();
    cplx* t = in; in = out; out = t;
  }
  return in;
}
// old call sites (stft_inverse_kernel): the whole workgroup on one frame
template <int N>
__device__ __forceinline__ cplx* fft_lds(cplx* a, cplx* b, const cplx* __restrict__ tw, bool inverse) {
  return fft_lds<N, 256>(a, b, tw, inverse, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------------------------------------
// One wave per frame, in place, high radix (round 3).  The whole-workgroup transform above spends its time in the LDS round
// trip + barrier of every radix-4 pass (5-6 per frame, one butterfly per thread).  Here a wave owns a frame: a lane holds the
// R inputs of a radix-R butterfly in registers (16 loads in flight), so 512 / 1,024 / 2,048 points take 3 passes
// (8.8.8 / 16.16.4 / 16.16.8) instead of 5-6, no workgroup barrier at all (a wave's LDS operations execute in order), and one
// [N] buffer per frame: every lane reads all its inputs before it writes, so the Stockham pass runs in place.
template <int R, bool INV>
struct SmallDft;
template <bool INV>
struct SmallDft<1, INV> { static SMT_HD void run(cplx (&)[1]) {} };
// twiddles of the small DFTs: exp(-2 pi i m / 16), m = 0..7 (forward); the inverse conjugates.  Functions of a constant after
// unrolling, so they fold into literals.
SMT_HD constexpr float cos16(int m) {
  return m == 0 ? 1.f : m == 1 ? 0.92387953251128674f : m == 2 ? 0.70710678118654752f : m == 3 ? 0.38268343236508977f
       : m == 4 ? 0.f : m == 5 ? -0.38268343236508977f : m == 6 ? -0.70710678118654752f : -0.92387953251128674f;
}
SMT_HD constexpr float sin16(int m) {
  return m == 0 ? 0.f : m == 1 ? 0.38268343236508977f : m == 2 ? 0.70710678118654752f : m == 3 ? 0.92387953251128674f
       : m == 4 ? 1.f : m == 5 ? 0.92387953251128674f : m == 6 ? 0.70710678118654752f : 0.38268343236508977f;
}
template <int R, bool INV>
struct SmallDft {
  // natural-order DFT of R points in registers by radix-2 decimation in time (R = 2, 4, 8, 16), fully unrolled
  static SMT_HD void run(cplx (&v)[R]) {
    cplx e[R / 2], o[R / 2];
#pragma unroll
    for (int i = 0; i < R / 2; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    SmallDft<R / 2, INV>::run(e);
    SmallDft<R / 2, INV>::run(o);
#pragma unroll
    for (int q = 0; q < R / 2; ++q) {
      cplx t;
      if (q == 0) t = o[q];
      else if (4 * q == R) t = INV ? cplx{-o[q].y, o[q].x} : cplx{o[q].y, -o[q].x};      // times -i (forward), +i (inverse)
      else {
        const float c = cos16(q * (16 / R)), sn = sin16(q * (16 / R));
        const cplx w = {c, INV ? sn : -sn};
        t = cmul(o[q], w);
      }
      v[q] = {e[q].x + t.x, e[q].y + t.y};
      v[q + R / 2] = {e[q].x - t.x, e[q].y - t.y};
    }
  }
};

// one in-place Stockham pass of radix R over buf[N] by one wave; ns = product of the radices of the earlier passes
template <int N, int R, bool INV>
__device__ __forceinline__ void fft_wave_pass(cplx* buf, const cplx* __restrict__ tw, int ns, int lane) {
  constexpr int NB = N / R;                     // butterflies
  constexpr int PER = (NB + 63) / 64;           // per lane
  cplx u[PER][R];
  const int tw_stride = N / (R * ns);
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int j = lane + 64 * p;
    if (NB % 64 == 0 || j < NB) {
#pragma unroll
      for (int r = 0; r < R; ++r) u[p][r] = buf[fidx<N, 64>(j + r * NB)];
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int j = lane + 64 * p;
    if (NB % 64 == 0 || j < NB) {
      const int k = j & (ns - 1);
      if (ns > 1) {
#pragma unroll
        for (int r = 1; r < R; ++r) u[p][r] = cmul(u[p][r], fft_twiddle<N>(tw, r * k * tw_stride, INV));
      }
      SmallDft<R, INV>::run(u[p]);
      const int j0 = (j - k) * R + k;
#pragma unroll
      for (int q = 0; q < R; ++q) buf[fidx<N, 64>(j0 + q * ns)] = u[p][q];
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}
template <int N, bool INV>
__device__ __forceinline__ void fft_wave(cplx* buf, const cplx* __restrict__ tw, int lane) {
  static_assert(N == 256 || N == 512 || N == 1024 || N == 2048, "built sizes");
  if constexpr (N == 256) { fft_wave_pass<N, 16, INV>(buf, tw, 1, lane); fft_wave_pass<N, 16, INV>(buf, tw, 16, lane); }
  if constexpr (N == 512) { fft_wave_pass<N, 8, INV>(buf, tw, 1, lane); fft_wave_pass<N, 8, INV>(buf, tw, 8, lane); fft_wave_pass<N, 8, INV>(buf, tw, 64, lane); }
  if constexpr (N == 1024) {
    FFT_T(t0); fft_wave_pass<N, 16, INV>(buf, tw, 1, lane);
    FFT_T(t1); fft_wave_pass<N, 16, INV>(buf, tw, 16, lane);
    FFT_T(t2); fft_wave_pass<N, 4, INV>(buf, tw, 256, lane);
    FFT_T(t3); FFT_ACC(1, t0, t1); FFT_ACC(2, t1, t2); FFT_ACC(3, t2, t3);
  }
  if constexpr (N == 2048) { fft_wave_pass<N, 16, INV>(buf, tw, 1, lane); fft_wave_pass<N, 16, INV>(buf, tw, 16, lane); fft_wave_pass<N, 8, INV>(buf, tw, 256, lane); }
}

__device__ __forceinline__ int reflect_index(int p, int T) {  // F.pad(mode="reflect") source index
  if (p < 0) p = -p;
  if (p >= T) p = 2 * (T - 1) - p;
  return p;
}

// ------------------------------------------------------------- magnitudes ------
// mag[b, k, f] for k in [0, N/2], f in [0, frames)   (reference layout [B, bins, frames])
// Frame geometry of the <N, NT> kernels: 256 / NT frames per workgroup.  Dynamic LDS per frame: two [N] complex buffers in
// the whole-workgroup form; ONE in the one-wave form (its transform runs in place) plus SCRATCH complex slots behind it
// (buf1) for the kernels that need a second array.
template <int N, int NT>
constexpr int frame_slots(int scratch) { return NT == 256 ? 2 * N : frame_padded<N>() + scratch; }
#define SMT_FRAME_PROLOGUE(SCRATCH)                                                         \
  constexpr int FPW = 256 / NT;                                                             \
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];                  \
  const int sub = threadIdx.x / NT, tid = threadIdx.x % NT;                                 \
  const int f = blockIdx.x * FPW + sub, b = blockIdx.y;                                     \
  cplx* buf0 = reinterpret_cast<cplx*>(smem_raw) + (size_t)sub * frame_slots<N, NT>(SCRATCH); \
  cplx* buf1 = buf0 + (NT == 256 ? N : frame_padded<N>());                                  \
  (void)buf1;                                                                               \
  if (NT != 256 && f >= frames) return;      /* wave-uniform; the one-wave form has no workgroup barrier */

template <int N, int NT>
__global__ __launch_bounds__(256) void stft_mag_kernel(const float* __restrict__ x, const float* __restrict__ window,
                                                       const cplx* __restrict__ tw, float* __restrict__ mag, int T,
                                                       int hop, int pad, int frames) {
  SMT_FRAME_PROLOGUE(0)
  const float* xb = x + (long long)b * T;
  // window and samples are loaded unconditionally and multiplied (as the reference's windowed basis does): a "skip the
  // sample where the window is zero" test makes every sample load wait for its window load -- 2 x 16 dependent round trips
  // per frame, measured as HALF of the one-wave kernel's time (tools/fft_phases.sh)
#pragma unroll 16
  for (int n = tid; n < N; n += NT) {
    const float w = window[n];
    const float xv = xb[reflect_index(f * hop + n - pad, T)];
    buf0[fidx<N, NT>(n)] = {w * xv, 0.f};
  }
  fft_sync<NT>();
  const cplx* Z = fft_lds<N, NT>(buf0, buf1, tw, false, tid);
  for (int k = tid; k <= N / 2; k += NT) {
    const cplx z = Z[fidx<N, NT>(k)];
    mag[((long long)b * (N / 2 + 1) + k) * frames + f] = sqrtf(z.x * z.x + z.y * z.y);
  }
}

// ------------------------------------------------------------- loss forward ----
// part[b, f, 0] = sum_k ((|Y| - |Yh|) m)^2, part[b, f, 1] = sum_k ((log|Y| - log|Yh|) m)^2  (clamp 1e-5)
template <int N, int NT>
__global__ __launch_bounds__(256) void stft_loss_fwd_kernel(const float* __restrict__ y, const float* __restrict__ yh,
                                                            const int* __restrict__ lens,
                                                            const float* __restrict__ window,
                                                            const cplx* __restrict__ tw, float* __restrict__ part, int T,
                                                            int hop, int pad, int frames) {
  SMT_FRAME_PROLOGUE(0)
  __shared__ float red[2][4];
  // frame kept iff the sample under its centre tap is unmasked (losses.py:33-37)
  const int len = lens ? lens[b] : T;
  const bool keep = (N / 2 - pad + f * hop) < len;
  float s_lin = 0.f, s_log = 0.f;
  FFT_T(ts0);
  if (keep) {  // uniform over the NT threads of the frame
    const float* yb = y + (long long)b * T;
    const float* hb = yh + (long long)b * T;
#pragma unroll 16
    for (int n = tid; n < N; n += NT) {
      const float w = window[n];
      const int src = reflect_index(f * hop + n - pad, T);
      const float a = yb[src], c = hb[src];
      buf0[fidx<N, NT>(n)] = {w * a, w * c};
    }
    fft_sync<NT>();
    FFT_T(ts1); FFT_ACC(0, ts0, ts1);
    const cplx* Z = fft_lds<N, NT>(buf0, buf1, tw, false, tid);
    FFT_T(ts2);
    for (int k = tid; k <= N / 2; k += NT) {
      const cplx a = Z[fidx<N, NT>(k)], c = Z[fidx<N, NT>((N - k) & (N - 1))];
      const float yr = 0.5f * (a.x + c.x), yi = 0.5f * (a.y - c.y);
      const float hr = 0.5f * (a.y + c.y), hi = -0.5f * (a.x - c.x);
      const float my = sqrtf(yr * yr + yi * yi), mh = sqrtf(hr * hr + hi * hi);
      const float d = my - mh;
      const float dl = __logf(fmaxf(my, 1e-5f)) - __logf(fmaxf(mh, 1e-5f));
      s_lin = fmaf(d, d, s_lin);
      s_log = fmaf(dl, dl, s_log);
    }
#if SMT_FFT_STAMP
    { FFT_T(ts3); FFT_ACC(4, ts2, ts3); FFT_ACC(5, ts0, ts3); FFT_ACC(6, 0ull, 1ull); }
#endif
  }
  s_lin = wave_sum(s_lin);
  s_log = wave_sum(s_log);
  float* o = part + ((long long)b * frames + f) * 2;
  if constexpr (NT == 256) {
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_lin; red[1][threadIdx.x >> 6] = s_log; }
    __syncthreads();
    if (threadIdx.x == 0) {
      o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
      o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
  } else {
    (void)red;
    if (tid == 0) { o[0] = s_lin; o[1] = s_log; }
  }
}

// ------------------------------------------------------------- loss backward ---
// dyh = adjoint( dL/dYh ),  dL/d|Yh| = -c_lin[b] (|Y|-|Yh|) - c_log[b] (log|Y|-log|Yh|) / |Yh| [|Yh| > 1e-5]
// coef[b] = {c_lin, c_log} already contains the upstream gradient and the 1/(2 sqrt(S)) factors.
// Two kernels: this one leaves every live frame's windowed time-domain gradient row in `rows` [B, frames, N]; the gather
// kernel below sums, for every sample, the rows that overlap it IN A FIXED ORDER -- an overlap-add with f32 atomics (the
// first version) made the gradient, and with it the whole train step, differ from run to run in the last bits.
template <int N, int NT>
__global__ __launch_bounds__(256) void stft_loss_bwd_kernel(const float* __restrict__ y, const float* __restrict__ yh,
                                                            const int* __restrict__ lens,
                                                            const float* __restrict__ window,
                                                            const cplx* __restrict__ tw, const float* __restrict__ coef,
                                                            float* __restrict__ rows, int T, int hop, int pad,
                                                            int frames) {
  SMT_FRAME_PROLOGUE(0)
  const int len = lens ? lens[b] : T;
  if (!((N / 2 - pad + f * hop) < len)) return;  // masked frame: no gradient (uniform over the frame's threads); the gather skips its row
  const float* yb = y + (long long)b * T;
  const float* hb = yh + (long long)b * T;
#pragma unroll 16
  for (int n = tid; n < N; n += NT) {
    const float w = window[n];
    const int src = reflect_index(f * hop + n - pad, T);
    const float a = yb[src], c = hb[src];
    buf0[fidx<N, NT>(n)] = {w * a, w * c};
  }
  fft_sync<NT>();
  cplx* Z = fft_lds<N, NT>(buf0, buf1, tw, false, tid);
  // one-sided gradient spectrum (upper half zero).  Whole-workgroup form: into the other buffer.  One-wave form: IN PLACE,
  // bin k and its mirror N - k are read and rewritten by the same lane in the same iteration, and no later iteration
  // (larger k) reads a slot an earlier one wrote (those are < k or > N - k).
  cplx* G = NT == 64 ? Z : ((Z == buf0) ? buf1 : buf0);
  const float c_lin = 2.f * coef[2 * b], c_log = 2.f * coef[2 * b + 1];
  for (int k = tid; k <= N / 2; k += NT) {
    const int km = (N - k) & (N - 1);
    const cplx a = Z[fidx<N, NT>(k)], c = Z[fidx<N, NT>(km)];
    const float yr = 0.5f * (a.x + c.x), yi = 0.5f * (a.y - c.y);
    const float hr = 0.5f * (a.y + c.y), hi = -0.5f * (a.x - c.x);
    const float my = sqrtf(yr * yr + yi * yi), mh = sqrtf(hr * hr + hi * hi);
    float dmag = -c_lin * (my - mh);
    if (mh > 1e-5f) dmag -= c_log * (__logf(fmaxf(my, 1e-5f)) - __logf(mh)) / mh;
    const float inv = mh > 0.f ? dmag / mh : 0.f;
    if (km != k) G[fidx<N, NT>(km)] = {0.f, 0.f};
    G[fidx<N, NT>(k)] = {inv * hr, inv * hi};
  }
  fft_sync<NT>();
  // x_grad[n] = w[n] * Re( sum_k G_k e^{+2 pi i k n / N} )
  cplx* other = (G == buf0) ? buf1 : buf0;
  const cplx* R = fft_lds<N, NT>(G, other, tw, true, tid);
  float* row = rows + ((long long)b * frames + f) * N;
  for (int n = tid; n < N; n += NT) row[n] = window[n] * R[fidx<N, NT>(n)].x;
}

// dyh[b, i] = sum of rows[b, f, n] over the (f, n) whose padded position f hop + n - pad reflects onto sample i
// (F.pad(mode="reflect"): p < 0 -> -p, p >= T -> 2 (T - 1) - p), live frames only, in the order: positions i, -i,
// 2 (T - 1) - i, frames ascending inside each.
__global__ __launch_bounds__(256) void stft_overlap_gather_kernel(const float* __restrict__ rows, const int* __restrict__ lens,
                                                                  float* __restrict__ dyh, int T, int N, int hop, int pad,
                                                                  int frames) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= T) return;
  const int len = lens ? lens[b] : T;
  const float* rb = rows + (long long)b * frames * N;
  const int pmax = (frames - 1) * hop + N - 1 - pad;          // last padded position any frame touches
  int cand[3] = {i, -i, 2 * (T - 1) - i};
  const bool use[3] = {true, i >= 1 && i <= pad, i <= T - 2 && cand[2] <= pmax};
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (!use[c]) continue;
    const int q = cand[c] + pad;                              // offset in the padded signal, >= 0
    const int f_hi = min(frames - 1, q / hop);
    const int f_lo = q - N + 1 > 0 ? (q - N + hop) / hop : 0; // ceil((q - N + 1) / hop)
    for (int f = f_lo; f <= f_hi; ++f)
      if ((N / 2 - pad + f * hop) < len) s += rb[(long long)f * N + (q - f * hop)];
  }
  dyh[(long long)b * T + i] = s;
}


// ------------------------------------------------------------- log-mel ---------
// mel[b, m, f] = log(max(sum_k basis[m][k] |X_k|, 1e-5)); the triangular filters are sparse, so
// each mel bin only walks its own band [lo[m], hi[m]).  (MelSpectrogram.forward, transforms.py:61-65)
template <int N, int NT>
__global__ __launch_bounds__(256) void melspec_kernel(const float* __restrict__ x, const float* __restrict__ window,
                                                      const cplx* __restrict__ tw, const float* __restrict__ basis,
                                                      const int* __restrict__ band, float* __restrict__ mel, int T,
                                                      int hop, int pad, int frames, int n_mels) {
  SMT_FRAME_PROLOGUE(N / 2)              // one-wave form: N floats behind the frame for the magnitudes
  const float* xb = x + (long long)b * T;
#pragma unroll 16
  for (int n = tid; n < N; n += NT) {
    const float w = window[n];
    const float xv = xb[reflect_index(f * hop + n - pad, T)];
    buf0[fidx<N, NT>(n)] = {w * xv, 0.f};
  }
  fft_sync<NT>();
  cplx* Z = fft_lds<N, NT>(buf0, buf1, tw, false, tid);
  // magnitudes into the buffer the transform did not end in (as floats)
  float* magn = reinterpret_cast<float*>(Z == buf0 ? buf1 : buf0);
  for (int k = tid; k <= N / 2; k += NT) {
    const cplx z = Z[fidx<N, NT>(k)];
    magn[k] = sqrtf(z.x * z.x + z.y * z.y);
  }
  fft_sync<NT>();
  // a lane walks the band of its mel bin (up to ~60 bins at the top of the scale): eight independent basis loads per trip, so
  // the walk costs band/8 L2 round trips instead of one per bin (the one-load-per-trip form was most of this kernel's time)
  for (int m = tid; m < n_mels; m += NT) {
    float s = 0.f;
    const float* row = basis + (long long)m * (N / 2 + 1);
    const int lo = band[2 * m], hi = band[2 * m + 1];
    for (int k = lo; k < hi; k += 8) {
      float w[8], v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = min(k + u, hi - 1);
        w[u] = row[kk];
        v[u] = magn[kk];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s = fmaf(k + u < hi ? w[u] : 0.f, v[u], s);      // same order of additions as before
    }
    mel[((long long)b * n_mels + m) * frames + f] = logf(fmaxf(s, 1e-5f));
  }
}


// ------------------------------------------------------------------------------------------------------------------
// One-wave-per-frame kernels, second step (round 3).  tools/fft_phases.sh showed where the first one-wave form
// spent a frame: a third in the loads (48 dword loads per lane through the 64 B/clk L1 path), a fifth in pass 2 waiting for
// its 15 gathered twiddles (whose index arithmetic also cost more VALU than a complex multiply), a fifth in the serial
// spectra loop.  Here a wave loads ONE twiddle per pass and butterfly (the R - 1 others are its powers, <= 6 roundings
// deep); interior frames and the window are read as 16-byte pieces (4 consecutive samples per lane); the spectra loop is
// unrolled.
#define WAVE_LB (N <= 1024 ? 4 : 2)     // waves per SIMD the LDS frames allow
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };   // 16-byte load at 4-byte alignment (hop need not divide 4)

template <int N>
struct WaveFft {
  static constexpr int R1 = (N == 512) ? 8 : 16;          // passes 1 and 2
  static constexpr int R3 = N / (R1 * R1);                // pass 3: 8 (512), 4 (1024), 8 (2048), none (256)
  static constexpr int NB1 = N / R1, PER1 = (NB1 + 63) / 64;
  static constexpr int NB3 = R3 > 1 ? N / R3 : 64, PER3 = NB3 / 64;
  cplx w2;                                                // exp(-2 pi i (lane & (R1 - 1)) R3 / N)
  cplx w3[PER3];                                          // exp(-2 pi i (lane + 64 p) / N)

  SMT_HD void load(const cplx* __restrict__ tw, int lane) {
    w2 = tw[(lane & (R1 - 1)) * (R3 > 1 ? R3 : 1)];
#pragma unroll
    for (int p = 0; p < PER3; ++p) w3[p] = tw[lane + 64 * p];
  }
};

SMT_HD void wave_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
#endif
}

// one pass of one lane: reads `in`, writes `out`; w1[p] = the twiddle of butterfly p of this lane (ignored when ns == 1).
// On the device in == out (every lane of the wave has read before any lane writes: LDS operations of a wave execute in
// order); the host test passes a snapshot as `in`.
template <int N, int R, int PER, bool INV>
SMT_HD void wave_pass(const cplx* in, cplx* out, int ns, int lane, const cplx (&w1)[PER]) {
  constexpr int NB = N / R;
  cplx u[PER][R];
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int j = lane + 64 * p;
    if (NB >= 64 || j < NB) {
#pragma unroll
      for (int r = 0; r < R; ++r) u[p][r] = in[fidx<N, 64>(j + r * NB)];
    }
  }
  wave_fence();
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int j = lane + 64 * p;
    if (NB >= 64 || j < NB) {
      const int k = j & (ns - 1);
      if (ns > 1) {
        // u[r] *= w1^r with few registers live: w1, w1^2, w1^3 and a running (w1^4)^a; at most 6 roundings deep
        cplx wa = w1[p];
        if (INV) wa.y = -wa.y;
        const cplx wb = cmul(wa, wa), wc3 = cmul(wb, wa), w4 = cmul(wb, wb);
        u[p][1] = cmul(u[p][1], wa);
        if (R > 2) { u[p][2] = cmul(u[p][2], wb); u[p][3] = cmul(u[p][3], wc3); }
        cplx run = w4;
#pragma unroll
        for (int a = 1; a < R / 4; ++a) {
          u[p][4 * a] = cmul(u[p][4 * a], run);
          u[p][4 * a + 1] = cmul(u[p][4 * a + 1], cmul(run, wa));
          u[p][4 * a + 2] = cmul(u[p][4 * a + 2], cmul(run, wb));
          u[p][4 * a + 3] = cmul(u[p][4 * a + 3], cmul(run, wc3));
          if (a + 1 < R / 4) run = cmul(run, w4);
        }
      }
      SmallDft<R, INV>::run(u[p]);
      const int j0 = (j - k) * R + k;
#pragma unroll
      for (int q = 0; q < R; ++q) out[fidx<N, 64>(j0 + q * ns)] = u[p][q];
    }
  }
  wave_fence();
}

template <int N, bool INV>
__device__ __forceinline__ void wave_fft(cplx* buf, const WaveFft<N>& wc, int lane) {
  using W = WaveFft<N>;
  cplx w1a[W::PER1];
#pragma unroll
  for (int p = 0; p < W::PER1; ++p) w1a[p] = wc.w2;
  wave_pass<N, W::R1, W::PER1, INV>(buf, buf, 1, lane, w1a);
  wave_pass<N, W::R1, W::PER1, INV>(buf, buf, W::R1, lane, w1a);
  if constexpr (W::R3 > 1) wave_pass<N, W::R3, W::PER3, INV>(buf, buf, W::R1 * W::R1, lane, wc.w3);
}

// windowed frame f of (ya [+ i yb]) into buf; TWO = two real signals packed as one complex one
template <int N, bool TWO>
__device__ __forceinline__ void wave_load_frame(cplx* buf, const WaveFft<N>& wc, const float* __restrict__ ya,
                                                const float* __restrict__ yb, const float* __restrict__ window, int p0,
                                                int T, int lane) {
  if (p0 >= 0 && p0 + N <= T) {                       // interior frame (wave-uniform): no reflection, 16-byte pieces
    f4u a[N / 256], c[N / 256], wv[N / 256];
#pragma unroll
    for (int i = 0; i < N / 256; ++i) {
      a[i] = *reinterpret_cast<const f4u*>(ya + p0 + 4 * lane + 256 * i);
      if (TWO) c[i] = *reinterpret_cast<const f4u*>(yb + p0 + 4 * lane + 256 * i);
      wv[i] = *reinterpret_cast<const f4u*>(window + 4 * lane + 256 * i);
    }
#pragma unroll
    for (int i = 0; i < N / 256; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w = wv[i].v[e];
        buf[fidx<N, 64>(4 * lane + 256 * i + e)] = {w * a[i].v[e], TWO ? w * c[i].v[e] : 0.f};
      }
  } else {
#pragma unroll 4
    for (int n = lane; n < N; n += 64) {
      const float w = window[n];
      const int src = reflect_index(p0 + n, T);
      buf[fidx<N, 64>(n)] = {w * ya[src], TWO ? w * yb[src] : 0.f};
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// the two real spectra packed in Z at bin k: Y = (Z_k + conj Z_{N-k}) / 2, Yh = (Z_k - conj Z_{N-k}) / 2i
__device__ __forceinline__ void split_bins(cplx a, cplx c, float& yr, float& yi, float& hr, float& hi) {
  yr = 0.5f * (a.x + c.x); yi = 0.5f * (a.y - c.y);
  hr = 0.5f * (a.y + c.y); hi = -0.5f * (a.x - c.x);
}

template <int N>
__global__ __launch_bounds__(256, WAVE_LB) void stft_loss_fwd_wave_kernel(const float* __restrict__ y, const float* __restrict__ yh,
                                                                 const int* __restrict__ lens,
                                                                 const float* __restrict__ window,
                                                                 const cplx* __restrict__ tw, float* __restrict__ part,
                                                                 int T, int hop, int pad, int frames) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, b = blockIdx.y;
  int lane = threadIdx.x & 63;
  cplx* Z = reinterpret_cast<cplx*>(smem_raw) + (size_t)wave * frame_padded<N>();
  WaveFft<N> wc;
  wc.load(tw, lane);
  const int len = lens ? lens[b] : T;
  const float* yb = y + (long long)b * T;
  const float* hb = yh + (long long)b * T;
  const int f = blockIdx.x * 4 + wave;      // one frame per wave (a frame LOOP makes the optimiser hoist the twiddle powers
  if (f < frames) {                         // and every LDS address out of it as invariants: 0.5-1.6 KB of spills per lane)
    float s_lin = 0.f, s_log = 0.f;
    if ((N / 2 - pad + f * hop) < len) {          // frame kept iff the sample under its centre tap is unmasked
        wave_load_frame<N, true>(Z, wc, yb, hb, window, f * hop - pad, T, lane);
      wave_fft<N, false>(Z, wc, lane);
      auto bin = [&](int k) {
        const cplx a = Z[fidx<N, 64>(k)], c = Z[fidx<N, 64>((N - k) & (N - 1))];
        float yr, yi, hr, hi;
        split_bins(a, c, yr, yi, hr, hi);
        const float my = sqrtf(yr * yr + yi * yi), mh = sqrtf(hr * hr + hi * hi);
        const float d = my - mh;
        const float dl = __logf(fmaxf(my, 1e-5f)) - __logf(fmaxf(mh, 1e-5f));
        s_lin = fmaf(d, d, s_lin);
        s_log = fmaf(dl, dl, s_log);
      };
#pragma unroll 4
      for (int i = 0; i < N / 128; ++i) bin(lane + 64 * i);
      if (lane == 0) bin(N / 2);
      __builtin_amdgcn_wave_barrier();             // the next frame's loads overwrite Z
    }
    s_lin = wave_sum(s_lin);
    s_log = wave_sum(s_log);
    if (lane == 0) {
      float* o = part + ((long long)b * frames + f) * 2;
      o[0] = s_lin; o[1] = s_log;
    }
  }
}

template <int N>
__global__ __launch_bounds__(256, WAVE_LB) void stft_loss_bwd_wave_kernel(const float* __restrict__ y, const float* __restrict__ yh,
                                                                 const int* __restrict__ lens,
                                                                 const float* __restrict__ window,
                                                                 const cplx* __restrict__ tw, const float* __restrict__ coef,
                                                                 float* __restrict__ rows, int T, int hop, int pad,
                                                                 int frames) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, b = blockIdx.y;
  int lane = threadIdx.x & 63;
  cplx* Z = reinterpret_cast<cplx*>(smem_raw) + (size_t)wave * frame_padded<N>();
  WaveFft<N> wc;
  wc.load(tw, lane);
  const int len = lens ? lens[b] : T;
  const float* yb = y + (long long)b * T;
  const float* hb = yh + (long long)b * T;
  const float c_lin = 2.f * coef[2 * b], c_log = 2.f * coef[2 * b + 1];
  const int f = blockIdx.x * 4 + wave;
  if (f < frames && (N / 2 - pad + f * hop) < len) {  // masked frame: no gradient; the gather skips its row
    wave_load_frame<N, true>(Z, wc, yb, hb, window, f * hop - pad, T, lane);
    wave_fft<N, false>(Z, wc, lane);
    // one-sided gradient spectrum IN PLACE: bin k and its mirror N - k are read and rewritten by the same lane in the same
    // step, and no later step (larger k) reads a slot an earlier one wrote (those are < k or > N - k)
    auto bin = [&](int k) {
      const int km = (N - k) & (N - 1);
      const cplx a = Z[fidx<N, 64>(k)], c = Z[fidx<N, 64>(km)];
      float yr, yi, hr, hi;
      split_bins(a, c, yr, yi, hr, hi);
      const float my = sqrtf(yr * yr + yi * yi), mh = sqrtf(hr * hr + hi * hi);
      float dmag = -c_lin * (my - mh);
      if (mh > 1e-5f) dmag -= c_log * (__logf(fmaxf(my, 1e-5f)) - __logf(mh)) / mh;
      const float inv = mh > 0.f ? dmag / mh : 0.f;
      if (km != k) Z[fidx<N, 64>(km)] = {0.f, 0.f};
      Z[fidx<N, 64>(k)] = {inv * hr, inv * hi};
    };
#pragma unroll 2
    for (int i = 0; i < N / 128; ++i) bin(lane + 64 * i);
    if (lane == 0) bin(N / 2);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    wave_fft<N, true>(Z, wc, lane);                   // x_grad[n] = w[n] Re(sum_k G_k e^{+2 pi i k n / N})
    float* row = rows + ((long long)b * frames + f) * N;
#pragma unroll
    for (int i = 0; i < N / 256; ++i) {
      const float4 w = *reinterpret_cast<const float4*>(window + 4 * lane + 256 * i);
      float4 o;
      o.x = w.x * Z[fidx<N, 64>(4 * lane + 256 * i)].x;
      o.y = w.y * Z[fidx<N, 64>(4 * lane + 256 * i + 1)].x;
      o.z = w.z * Z[fidx<N, 64>(4 * lane + 256 * i + 2)].x;
      o.w = w.w * Z[fidx<N, 64>(4 * lane + 256 * i + 3)].x;
      *reinterpret_cast<float4*>(row + 4 * lane + 256 * i) = o;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
}

template <int N>
__global__ __launch_bounds__(256, WAVE_LB) void melspec_wave_kernel(const float* __restrict__ x, const float* __restrict__ window,
                                                           const cplx* __restrict__ tw, const float* __restrict__ basis,
                                                           const int* __restrict__ band, float* __restrict__ mel, int T,
                                                           int hop, int pad, int frames, int n_mels) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, b = blockIdx.y;
  int lane = threadIdx.x & 63;
  constexpr int SLOTS = frame_padded<N>() + N / 2;    // N floats behind the frame for the magnitudes
  cplx* Z = reinterpret_cast<cplx*>(smem_raw) + (size_t)wave * SLOTS;
  float* magn = reinterpret_cast<float*>(Z + frame_padded<N>());
  WaveFft<N> wc;
  wc.load(tw, lane);
  const float* xb = x + (long long)b * T;
  const int f = blockIdx.x * 4 + wave;
  if (f < frames) {
    wave_load_frame<N, false>(Z, wc, xb, xb, window, f * hop - pad, T, lane);
    wave_fft<N, false>(Z, wc, lane);
#pragma unroll 4
    for (int i = 0; i < N / 128; ++i) {
      const cplx z = Z[fidx<N, 64>(lane + 64 * i)];
      magn[lane + 64 * i] = sqrtf(z.x * z.x + z.y * z.y);
    }
    if (lane == 0) { const cplx z = Z[fidx<N, 64>(N / 2)]; magn[N / 2] = sqrtf(z.x * z.x + z.y * z.y); }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    for (int m = lane; m < n_mels; m += 64) {
      float s = 0.f;
      const float* row = basis + (long long)m * (N / 2 + 1);
      const int lo = band[2 * m], hi = band[2 * m + 1];
      for (int k = lo; k < hi; k += 8) {
        float w[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int kk = min(k + u, hi - 1);
          w[u] = row[kk];
          v[u] = magn[kk];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s = fmaf(k + u < hi ? w[u] : 0.f, v[u], s);
      }
      mel[((long long)b * n_mels + m) * frames + f] = logf(fmaxf(s, 1e-5f));
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
}

// test hook: the one-wave transform alone on one complex frame (tests/test_spectral_gpu.py)
template <int N>
__global__ __launch_bounds__(64) void fft_selftest_kernel(const cplx* __restrict__ in, const cplx* __restrict__ tw,
                                                          cplx* __restrict__ out, int inverse) {
  __shared__ cplx Z[frame_padded<N>()];
  int lane = threadIdx.x;
  WaveFft<N> wc;
  wc.load(tw, lane);
  for (int n = lane; n < N; n += 64) Z[fidx<N, 64>(n)] = in[n];
  wave_fence();
  if (inverse) wave_fft<N, true>(Z, wc, lane); else wave_fft<N, false>(Z, wc, lane);
  for (int n = lane; n < N; n += 64) out[n] = Z[fidx<N, 64>(n)];
}

}  // namespace smt

using namespace smt;

// NT = 64: one wave per frame, four frames per workgroup, in-place radix-8/16 passes (fft_wave).  SMT_FFT_NT=256 in the
// environment selects the round-2 form (the whole workgroup on one frame, radix-4 passes over two buffers) for A/B runs.
// (Round 3 first tried the one-wave form WITH the radix-4 two-buffer transform: slower, melspec 189 vs 133 us -- two [N]
// buffers per frame capped the occupancy at 8 waves per CU and a lone wave paid the LDS latency of each of the 5-6 passes.)
static int fft_nt() {
  static const int nt = [] { const char* e = getenv("SMT_FFT_NT"); return (e && atoi(e) == 256) ? 256 : 64; }();
  return nt;
}
#define SMT_FFT_DISPATCH_NT(NN, CALL)                                       \
  if (fft_nt() == 256) { constexpr int N = NN, NT = 256; (void)NT; CALL; }  \
  else { constexpr int N = NN, NT = 64; (void)NT; CALL; }
#define SMT_FFT_DISPATCH(NFFT, CALL)                                        \
  switch (NFFT) {                                                           \
    case 256: { SMT_FFT_DISPATCH_NT(256, CALL) } break;                     \
    case 512: { SMT_FFT_DISPATCH_NT(512, CALL) } break;                     \
    case 1024: { SMT_FFT_DISPATCH_NT(1024, CALL) } break;                   \
    case 2048: { SMT_FFT_DISPATCH_NT(2048, CALL) } break;                   \
    default:                                                                \
      set_error("stft: n_fft=%d unsupported (256, 512, 1024, 2048)", NFFT); \
      return 1;                                                             \
  }

// ------------------------------------------------------------- inverse ------
// STFT.inverse (datasets/transforms.py:125-156).  The reference multiplies [mag cos(phi); mag sin(phi)] by the
// pseudo-inverse of its stacked real DFT basis (conv_transpose1d, stride = hop): the rows of that basis are orthogonal, so
// the pseudo-inverse IS the inverse real FFT, x[n] = (1/N)(X_0 + (-1)^n X_{N/2}) + (2/N) sum_{0<k<N/2} Re(X_k e^{+2 pi i k n/N})
// (the imaginary parts of bins 0 and N/2 meet zero rows and drop out), times 1/scale; then the synthesis window, the
// overlap-add, the division by the window sum-square (librosa.filters.window_sumsquare) where it is above float tiny,
// the factor scale = n_fft / hop (cancelling the 1/scale), and the trim of pad_amount samples on both sides.
// One workgroup per frame: one-sided weighted spectrum -> complex in-LDS FFT with conjugate twiddles -> windowed real
// part added into the trimmed output (f32 atomics); a second elementwise kernel divides by the window sum-square.
template <int N>
__global__ __launch_bounds__(256) void stft_inverse_kernel(const float* __restrict__ mag, const float* __restrict__ phase,
                                                           const float* __restrict__ window, const cplx* __restrict__ tw,
                                                           float* __restrict__ out, int hop, int pad, int frames, int t_out) {
  __shared__ cplx buf[2][N];
  const int f = blockIdx.x, b = blockIdx.y;
  const size_t bins = N / 2 + 1;
  const float* mb = mag + (size_t)b * bins * frames;
  const float* pb = phase + (size_t)b * bins * frames;
  for (int k = threadIdx.x; k < N; k += 256) {
    cplx g = {0.f, 0.f};
    if (k <= N / 2) {
      const float m = mb[(size_t)k * frames + f], ph = pb[(size_t)k * frames + f];
      float s, c;
      sincosf(ph, &s, &c);
      const bool edge = (k == 0) || (k == N / 2);
      const float wgt = edge ? 1.f / (float)N : 2.f / (float)N;
      g = {wgt * m * c, edge ? 0.f : wgt * m * s};
    }
    buf[0][k] = g;
  }
  __syncthreads();
  const cplx* R = fft_lds<N>(buf[0], buf[1], tw, true);     // R[n].x = sum_k Re(G_k e^{+2 pi i k n / N})
  float* ob = out + (size_t)b * t_out;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float w = window[n];
    const int i = f * hop + n - pad;                         // position in the trimmed signal
    if (w != 0.f && i >= 0 && i < t_out) atomicAdd(ob + i, w * R[n].x);
  }
}

__global__ __launch_bounds__(256) void stft_inverse_norm_kernel(const float* __restrict__ window, float* __restrict__ out,
                                                                int n_fft, int hop, int pad, int frames, int t_out, int batch) {
  const long long total = (long long)batch * t_out;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int i = (int)(e % t_out) + pad;                    // position in the untrimmed overlap-add
    // window sum-square at i: frames f with 0 <= i - f hop < n_fft
    const int f_hi = min(frames - 1, i / hop);
    const int f_lo = max(0, (i - n_fft + hop) / hop);
    float wss = 0.f;
    for (int f = f_lo; f <= f_hi; ++f) {
      const int n = i - f * hop;
      if (n >= 0 && n < n_fft) { const float w = window[n]; wss += w * w; }
    }
    if (wss > 1.17549435e-38f) out[e] = out[e] / wss;        // librosa.util.tiny(float32)
  }
}

// launch of an <N, NT> frame kernel: 256 / NT frames per workgroup, `slots` complex LDS slots per frame (frame_slots)
template <typename K, typename... Args>
static void frame_launch(K kernel, int slots, int nt, int frames, int batch, hipStream_t stream, Args... args) {
  const int fpw = 256 / nt;
  const size_t lds = (size_t)fpw * slots * sizeof(cplx);
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kernel<<<dim3((frames + fpw - 1) / fpw, batch), 256, lds, stream>>>(args...);
}

#if SMT_FFT_STAMP
extern "C" int smt_fft_debug_dump(unsigned long long* host, int reset) {      // host: [8192][8]
  int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fft_dbg), sizeof(unsigned long long) * FFT_DBG_WAVES * 8);
  if (reset) { static unsigned long long z[FFT_DBG_WAVES * 8]; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(fft_dbg), z, sizeof(z)); }
  return rc;
}
#endif

// launch of a one-wave-per-frame kernel: four frames per workgroup
template <typename K, typename... Args>
static void wave_launch(K kernel, int slots, int frames, int batch, hipStream_t stream, Args... args) {
  const size_t lds = (size_t)4 * slots * sizeof(cplx);
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kernel<<<dim3((frames + 3) / 4, batch), 256, lds, stream>>>(args...);
}
#define SMT_WAVE_DISPATCH(NFFT, CALL)                                       \
  switch (NFFT) {                                                           \
    case 256: { constexpr int N = 256; CALL; } break;                       \
    case 512: { constexpr int N = 512; CALL; } break;                       \
    case 1024: { constexpr int N = 1024; CALL; } break;                     \
    case 2048: { constexpr int N = 2048; CALL; } break;                     \
    default:                                                                \
      set_error("stft: n_fft=%d unsupported (256, 512, 1024, 2048)", NFFT); \
      return 1;                                                             \
  }

static int stft_frames(int T, int n_fft, int hop) { return (T + 2 * ((n_fft - hop) / 2) - n_fft) / hop + 1; }

extern "C" int smt_stft_num_frames(int t, int n_fft, int hop) { return stft_frames(t, n_fft, hop); }

extern "C" int smt_stft_magnitude(const float* x, const float* window, const float* twiddle, float* mag, int batch,
                                  int t, int n_fft, int hop, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(x && window && twiddle && mag, "smt_stft_magnitude: null pointer");
  const int pad = (n_fft - hop) / 2;
  SMT_CHECK_ARG(t > pad, "smt_stft_magnitude: signal shorter than the reflect padding");
  const int frames = stft_frames(t, n_fft, hop);
  if (batch == 0 || frames <= 0) return 0;
  SMT_FFT_DISPATCH(n_fft, (frame_launch(stft_mag_kernel<N, NT>, frame_slots<N, NT>(0), NT, frames, batch, stream, x, window, (const cplx*)twiddle, mag, t, hop,
                                        pad, frames)));
  SMT_CHECK_LAUNCH("stft_mag");
  return 0;
}

extern "C" int smt_stft_loss_fwd(const float* y, const float* yh, const int* lens, const float* window,
                                 const float* twiddle, float* partial, int batch, int t, int n_fft, int hop,
                                 smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(y && yh && window && twiddle && partial, "smt_stft_loss_fwd: null pointer");
  const int pad = (n_fft - hop) / 2;
  SMT_CHECK_ARG(t > pad, "smt_stft_loss_fwd: signal shorter than the reflect padding");
  const int frames = stft_frames(t, n_fft, hop);
  if (batch == 0 || frames <= 0) return 0;
  if (fft_nt() == 64) {
    SMT_WAVE_DISPATCH(n_fft, (wave_launch(stft_loss_fwd_wave_kernel<N>, frame_padded<N>(), frames, batch, stream, y, yh, lens, window,
                                          (const cplx*)twiddle, partial, t, hop, pad, frames)));
  } else {
    SMT_FFT_DISPATCH(n_fft, (frame_launch(stft_loss_fwd_kernel<N, NT>, frame_slots<N, NT>(0), NT, frames, batch, stream, y, yh, lens, window, (const cplx*)twiddle,
                                          partial, t, hop, pad, frames)));
  }
  SMT_CHECK_LAUNCH("stft_loss_fwd");
  return 0;
}

extern "C" size_t smt_stft_loss_bwd_workspace_bytes(int batch, int t, int n_fft, int hop) {
  const int frames = stft_frames(t, n_fft, hop);
  return frames <= 0 ? 0 : (size_t)batch * frames * n_fft * sizeof(float);
}

extern "C" int smt_stft_loss_bwd(const float* y, const float* yh, const int* lens, const float* window,
                                 const float* twiddle, const float* coef, float* dyh, int batch, int t, int n_fft,
                                 int hop, void* workspace, size_t workspace_bytes, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(y && yh && window && twiddle && coef && dyh, "smt_stft_loss_bwd: null pointer");
  const int pad = (n_fft - hop) / 2;
  const int frames = stft_frames(t, n_fft, hop);
  if (batch == 0 || t <= 0) return 0;
  if (frames <= 0) { (void)hipMemsetAsync(dyh, 0, (size_t)batch * t * sizeof(float), stream); return 0; }
  SMT_CHECK_ARG(t > pad, "smt_stft_loss_bwd: signal shorter than the reflect padding");
  SMT_CHECK_ARG(workspace && workspace_bytes >= smt_stft_loss_bwd_workspace_bytes(batch, t, n_fft, hop),
                "smt_stft_loss_bwd: workspace too small");
  float* rows = (float*)workspace;
  if (fft_nt() == 64) {
    SMT_WAVE_DISPATCH(n_fft, (wave_launch(stft_loss_bwd_wave_kernel<N>, frame_padded<N>(), frames, batch, stream, y, yh, lens, window,
                                          (const cplx*)twiddle, coef, rows, t, hop, pad, frames)));
  } else {
    SMT_FFT_DISPATCH(n_fft, (frame_launch(stft_loss_bwd_kernel<N, NT>, frame_slots<N, NT>(0), NT, frames, batch, stream, y, yh, lens, window, (const cplx*)twiddle,
                                          coef, rows, t, hop, pad, frames)));
  }
  SMT_CHECK_LAUNCH("stft_loss_bwd");
  stft_overlap_gather_kernel<<<dim3((t + 255) / 256, batch), 256, 0, stream>>>(rows, lens, dyh, t, n_fft, hop, pad, frames);
  SMT_CHECK_LAUNCH("stft_overlap_gather");
  return 0;
}

extern "C" int smt_melspec(const float* x, const float* window, const float* twiddle, const float* mel_basis,
                           const int* band, float* mel, int batch, int t, int n_fft, int hop, int n_mels,
                           smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(x && window && twiddle && mel_basis && band && mel, "smt_melspec: null pointer");
  const int pad = (n_fft - hop) / 2;
  SMT_CHECK_ARG(t > pad, "smt_melspec: signal shorter than the reflect padding");
  const int frames = stft_frames(t, n_fft, hop);
  if (batch == 0 || frames <= 0) return 0;
  if (fft_nt() == 64) {
    SMT_WAVE_DISPATCH(n_fft, (wave_launch(melspec_wave_kernel<N>, frame_padded<N>() + N / 2, frames, batch, stream, x, window,
                                          (const cplx*)twiddle, mel_basis, band, mel, t, hop, pad, frames, n_mels)));
  } else {
    SMT_FFT_DISPATCH(n_fft, (frame_launch(melspec_kernel<N, NT>, frame_slots<N, NT>(N / 2), NT, frames, batch, stream, x, window, (const cplx*)twiddle, mel_basis, band,
                                          mel, t, hop, pad, frames, n_mels)));
  }
  SMT_CHECK_LAUNCH("melspec");
  return 0;
}

extern "C" int smt_fft_selftest(const float* in, const float* twiddle, float* out, int n_fft, int inverse, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(in && twiddle && out, "smt_fft_selftest: null pointer");
  SMT_WAVE_DISPATCH(n_fft, (fft_selftest_kernel<N><<<1, 64, 0, stream>>>((const cplx*)in, (const cplx*)twiddle, (cplx*)out, inverse)));
  SMT_CHECK_LAUNCH("fft_selftest");
  return 0;
}

extern "C" int smt_stft_inverse(const float* magnitude, const float* phase, const float* window, const float* twiddle,
                                float* out, int batch, int n_fft, int hop, int frames, smt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SMT_CHECK_ARG(magnitude && phase && window && twiddle && out, "smt_stft_inverse: null pointer");
  const int pad = (n_fft - hop) / 2;
  const int t_out = (frames - 1) * hop + n_fft - 2 * pad;
  if (batch <= 0 || frames <= 0 || t_out <= 0) return 0;
  (void)hipMemsetAsync(out, 0, (size_t)batch * t_out * sizeof(float), stream);
  dim3 grid(frames, batch);
  SMT_FFT_DISPATCH(n_fft, (stft_inverse_kernel<N><<<grid, 256, 0, stream>>>(magnitude, phase, window, (const cplx*)twiddle, out,
                                                                          hop, pad, frames, t_out)));
  SMT_CHECK_LAUNCH("stft_inverse");
  const long long total = (long long)batch * t_out;
  stft_inverse_norm_kernel<<<(unsigned)std::min<long long>(2048, (total + 255) / 256), 256, 0, stream>>>(window, out, n_fft, hop,
                                                                                                   pad, frames, t_out, batch);
  SMT_CHECK_LAUNCH("stft_inverse_norm");
  return 0;
}
