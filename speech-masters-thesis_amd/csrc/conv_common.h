// Pieces shared by the convolution kernels: element traits, 16-byte vectors, counter-based dropout.
#pragma once
#include <algorithm>
#include <cstdlib>

#include "smt_common.h"

namespace smt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// ---- counter-based dropout (spec: include/smt_hip.h "dropout") -------------------------------
__device__ __forceinline__ unsigned fmix32(unsigned h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
// keep-bit of linear element index i: 16 random bits per element, two elements per hash
__device__ __forceinline__ bool drop_keep(unsigned long long i, unsigned key, unsigned thresh16) {
  unsigned h = fmix32((unsigned)(i >> 1) * 0x9E3779B1u + key);
  unsigned bits = (h >> (16 * (unsigned)(i & 1))) & 0xFFFFu;
  return bits >= thresh16;
}

template <typename T> struct Tr;
template <> struct Tr<__bf16> {
  static constexpr int EPV = 8;      // elements per 16-byte vector
  static constexpr int CCH = 128;    // input channels staged per LDS A chunk
  static constexpr int KC = 128;     // K per weight chunk (one barrier per tap for 128 input channels)
  static constexpr int BM = 128;
};
template <> struct Tr<float> {
  static constexpr int EPV = 4;
  static constexpr int CCH = 64;
  static constexpr int KC = 64;
  static constexpr int BM = 64;
};

template <typename T, int EPV> struct Vec { T v[EPV]; } __attribute__((aligned(16)));

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

// tanh for the gate (resnet.py:233): 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp -- 5 instructions instead of the
// ~40 of tanhf; absolute error <= 2e-7 (the gate's output is O(1) and is stored in bf16 on the fast path), exact limits
// +-1 at +-inf.  Every gate kernel (forward, fused forward, backward) uses this one function, so they stay consistent.
__device__ __forceinline__ float gate_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);        // exp(2x) = 2^(2x log2 e)
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// LDS-DMA: one wave-instruction copies 64 x 16 B (per-lane global source) to 1 KiB of LDS at a wave-uniform base
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_dst_wave_base, 16, 0, 0);
}

// Buffer addressing for the streaming kernels: a range-checked V# over `bytes` bytes from base + byte_off -- per-lane
// offsets are 32-bit, loads beyond the range return zero and stores beyond it are dropped (no per-lane bounds selects, no
// zero page), and every such instruction IS issued, so `s_waitcnt vmcnt(N)` counts stay compile-time constants.
typedef int i32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(const void* base, long long byte_off, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + byte_off, 0, (int)bytes, 0x00020000);
}
// LDS-DMA through a V#: 64 x 16 B (per-lane byte offset) to 1 KiB of LDS at a wave-uniform base
__device__ __forceinline__ void ws_dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, 0, 0, 0);
}
// The same instruction as inline asm, for loops that own their vmcnt waits: the compiler tracks LDS-DMA builtins and puts a
// wait for every DMA in flight in front of the first LDS intrinsic it cannot disambiguate (ds_read_b64_tr_b16: measured in
// conv1x1_bwd, a vmcnt wait for the NEXT tile's prefetch in the middle of the current tile) and answers the first use of
// any ordinary load with vmcnt(0).  Untracked, nothing is inserted; the caller's counted waits are the only ones.
struct UntrackedRsrc { i32x4v w; };
__device__ __forceinline__ UntrackedRsrc untracked_rsrc(const void* base, long long byte_off, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base) + (unsigned long long)byte_off;
  UntrackedRsrc r;
  r.w = i32x4v{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu)),
               __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
  return r;
}
__device__ __forceinline__ void untracked_dma16(const UntrackedRsrc& rs, unsigned voff, const void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(
      (int)(unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) const void*)lds_wave_base));
  // Hazards the compiler pads for its own instructions but not inside inline asm: an SALU write of M0 needs one wait state
  // before an LDS-DMA reads it, and an SGPR written by the VALU (v_readfirstlane: the descriptor words, the LDS address) needs
  // five before a vector-memory instruction reads it.  s_nop 4 after the M0 write covers both.
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(rs.w), "s"(m0v) : "memory");
}   // writes M0 (not declarable as a clobber: reserved): do not mix with the ws_dma16 builtin in one kernel

// A 16-byte global load the COMPILER DOES NOT TRACK (scalar base + 32-bit lane offset).  While LDS-DMA is in flight hipcc
// answers the first use of any ordinary load result with s_waitcnt vmcnt(0), which also drains the DMA of the NEXT tile and
// serialises a double-buffered stream (measured in round 3 on conv_k3gate: every step waited for its own prefetch).  The
// caller owns the wait: s_waitcnt vmcnt(N) with N = the vector-memory instructions issued after this one.
__device__ __forceinline__ i32x4v untracked_load16(const void* sbase, unsigned voff) {
  i32x4v v;
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");   // s_nop: see untracked_dma16
  return v;
}

__device__ __forceinline__ void untracked_load8(const void* sbase, unsigned voff, unsigned& lo, unsigned& hi) {
  typedef int i32x2v __attribute__((ext_vector_type(2)));
  i32x2v v;
  asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  lo = (unsigned)v[0]; hi = (unsigned)v[1];
}

// lens[b] as a SCALAR load (the compiler picks a vector load for a pointer the kernel may also write through, tracks it,
// and waits vmcnt(0) for it -- draining the LDS-DMA in flight); the address must be wave-uniform.
__device__ __forceinline__ int scalar_load_i32(const int* ptr) {
  int v;
  asm volatile("s_nop 4\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");   // s_nop: the address may
                                                        // come from v_readfirstlane (VALU-written SGPR read by a memory instruction)
  return v;
}

// Tiles a persistent workgroup of the fused backward kernels takes at least (SMT_FUSED_MIN_TPW): every workgroup leaves a
// partial weight-gradient slab that the reduce kernel reads back, so at the small levels of the model fewer, longer
// workgroups cost less than the slabs of 256 short ones.
inline int fused_min_tpw() {
  static const int v = [] { const char* e = getenv("SMT_FUSED_MIN_TPW"); return e ? std::max(1, atoi(e)) : 2; }();
  return v;
}

// Fixed-order reduction of weight-gradient partial slabs (conv_wgrad.hip): slab[chunk][blk = co/64 * nblk_ci + ci/cib]
// [plane = tap | bias][64 co][cib ci] -> dw[co*so + ci*si + jmap[tap]*sj], db[co] (column 0 of the bias plane).
int launch_wgrad_reduce(const float* slab, float* dw, float* db, int n_chunks, int nblk_co, int nblk_ci, int taps,
                        int c_in, int c_out, int cib, long long so, long long si, long long sj, const int* jmap,
                        hipStream_t stream, int bias_cols = 1,    // bias_cols > 1: db = sum of columns 0, 32, .. of the bias plane
                        int vsplit = 0);                          // vsplit > 0: slab column ci = channel ci % vsplit of tap ci / vsplit


}  // namespace smt
