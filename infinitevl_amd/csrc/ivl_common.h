// Shared device/host helpers for the gfx950 kernels of libivl_hip.so.
// Written for CDNA4 only: 64-lane wavefronts, MFMA 16x16x32 / 32x32x16 bf16, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ivl_hip.h"

namespace ivl {

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;     // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// 16-byte store of a kernel OUTPUT that the next kernel (a GEMM on all eight dies) reads.  The lines have to reach the memory
// side before that kernel starts whatever the policy; written through (`sc0 sc1`) they leave while the kernel still runs instead
// of in its end-of-kernel write-back.  IVL_OUT_STORE: 0 = plain, 1 = non-temporal, 2 = write-through (default).  Same-box ABAB
// of the default bench line for the add+norm / gated norm / SwiGLU-gate kernels (136 launches of a step): plain 5.261 / 5.297
// -> write-through 5.230 / 5.266 ms per step (their in-step time 1.431 / 1.425 -> 1.392 / 1.368 ms); non-temporal: no change.
#ifndef IVL_OUT_STORE
#define IVL_OUT_STORE 2
#endif
__device__ __forceinline__ void store_out16(void* p, u32x4 v) {
#if IVL_OUT_STORE == 1
  __builtin_nontemporal_store(v, (u32x4*)p);
#elif IVL_OUT_STORE == 2
  // (the s_nop: a store of more than 8 bytes reads its data registers after issue, and hipcc does not pad an asm statement --
  //  without it the next instruction may overwrite them first: cdna_hip_programming.md section 5.7)
  // CONTRACT for callers: hipcc inserts no wait states in FRONT of an asm statement either -- `v` and `p` must be the
  // results of ordinary VALU instructions (packs, converts, address arithmetic) or of waited loads, never the direct destination
  // of an MFMA / v_dot or a freshly written SGPR-derived VGPR pair; every caller today passes a pack2bf / cvt_pk result or an LDS
  // read behind its s_waitcnt.  A new call site that stores an accumulator register as is must copy it through a VALU move first.
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  *(u32x4*)p = v;
#endif
}
typedef __attribute__((ext_vector_type(2))) unsigned int ivl_u32x2_t;
__device__ __forceinline__ void store_out8(void* p, ivl_u32x2_t v) {
#if IVL_OUT_STORE == 2
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#else
  *(ivl_u32x2_t*)p = v;
#endif
}

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((unsigned int)x) << 16); }

// fp32 -> bf16, round-to-nearest-even, through the native __bf16 type: hipcc lowers the casts to the
// gfx950 hardware conversion (v_cvt_pk_bf16_f32, two elements per instruction).  A hand-rolled integer
// rounding with a NaN test costs ~10 VALU + a branch per element and made whole kernels
// instruction-bound (gdn_chunk_prepare S1: 26k of 64k cycles).
typedef __bf16 bf16_native2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
  const bf16_native2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float bf_round(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ float bflo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

// eight bf16 of a 16-byte piece <-> fp32
__device__ __forceinline__ void unpack8(u32x4 v, float* f) {
  f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
  f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  return u32x4{pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
}

__device__ __forceinline__ float h2f_bits(unsigned short h) {
  _Float16 x;
  __builtin_memcpy(&x, &h, 2);
  return (float)x;
}

// state element load/store in IVL_F32 / IVL_BF16 (wave-uniform dtype)
__device__ __forceinline__ float load_state(const void* p, size_t idx, int dtype) {
  return dtype == IVL_F32 ? ((const float*)p)[idx] : bf2f(((const bf16_t*)p)[idx]);
}
__device__ __forceinline__ void store_state(void* p, size_t idx, int dtype, float v) {
  if (dtype == IVL_F32) ((float*)p)[idx] = v;
  else ((bf16_t*)p)[idx] = f2bf(v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// v_rcp_f32 (1 ulp) instead of an IEEE division (~10 instructions): every SiLU / swish gate of the package goes through
// here (the results are rounded to bf16 right after), so all paths stay bit-identical to each other
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// beta = sigmoid(b) (std:1293) keeps the IEEE division and expf: it is H values per token, and a 1-ulp fp32 difference can
// flip the bf16 rounding against the reference's `b.sigmoid()`
__device__ __forceinline__ float sigmoid_exact_(float x) { return 1.0f / (1.0f + expf(-x)); }

// x / d and x % d for an index x >= 0 and a divisor d > 0: 32-bit expansion while x < 2^32 (the index arithmetic of the
// element-wise kernels around the attention is otherwise two or three 64-bit divisions per thread and iteration)
__device__ __forceinline__ long long divmod_idx(long long x, int d, int& rem) {
  if ((unsigned long long)x < 0x100000000ull) {
    const unsigned int xx = (unsigned int)x, q = xx / (unsigned int)d;
    rem = (int)(xx - q * (unsigned int)d);
    return (long long)q;
  }
  rem = (int)(x % d);
  return x / d;
}

// ---- in-kernel timeline: DEVELOPER build only (`make trace` -> libivl_hip_trace.so, -DIVL_TRACE) ----------
// The product library contains neither the clock reads nor the setter: every macro below compiles to nothing.
// In the trace build a kernel keeps shader-clock readings in registers (IVL_T(name) declares/reads one) and
// block (0,0,0) / thread 0 writes differences into the device buffer ONCE at the end (IVL_TOUT): no memory
// traffic or extra vmcnt/lgkmcnt waits perturb the timed regions.  Each translation unit that traces owns a
// device pointer, set by ivl_debug_set_trace through the per-unit setter.
#ifdef IVL_TRACE
#define IVL_TRACE_DECL(unit)                                                                      \
  static __device__ long long* ivl_trace_buf = nullptr;                                            \
  void trace_set_##unit(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(ivl_trace_buf), &p, sizeof(p)); }
#define IVL_T(name)                                    \
  __builtin_amdgcn_sched_barrier(0);                   \
  const long long name = (long long)__builtin_readcyclecounter(); \
  __builtin_amdgcn_sched_barrier(0)
#define IVL_TVAR(name) long long name = 0
#define IVL_TACC(acc, t1, t0) acc += (t1) - (t0)
#define IVL_TOUT(slot, value)                                                                      \
  do {                                                                                             \
    long long* tb_ = ivl_trace_buf;                                                                \
    if (tb_ != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) \
      tb_[slot] = (long long)(value);                                                              \
  } while (0)
#define IVL_TOUT_AT(tid, slot, value)                                                              \
  do {                                                                                             \
    long long* tb_ = ivl_trace_buf;                                                                \
    if (tb_ != nullptr && threadIdx.x == (tid) && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) \
      tb_[slot] = (long long)(value);                                                              \
  } while (0)
#else
#define IVL_TOUT_AT(tid, slot, value) ((void)0)
#define IVL_TRACE_DECL(unit)
#define IVL_T(name) ((void)0)
#define IVL_TVAR(name) ((void)0)
#define IVL_TACC(acc, t1, t0) ((void)0)
#define IVL_TOUT(slot, value) ((void)0)
#endif

// ---- host side -------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define IVL_REQUIRE(cond, code, ...)        \
  do {                                      \
    if (!(cond)) {                          \
      ivl::set_error(__VA_ARGS__);          \
      return (code);                        \
    }                                       \
  } while (0)

}  // namespace ivl
