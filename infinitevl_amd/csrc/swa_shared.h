// Helpers shared by the sliding-window attention kernels (swa.hip, swa_ring256.hip): head size, ring-slot arithmetic, the
// single-instruction max helpers and the M-RoPE arithmetic (bit-identical to ivl_mrope_fwd).
#pragma once
#include "ivl_common.h"

namespace ivl {

typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int SWA_D = 128;
constexpr float LOG2E = 1.4426950408889634f;

// x mod C for a token position x >= 0 and a ring capacity C > 0: every workgroup computes the ring slot of its first key
// from the device-resident position before it can request a tile, and a 64-bit remainder by a run-time divisor is ~150
// instructions; positions below 2^32 (4 G tokens: the usual case, wave-uniform branch) take the 32-bit expansion.
__device__ __forceinline__ int mod_pos(long long x, int C) {
  if ((unsigned long long)x < 0x100000000ull) return (int)((unsigned int)x % (unsigned int)C);
  return (int)(x % C);
}

__device__ __forceinline__ mfma_bf16x8 as_mfma(u32x4 v) {
  mfma_bf16x8 r;
  __builtin_memcpy(&r, &v, 16);
  return r;
}

// single-instruction max helpers (a plain fmaxf on MFMA results draws a canonicalising v_max per operand)
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// M-RoPE (std:949-984) on one pair of 8-channel groups of one row: channels c0..c0+7 ("lo") and c0+64..c0+71 ("hi").
// cos/sin: [3, B, T, 128] bf16 tables (t, h, w); the channel block selects its table by the mrope sections (s0 | s1 | rest,
// multiples of 8).  Products and the sum are each rounded to bf16 like the reference's eager bf16 arithmetic: bit-identical
// to ivl_mrope_fwd.  `row_off` = (b * T + t) * 128, `plane` = B * T * 128.
// the arithmetic of rope_pair on tables already in registers
__device__ __forceinline__ void rope_apply(u32x4& lo, u32x4& hi, const u32x4 c1, const u32x4 n1, const u32x4 c2, const u32x4 n2) {
  const unsigned int x1[4] = {lo.x, lo.y, lo.z, lo.w}, x2[4] = {hi.x, hi.y, hi.z, hi.w};
  const unsigned int cc1[4] = {c1.x, c1.y, c1.z, c1.w}, nn1[4] = {n1.x, n1.y, n1.z, n1.w};
  const unsigned int cc2[4] = {c2.x, c2.y, c2.z, c2.w}, nn2[4] = {n2.x, n2.y, n2.z, n2.w};
  unsigned int o1[4], o2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a0 = bflo(x1[i]), a1 = bfhi(x1[i]), b0 = bflo(x2[i]), b1 = bfhi(x2[i]);
    o1[i] = pack2bf(bf_round(a0 * bflo(cc1[i])) + bf_round(-b0 * bflo(nn1[i])), bf_round(a1 * bfhi(cc1[i])) + bf_round(-b1 * bfhi(nn1[i])));
    o2[i] = pack2bf(bf_round(b0 * bflo(cc2[i])) + bf_round(a0 * bflo(nn2[i])), bf_round(b1 * bfhi(cc2[i])) + bf_round(a1 * bfhi(nn2[i])));
  }
  lo = u32x4{o1[0], o1[1], o1[2], o1[3]};
  hi = u32x4{o2[0], o2[1], o2[2], o2[3]};
}
__device__ __forceinline__ void rope_pair(u32x4& lo, u32x4& hi, const bf16_t* cosp, const bf16_t* sinp, long long plane,
                                          long long row_off, int c0, int s0, int s1) {
  const int sec = c0 < s0 ? 0 : (c0 < s0 + s1 ? 1 : 2);
  const long long off = sec * plane + row_off + c0;
  const u32x4 c1 = *(const u32x4*)(cosp + off), n1 = *(const u32x4*)(sinp + off);
  const u32x4 c2 = *(const u32x4*)(cosp + off + 64), n2 = *(const u32x4*)(sinp + off + 64);
  const unsigned int x1[4] = {lo.x, lo.y, lo.z, lo.w}, x2[4] = {hi.x, hi.y, hi.z, hi.w};
  const unsigned int cc1[4] = {c1.x, c1.y, c1.z, c1.w}, nn1[4] = {n1.x, n1.y, n1.z, n1.w};
  const unsigned int cc2[4] = {c2.x, c2.y, c2.z, c2.w}, nn2[4] = {n2.x, n2.y, n2.z, n2.w};
  unsigned int o1[4], o2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a0 = bflo(x1[i]), a1 = bfhi(x1[i]), b0 = bflo(x2[i]), b1 = bfhi(x2[i]);
    o1[i] = pack2bf(bf_round(a0 * bflo(cc1[i])) + bf_round(-b0 * bflo(nn1[i])), bf_round(a1 * bfhi(cc1[i])) + bf_round(-b1 * bfhi(nn1[i])));
    o2[i] = pack2bf(bf_round(b0 * bflo(cc2[i])) + bf_round(a0 * bflo(nn2[i])), bf_round(b1 * bfhi(cc2[i])) + bf_round(a1 * bfhi(nn2[i])));
  }
  lo = u32x4{o1[0], o1[1], o1[2], o1[3]};
  hi = u32x4{o2[0], o2[1], o2[2], o2[3]};
}

}  // namespace ivl
