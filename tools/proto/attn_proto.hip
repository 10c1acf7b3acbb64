// PROTOTYPE (developer experiment, not part of libivl_hip.so): compute-loop ceiling of a one-wave-per-SIMD attention tile loop.
// Workgroup = 4 waves x 64 query rows (two 32-row q-blocks per wave); a 64-key K / V tile sits in LDS and is re-used for every
// iteration (no loaders: this measures what the MFMA / softmax pipeline of ONE wave per SIMD reaches); the two 32-key halves of
// a tile are software-pipelined inside the wave: QK(h0) | QK(h1) || softmax(h0) | PV(h0) || softmax(h1) | PV(h1).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __bf16 bf16_native2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) { const bf16_native2 v = {(__bf16)lo, (__bf16)hi}; return __builtin_bit_cast(unsigned int, v); }
__device__ __forceinline__ mfma_bf16x8 mf(u32x4 v) { mfma_bf16x8 r; __builtin_memcpy(&r, &v, 16); return r; }
constexpr int KS = 272, VS = 320, KT = 64, D = 128;
constexpr int K_BYTES = KT * KS, V_OFF = K_BYTES, Q_OFF = K_BYTES + KT * VS, LDS = Q_OFF + 256 * KS;

template <int PIPE, bool QLDS>
__global__ __launch_bounds__(256, 1) void attn_proto(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                    bf16_t* __restrict__ o, int ntiles, float sc) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi5 = lane >> 5, l15 = lane & 15;
  // stage the one K / V tile (rows 0..63 of k, v)
  for (int i = tid; i < KT * 16; i += 256) {
    const int r = i >> 4, c = i & 15;
    *(u32x4*)(smem + r * KS + c * 16) = *(const u32x4*)(k + r * D + c * 8);
    *(u32x4*)(smem + V_OFF + r * VS + c * 16) = *(const u32x4*)(v + r * D + c * 8);
  }
  const int row0 = (blockIdx.x * 4 + wave) * 64;
  u32x4 qf[2][8];
  if constexpr (QLDS) {
    for (int i = tid; i < 256 * 16; i += 256) {
      const int r = i >> 4, c = i & 15;
      *(u32x4*)(smem + Q_OFF + r * KS + c * 16) = *(const u32x4*)(q + (size_t)(blockIdx.x * 256 + r) * D + c * 8);
    }
  } else {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int kd = 0; kd < 8; ++kd) qf[qb][kd] = *(const u32x4*)(q + (size_t)(row0 + 32 * qb + l31) * D + 16 * kd + 8 * hi5);
  }
  const int q_off = Q_OFF + (wave * 64 + l31) * KS + 16 * hi5;
  f32x16 oacc[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][mt][r] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  __syncthreads();
  const int k_off = l31 * KS + 16 * hi5;
  const int v_off = V_OFF + (4 * hi5 + (l15 >> 2)) * VS + (16 * ((lane >> 4) & 1) + 4 * (l15 & 3)) * 2;

  auto qk_half = [&](int kh, f32x16 (&s)[2]) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[qb][r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) {
      const u32x4 fr = *(const u32x4*)(smem + 32 * kh * KS + k_off + 32 * kd);
      const u32x4 q0 = QLDS ? *(const u32x4*)(smem + q_off + 32 * kd) : qf[0][kd];
      const u32x4 q1 = QLDS ? *(const u32x4*)(smem + q_off + 32 * KS + 32 * kd) : qf[1][kd];
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fr), mf(q0), s[0], 0, 0, 0);
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fr), mf(q1), s[1], 0, 0, 0);
    }
  };
  auto softmax_half = [&](f32x16 (&s)[2], u32x4 (&pf)[2][2]) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float rmax = fmaxf(s[qb][0], s[qb][1]);
#pragma unroll
      for (int r = 2; r < 16; ++r) rmax = fmaxf(rmax, s[qb][r]);
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
      rmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * sc;
      const float m_new = fmaxf(m_run[qb], rmax);
      if (__any(m_new > m_run[qb] + 8.0f)) {
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) oacc[qb][mt] *= alpha;
        l_run[qb] *= alpha;
        m_run[qb] = m_new;
      }
      const float mu = m_run[qb];
      float rsum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][r], sc, -mu));
        s[qb][r] = p;
        rsum += p;
      }
      l_run[qb] += rsum;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        pf[qb][ks] = u32x4{pack2bf(s[qb][8 * ks + 0], s[qb][8 * ks + 1]), pack2bf(s[qb][8 * ks + 2], s[qb][8 * ks + 3]),
                           pack2bf(s[qb][8 * ks + 4], s[qb][8 * ks + 5]), pack2bf(s[qb][8 * ks + 6], s[qb][8 * ks + 7])};
    }
  };
  auto pv_half = [&](int kh, const u32x4 (&pf)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const unsigned char* vp = smem + v_off + (32 * kh + 16 * ks) * VS + 64 * mt;
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)vp);
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp + 8 * VS));
        u32x2 w0, w1;
        __builtin_memcpy(&w0, &a0, 8);
        __builtin_memcpy(&w1, &a1, 8);
        const u32x4 fv = u32x4{w0.x, w0.y, w1.x, w1.y};
        oacc[0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fv), mf(pf[0][ks]), oacc[0][mt], 0, 0, 0);
        oacc[1][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fv), mf(pf[1][ks]), oacc[1][mt], 0, 0, 0);
      }
  };
  // --- pieces of the hand-interleaved schedule (PIPE == 2): the row-maximum / rescale step (may branch) is separate from the
  //     exponentials + packing (straight-line VALU) so that the latter can be interleaved with the other half's MFMAs
  auto max_half = [&](f32x16 (&s)[2], float (&mu)[2]) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float rmax = fmaxf(s[qb][0], s[qb][1]);
#pragma unroll
      for (int r = 2; r < 16; ++r) rmax = fmaxf(rmax, s[qb][r]);
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
      rmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * sc;
      const float m_new = fmaxf(m_run[qb], rmax);
      if (__any(m_new > m_run[qb] + 8.0f)) {
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) oacc[qb][mt] *= alpha;
        l_run[qb] *= alpha;
        m_run[qb] = m_new;
      }
      mu[qb] = m_run[qb];
    }
  };
  auto exp_half = [&](f32x16 (&s)[2], const float (&mu)[2], u32x4 (&pf)[2][2]) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float rsum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][r], sc, -mu[qb]));
        s[qb][r] = p;
        rsum += p;
      }
      l_run[qb] += rsum;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        pf[qb][ks] = u32x4{pack2bf(s[qb][8 * ks + 0], s[qb][8 * ks + 1]), pack2bf(s[qb][8 * ks + 2], s[qb][8 * ks + 3]),
                           pack2bf(s[qb][8 * ks + 4], s[qb][8 * ks + 5]), pack2bf(s[qb][8 * ks + 6], s[qb][8 * ks + 7])};
    }
  };
  auto interleave = [&]() {              // 16 x { 1 MFMA, 2 LDS reads, 7 VALU }: the scheduler pipelines the block this way
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
    }
  };
  f32x16 sA[2], sB[2];
  u32x4 pA[2][2], pB[2][2];
  float muA[2], muB[2];
  for (int t = 0; t < ntiles; ++t) {
    if (PIPE == 2) {
      qk_half(0, sA);
      max_half(sA, muA);
      __builtin_amdgcn_sched_barrier(0);
      qk_half(1, sB);
      exp_half(sA, muA, pA);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      max_half(sB, muB);
      __builtin_amdgcn_sched_barrier(0);
      pv_half(0, pA);
      exp_half(sB, muB, pB);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      pv_half(1, pB);
    } else if (PIPE == 0) {             // plain order
      qk_half(0, sA); softmax_half(sA, pA); pv_half(0, pA);
      qk_half(1, sB); softmax_half(sB, pB); pv_half(1, pB);
    } else {                            // source order of the pipelined schedule: the compiler may overlap neighbours
      qk_half(0, sA);
      qk_half(1, sB);
      softmax_half(sA, pA);
      pv_half(0, pA);
      softmax_half(sB, pB);
      pv_half(1, pB);
    }
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[qb]), __float_as_uint(l_run[qb]), false, false);
    const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    const float inv = 1.0f / l;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16_t* op = o + (size_t)(row0 + 32 * qb + l31) * D + 32 * mt + 8 * qd + 4 * hi5;
        *(u32x2*)op = u32x2{pack2bf(oacc[qb][mt][4 * qd] * inv, oacc[qb][mt][4 * qd + 1] * inv),
                            pack2bf(oacc[qb][mt][4 * qd + 2] * inv, oacc[qb][mt][4 * qd + 3] * inv)};
      }
  }
}
extern "C" int attn_proto_launch(const void* q, const void* k, const void* v, void* o, int rows, int ntiles, float sc, int pipe, void* stream) {
  dim3 grid(rows / 256), block(256);
#define GO(P, Q) { (void)hipFuncSetAttribute((const void*)attn_proto<P, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, 0); \
    hipLaunchKernelGGL((attn_proto<P, Q>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ntiles, sc); }
  if (pipe == 0) GO(0, false) else if (pipe == 1) GO(1, false) else if (pipe == 2) GO(0, true) else if (pipe == 3) GO(1, true) else GO(2, true)
  return (int)hipGetLastError();
}
