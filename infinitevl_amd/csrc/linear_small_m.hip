// y[M,N] = x[M,K] W[N,K]^T (+ bias), bf16 in / fp32 accumulate / bf16 out, for M <= 4 rows: the projections of a
// single-token decode step (InfiniteVLSelfAttention / GatedDeltaNet / MLP / lm_head at T = 1, demo:399-422 loop).
// At M = 1 a GEMM is a pure weight stream (2 FLOP per weight byte): the bound is HBM, so the kernel is a
// bandwidth kernel, not an MFMA kernel -- every wave owns RPW weight rows, every lane streams 16-byte pieces of
// them (whole 1 KB wavefront loads, non-temporal: each weight byte is used once per step) with 4*RPW loads in
// flight, multiplies in fp32 (bf16 -> fp32 is a shift; v_dot2c_f32_bf16 did not reproduce the fp32 reference on
// gfx950 and is not used) and the 64 partial sums of a row meet in one wave reduction.
//
// GLU variant (the SwiGLU MLP, std:945): W is the fused gate|up weight [2I,K]; a wave owns rows n and I+n and writes
// act[n] = bf16(bf16(silu(bf16(gate_n))) * bf16(up_n)) -- the gate costs one exp per OUTPUT element.  (Forming the
// gate on the input side of down_proj instead repeats it in every wave: measured 2.49 -> 2.88 ms per decode token.)
#include "ivl_common.h"

namespace ivl {

__device__ __forceinline__ float dot8(u32x4 a, u32x4 b, float acc) {
  acc = fmaf(bflo(a.x), bflo(b.x), acc); acc = fmaf(bfhi(a.x), bfhi(b.x), acc);
  acc = fmaf(bflo(a.y), bflo(b.y), acc); acc = fmaf(bfhi(a.y), bfhi(b.y), acc);
  acc = fmaf(bflo(a.z), bflo(b.z), acc); acc = fmaf(bfhi(a.z), bfhi(b.z), acc);
  acc = fmaf(bflo(a.w), bflo(b.w), acc); acc = fmaf(bfhi(a.w), bfhi(b.w), acc);
  return acc;
}

constexpr int LSM_U = 4;          // 16-byte pieces per lane per row in flight

// N = number of OUTPUT columns (GLU: I; the weight then has 2N rows)
template <int RPW, int M, bool GLU>
__global__ __launch_bounds__(256) void linear_small_m_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ bias, bf16_t* __restrict__ y,
                                                            int N, int K) {
  constexpr int NR = GLU ? 2 * RPW : RPW;          // weight rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * RPW;
  if (row0 >= N) return;
  const int nchunk = K >> 3;
  const u32x4* wrow[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int rr = min(row0 + (r % RPW), N - 1) + (r >= RPW ? N : 0);      // GLU: rows [RPW, 2RPW) are the "up" rows
    wrow[r] = (const u32x4*)(w + (size_t)rr * K);
  }
  float acc[NR][M];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int m = 0; m < M; ++m) acc[r][m] = 0.f;

  for (int c0 = lane; c0 < nchunk; c0 += 64 * LSM_U) {
    u32x4 wv[LSM_U][NR], xv[LSM_U][M];
#pragma unroll
    for (int u = 0; u < LSM_U; ++u) {
      const int c = c0 + 64 * u;
      const int cc = min(c, nchunk - 1);
#pragma unroll
      for (int r = 0; r < NR; ++r) wv[u][r] = __builtin_nontemporal_load(wrow[r] + cc);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const u32x4 xl = *((const u32x4*)(x + (size_t)m * K) + cc);
        xv[u][m] = c < nchunk ? xl : u32x4{0u, 0u, 0u, 0u};      // pieces past the row end contribute nothing
      }
    }
#pragma unroll
    for (int u = 0; u < LSM_U; ++u)
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = dot8(wv[u][r], xv[u][m], acc[r][m]);
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float s = wave_sum(acc[r][m]);
      float s2 = 0.f;
      if (GLU) s2 = wave_sum(acc[RPW + r][m]);
      if (lane == 0 && row0 + r < N) {
        if (GLU) {
          const float gt = bf_round(s + (bias != nullptr ? bf2f(bias[row0 + r]) : 0.f));          // gate_proj output
          const float up = bf_round(s2 + (bias != nullptr ? bf2f(bias[N + row0 + r]) : 0.f));     // up_proj output
          y[(size_t)m * N + row0 + r] = f2bf(bf_round(gt * sigmoidf_(gt)) * up);                  // = silu_mul_kernel
        } else {
          y[(size_t)m * N + row0 + r] = f2bf(s + (bias != nullptr ? bf2f(bias[row0 + r]) : 0.f));
        }
      }
    }
}

template <int RPW, bool GLU>
static void launch_lsm(const bf16_t* x, const bf16_t* w, const bf16_t* bias, bf16_t* y, int M, int N, int K, hipStream_t st) {
  const int rows_per_wg = 4 * RPW;
  dim3 grid((N + rows_per_wg - 1) / rows_per_wg);
  switch (M) {
    case 1: hipLaunchKernelGGL((linear_small_m_kernel<RPW, 1, GLU>), grid, dim3(256), 0, st, x, w, bias, y, N, K); break;
    case 2: hipLaunchKernelGGL((linear_small_m_kernel<RPW, 2, GLU>), grid, dim3(256), 0, st, x, w, bias, y, N, K); break;
    case 3: hipLaunchKernelGGL((linear_small_m_kernel<RPW, 3, GLU>), grid, dim3(256), 0, st, x, w, bias, y, N, K); break;
    default: hipLaunchKernelGGL((linear_small_m_kernel<RPW, 4, GLU>), grid, dim3(256), 0, st, x, w, bias, y, N, K); break;
  }
}

static int lsm_dispatch(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, bool glu,
                        void* stream, const char* who) {
  IVL_REQUIRE(x && w && y, IVL_ERR_INVALID_ARG, "%s: NULL pointer", who);
  IVL_REQUIRE(M >= 1 && M <= 4, IVL_ERR_UNSUPPORTED, "%s: M=%d (built for 1..4 rows; use a GEMM)", who, M);
  IVL_REQUIRE(N > 0 && K > 0 && K % 8 == 0, IVL_ERR_INVALID_ARG, "%s: N=%d K=%d (K must be a multiple of 8)", who, N, K);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t *xp = (const bf16_t*)x, *wp = (const bf16_t*)w, *bp = (const bf16_t*)bias;
  bf16_t* yp = (bf16_t*)y;
  // rows per wave: enough workgroups to cover the chip (>= ~2 per CU) before amortising the x reads over more rows
  if (glu) {
    if (N >= 4096) launch_lsm<2, true>(xp, wp, bp, yp, M, N, K, st);       // 4 weight rows per wave
    else launch_lsm<1, true>(xp, wp, bp, yp, M, N, K, st);
  } else {
    if (N >= 8192) launch_lsm<4, false>(xp, wp, bp, yp, M, N, K, st);
    else if (N >= 4096) launch_lsm<2, false>(xp, wp, bp, yp, M, N, K, st);
    else launch_lsm<1, false>(xp, wp, bp, yp, M, N, K, st);
  }
  return check_launch(who);
}

}  // namespace ivl

using namespace ivl;

extern "C" int ivl_linear_small_m_fwd(const void* x, const void* w, const void* bias, void* y, int M, int N, int K,
                                      void* stream) {
  return lsm_dispatch(x, w, bias, y, M, N, K, false, stream, "ivl_linear_small_m_fwd");
}

extern "C" int ivl_linear_swiglu_small_m_fwd(const void* x, const void* w_gate_up, const void* bias, void* y, int M, int I,
                                             int K, void* stream) {
  return lsm_dispatch(x, w_gate_up, bias, y, M, I, K, true, stream, "ivl_linear_swiglu_small_m_fwd");
}
