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
//
// NORM variant (round 3): the RMSNorm in front of the projection -- (residual add +) Qwen2RMSNorm, 73 one-row launches of
// ivl_add_rmsnorm_fwd per decode token, each ~4.2 us inside the token's graph: 15 % of the token -- runs in the prologue of
// the weight stream.  Every workgroup forms the normalised rows itself (M x K <= 4 x 4096 elements: 8 KB of L2-resident
// input per row, the same thread -> element mapping, summation order and rounding points as add_rmsnorm_kernel: the rows
// are bit-identical to that kernel's) into LDS while its first weight pieces are in flight, and reads x from there;
// workgroup 0 also writes the new residual stream h = bf16(x + residual).
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

struct NormArgs {
  const bf16_t* residual;          // NULL: plain norm of x
  const bf16_t* weight;            // RMSNorm weight [K]
  bf16_t* h_out;                   // new residual stream (written by workgroup 0 when residual != NULL)
  float eps;
};
constexpr int LSM_NORM_KMAX = 4096;

// N = number of OUTPUT columns (GLU: I; the weight then has 2N rows)
// NIT: 16-byte pieces of a row per thread in the norm prologue (1: K <= 2048 -- the hidden size of the model; 2: K <= 4096);
// the prologue's registers are live while the first weight batch is in flight, and 218 VGPRs meant 2 workgroups per CU: the M = 1
// instances (the decode token) are capped at 168 (three workgroups per CU); M >= 2 would spill there and keeps the default
template <int RPW, int M, bool GLU, bool NORM, int NIT = 2>
__global__ __launch_bounds__(256, (NORM && M == 1) ? 3 : 1) void linear_small_m_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ bias, bf16_t* __restrict__ y,
                                                            int N, int K, NormArgs na) {
  constexpr int NR = GLU ? 2 * RPW : RPW;          // weight rows per wave
  __shared__ __attribute__((aligned(16))) u32x4 s_x[NORM ? M * (LSM_NORM_KMAX / 8) : 1];
  __shared__ float s_part[NORM ? M : 1][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = K >> 3;
  const int row0 = (blockIdx.x * 4 + wave) * RPW;
  if (!NORM && row0 >= N) return;                  // (NORM: every wave takes part in the barriers; its rows are clamped, its stores guarded)
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

  // NORM: the rows' pieces (x, residual, norm weight) are requested FIRST -- vector-memory results return in issue order, so
  // requested behind the weight pieces they would only arrive behind the whole weight stream of the wave -- then the weight
  // pieces; norm_rows (every wave: workgroup barriers inside) then runs while the weights are in flight.
  const int tid = threadIdx.x;
  u32x4 xraw[NORM ? M : 1][NIT], rraw[NORM ? M : 1][NIT], wraw[NIT];
  if constexpr (NORM) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int v = min(tid + it * 256, nchunk - 1);
      wraw[it] = *(const u32x4*)(na.weight + v * 8);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        xraw[m][it] = *(const u32x4*)(x + (size_t)m * K + v * 8);
        rraw[m][it] = na.residual != nullptr ? *(const u32x4*)(na.residual + (size_t)m * K + v * 8) : u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  auto norm_rows = [&]() {
    float hv[M][NIT][8];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float ss = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int v = tid + it * 256;
        if (v < nchunk) {
          unpack8(xraw[m][it], hv[m][it]);
          if (na.residual != nullptr) {
            float rv[8];
            unpack8(rraw[m][it], rv);
#pragma unroll
            for (int i = 0; i < 8; ++i) hv[m][it][i] = bf_round(hv[m][it][i] + rv[i]);
            if (blockIdx.x == 0) *(u32x4*)(na.h_out + (size_t)m * K + v * 8) = pack8(hv[m][it]);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) ss = fmaf(hv[m][it][i], hv[m][it][i], ss);
        }
      }
      ss = wave_sum(ss);
      if (lane == 0) s_part[m][wave] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const float tot = s_part[m][0] + s_part[m][1] + s_part[m][2] + s_part[m][3];
      const float rstd = rsqrtf(tot / (float)K + na.eps);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int v = tid + it * 256;
        if (v < nchunk) {
          float wv8[8], o8[8];
          unpack8(wraw[it], wv8);
#pragma unroll
          for (int i = 0; i < 8; ++i) o8[i] = wv8[i] * bf_round(hv[m][it][i] * rstd);
          s_x[m * (LSM_NORM_KMAX / 8) + v] = pack8(o8);
        }
      }
    }
    __syncthreads();
  };
  bool first = true;
  for (int c0 = lane; c0 < nchunk; c0 += 64 * LSM_U) {
    u32x4 wv[LSM_U][NR], xv[LSM_U][M];
#pragma unroll
    for (int u = 0; u < LSM_U; ++u) {
      const int cc = min(c0 + 64 * u, nchunk - 1);
#pragma unroll
      for (int r = 0; r < NR; ++r) wv[u][r] = __builtin_nontemporal_load(wrow[r] + cc);
    }
    if constexpr (NORM) {
      if (first) norm_rows();                      // (the loop trip count is workgroup-uniform)
      first = false;
    }
#pragma unroll
    for (int u = 0; u < LSM_U; ++u) {
      const int c = c0 + 64 * u;
      const int cc = min(c, nchunk - 1);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        u32x4 xl;
        if constexpr (NORM) xl = s_x[m * (LSM_NORM_KMAX / 8) + cc];
        else xl = *((const u32x4*)(x + (size_t)m * K) + cc);
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

template <int RPW, bool GLU, bool NORM>
static void launch_lsm(const bf16_t* x, const bf16_t* w, const bf16_t* bias, bf16_t* y, int M, int N, int K, const NormArgs& na,
                       hipStream_t st) {
  const int rows_per_wg = 4 * RPW;
  dim3 grid((N + rows_per_wg - 1) / rows_per_wg);
  if (NORM && K <= 2048) {
    switch (M) {
      case 1: hipLaunchKernelGGL((linear_small_m_kernel<RPW, 1, GLU, NORM, 1>), grid, dim3(256), 0, st, x, w, bias, y, N, K, na); break;
      case 2: hipLaunchKernelGGL((linear_small_m_kernel<RPW, 2, GLU, NORM, 1>), grid, dim3(256), 0, st, x, w, bias, y, N, K, na); break;
      case 3: hipLaunchKernelGGL((linear_small_m_kernel<RPW, 3, GLU, NORM, 1>), grid, dim3(256), 0, st, x, w, bias, y, N, K, na); break;
      default: hipLaunchKernelGGL((linear_small_m_kernel<RPW, 4, GLU, NORM, 1>), grid, dim3(256), 0, st, x, w, bias, y, N, K, na); break;
    }
    return;
  }
  switch (M) {
    case 1: hipLaunchKernelGGL((linear_small_m_kernel<RPW, 1, GLU, NORM>), grid, dim3(256), 0, st, x, w, bias, y, N, K, na); break;
    case 2: hipLaunchKernelGGL((linear_small_m_kernel<RPW, 2, GLU, NORM>), grid, dim3(256), 0, st, x, w, bias, y, N, K, na); break;
    case 3: hipLaunchKernelGGL((linear_small_m_kernel<RPW, 3, GLU, NORM>), grid, dim3(256), 0, st, x, w, bias, y, N, K, na); break;
    default: hipLaunchKernelGGL((linear_small_m_kernel<RPW, 4, GLU, NORM>), grid, dim3(256), 0, st, x, w, bias, y, N, K, na); break;
  }
}

template <bool NORM>
static int lsm_dispatch(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, bool glu, const NormArgs& na,
                        void* stream, const char* who) {
  IVL_REQUIRE(x && w && y, IVL_ERR_INVALID_ARG, "%s: NULL pointer", who);
  IVL_REQUIRE(M >= 1 && M <= 4, IVL_ERR_UNSUPPORTED, "%s: M=%d (built for 1..4 rows; use a GEMM)", who, M);
  IVL_REQUIRE(N > 0 && K > 0 && K % 8 == 0, IVL_ERR_INVALID_ARG, "%s: N=%d K=%d (K must be a multiple of 8)", who, N, K);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t *xp = (const bf16_t*)x, *wp = (const bf16_t*)w, *bp = (const bf16_t*)bias;
  bf16_t* yp = (bf16_t*)y;
  // rows per wave: enough workgroups to cover the chip (>= ~2 per CU) before amortising the x reads over more rows
  if (glu) {
    if (N >= 4096) launch_lsm<2, true, NORM>(xp, wp, bp, yp, M, N, K, na, st);       // 4 weight rows per wave
    else launch_lsm<1, true, NORM>(xp, wp, bp, yp, M, N, K, na, st);
  } else {
    if (N >= 8192) launch_lsm<4, false, NORM>(xp, wp, bp, yp, M, N, K, na, st);
    else if (N >= 4096) launch_lsm<2, false, NORM>(xp, wp, bp, yp, M, N, K, na, st);
    else launch_lsm<1, false, NORM>(xp, wp, bp, yp, M, N, K, na, st);
  }
  return check_launch(who);
}

// ------------------------------------------------------------------------------------------------------------------------------
// GDN output projection of a decode step with the GATED RMSNorm in its prologue (round 5): x = o_norm(o, gate) is
// formed by every workgroup itself -- per head a 256-wide statistic of the un-normalised bf16 delta-rule output (8 KB per row,
// L2-resident) and the gate columns of the projection row -- with the arithmetic of gdn_decode_step_kernel's last block (lane l of
// a wave holds elements 4l .. 4l + 3 of a head: same sums, same roundings: bit-identical), staged in LDS, then the o_proj weight
// stream reads x from there.  What makes it worth a kernel: the decode step itself can then run on 4 x as many workgroups
// (gdn_decode_split_kernel) because nothing spans a head any more.  The first workgroups also shift the conv states of q and k
// (new state = taps 1..3 + the step's raw projection value): every quarter workgroup of the split step READS those states, so
// they can only be written by the NEXT launch.
struct GdnOutArgs {
  const bf16_t* gate; long long gate_ld;           // gate[m][h * 256 + c] = gate[m * gate_ld + h * 256 + c]
  const bf16_t* norm_w; float eps; int H;
  const bf16_t* proj; long long proj_ld; int col_q, col_k;
  bf16_t* cq; bf16_t* ck; int Dq;                  // conv states [M, Dq, 4] of q and k
};
template <int M>
__global__ __launch_bounds__(256) void linear_gdn_out_kernel(const bf16_t* __restrict__ o_raw, const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ bias, bf16_t* __restrict__ y, int N, int K, GdnOutArgs ga) {
  __shared__ __attribute__((aligned(16))) u32x4 s_x[M * (LSM_NORM_KMAX / 8)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
  const int nchunk = K >> 3;
  const int row = min((int)blockIdx.x * 4 + wave, N - 1);
  const u32x4* wrow = (const u32x4*)(w + (size_t)row * K);
  // ---- conv-state shift of q and k (a few items per workgroup; independent of everything else here) ----
  {
    const int total = M * 2 * ga.Dq;
    for (int item = (int)blockIdx.x * 256 + tid; item < total; item += (int)gridDim.x * 256) {
      const int ch = item % ga.Dq, rest = item / ga.Dq, isk = rest & 1, m = rest >> 1;
      bf16_t* st = (isk ? ga.ck : ga.cq) + ((size_t)m * ga.Dq + ch) * 4;
      const u32x2 sv = *(const u32x2*)st;
      const bf16_t xr = ga.proj[(size_t)m * ga.proj_ld + (isk ? ga.col_k : ga.col_q) + ch];
      *(u32x2*)st = u32x2{(sv.x >> 16) | (sv.y << 16), (sv.y >> 16) | ((unsigned int)xr << 16)};
    }
  }
  // ---- the rows' pieces first (o, gate, norm weight: 4 heads per wave), then the first weight batch, then the norm ----
  constexpr int HPW = 4;                           // heads per wave: head = wave + 4 i (H <= 16)
  u32x2 oraw[M][HPW], graw[M][HPW];
  const u32x2 wv = *(const u32x2*)(ga.norm_w + 4 * lane);
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    const int hh = min(wave + 4 * i, ga.H - 1);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      oraw[m][i] = *(const u32x2*)(o_raw + (size_t)m * K + hh * 256 + 4 * lane);
      graw[m][i] = *(const u32x2*)(ga.gate + (size_t)m * ga.gate_ld + hh * 256 + 4 * lane);
    }
  }
  float acc[M];
#pragma unroll
  for (int m = 0; m < M; ++m) acc[m] = 0.f;
  bool first = true;
  for (int c0 = lane; c0 < nchunk; c0 += 64 * LSM_U) {
    u32x4 wvv[LSM_U];
#pragma unroll
    for (int u = 0; u < LSM_U; ++u) wvv[u] = __builtin_nontemporal_load(wrow + min(c0 + 64 * u, nchunk - 1));
    if (first) {                                   // (workgroup-uniform trip count)
      const float wf[4] = {bflo(wv.x), bfhi(wv.x), bflo(wv.y), bfhi(wv.y)};
#pragma unroll
      for (int i = 0; i < HPW; ++i) {
        const int hh = wave + 4 * i;
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const float o4[4] = {bflo(oraw[m][i].x), bfhi(oraw[m][i].x), bflo(oraw[m][i].y), bfhi(oraw[m][i].y)};
          const float gf[4] = {bflo(graw[m][i].x), bfhi(graw[m][i].x), bflo(graw[m][i].y), bfhi(graw[m][i].y)};
          const float ss = wave_sum(o4[0] * o4[0] + o4[1] * o4[1] + o4[2] * o4[2] + o4[3] * o4[3]);
          const float rstd = 1.0f / sqrtf(ss * (1.0f / 256.0f) + ga.eps);
          float y4[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) y4[c] = o4[c] * rstd * wf[c] * gf[c] * sigmoidf_(gf[c]);
          if (hh < ga.H)
            *(u32x2*)((bf16_t*)s_x + (size_t)m * LSM_NORM_KMAX + hh * 256 + 4 * lane) = u32x2{pack2bf(y4[0], y4[1]), pack2bf(y4[2], y4[3])};
        }
      }
      __syncthreads();
      first = false;
    }
#pragma unroll
    for (int u = 0; u < LSM_U; ++u) {
      const int c = c0 + 64 * u;
      const int cc = min(c, nchunk - 1);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const u32x4 xl = c < nchunk ? s_x[m * (LSM_NORM_KMAX / 8) + cc] : u32x4{0u, 0u, 0u, 0u};
        acc[m] = dot8(wvv[u], xl, acc[m]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const float sum = wave_sum(acc[m]);
    if (lane == 0 && (int)blockIdx.x * 4 + wave < N)
      y[(size_t)m * N + row] = f2bf(sum + (bias != nullptr ? bf2f(bias[row]) : 0.f));
  }
}

}  // namespace ivl

using namespace ivl;

extern "C" int ivl_linear_small_m_fwd(const void* x, const void* w, const void* bias, void* y, int M, int N, int K,
                                      void* stream) {
  return lsm_dispatch<false>(x, w, bias, y, M, N, K, false, NormArgs{}, stream, "ivl_linear_small_m_fwd");
}

extern "C" int ivl_linear_swiglu_small_m_fwd(const void* x, const void* w_gate_up, const void* bias, void* y, int M, int I,
                                             int K, void* stream) {
  return lsm_dispatch<false>(x, w_gate_up, bias, y, M, I, K, true, NormArgs{}, stream, "ivl_linear_swiglu_small_m_fwd");
}

extern "C" int ivl_norm_linear_small_m_fwd(const void* x, const void* residual, const void* norm_weight, float eps, void* h_out,
                                           const void* w, const void* bias, void* y, int M, int N, int K, int glu, void* stream) {
  IVL_REQUIRE(norm_weight != nullptr, IVL_ERR_INVALID_ARG, "ivl_norm_linear_small_m_fwd: NULL norm weight");
  IVL_REQUIRE(residual == nullptr || h_out != nullptr, IVL_ERR_INVALID_ARG, "ivl_norm_linear_small_m_fwd: residual needs h_out");
  // (K >= 512: every lane of a wave must enter the weight loop, whose first trip holds the workgroup barriers of the norm)
  IVL_REQUIRE(K >= 512 && K <= LSM_NORM_KMAX, IVL_ERR_UNSUPPORTED,
              "ivl_norm_linear_small_m_fwd: K=%d (the normalised rows are staged in LDS by the whole workgroup: 512 <= K <= %d)", K, LSM_NORM_KMAX);
  NormArgs na;
  na.residual = (const bf16_t*)residual; na.weight = (const bf16_t*)norm_weight; na.h_out = (bf16_t*)h_out; na.eps = eps;
  return lsm_dispatch<true>(x, w, bias, y, M, N, K, glu != 0, na, stream, "ivl_norm_linear_small_m_fwd");
}

// o_proj of a GDN decode step with the gated RMSNorm in the prologue and the q / k conv-state shift riding along (see
// linear_gdn_out_kernel; pairs with ivl_gdn_decode_split_fwd).  o_raw bf16 [M, H*256] (un-normalised delta-rule output), gate =
// first gate element of row 0 (row stride gate_ld elements), proj = the step's projection rows (row stride proj_ld), conv states
// [M, Dq, 4] of q and k (updated in place).  y[M, N] = o_norm(o_raw, gate) W^T (+ bias).
extern "C" int ivl_gdn_out_linear_small_m_fwd(const void* o_raw, const void* gate, int64_t gate_ld, const void* norm_weight, float eps,
                                              int H, const void* proj, int64_t proj_ld, int col_q, int col_k, void* conv_state_q,
                                              void* conv_state_k, const void* w, const void* bias, void* y, int M, int N, int K,
                                              void* stream) {
  IVL_REQUIRE(o_raw && gate && norm_weight && proj && conv_state_q && conv_state_k && w && y, IVL_ERR_INVALID_ARG,
              "ivl_gdn_out_linear_small_m_fwd: NULL pointer");
  IVL_REQUIRE(M >= 1 && M <= 4, IVL_ERR_UNSUPPORTED, "ivl_gdn_out_linear_small_m_fwd: M=%d (built for 1..4 rows)", M);
  IVL_REQUIRE(H >= 1 && H <= 16 && K == H * 256 && K <= LSM_NORM_KMAX && N > 0, IVL_ERR_UNSUPPORTED,
              "ivl_gdn_out_linear_small_m_fwd: H=%d K=%d N=%d (K = H * 256 <= %d)", H, K, N, LSM_NORM_KMAX);
  IVL_REQUIRE(gate_ld % 4 == 0 && ((size_t)gate & 7) == 0 && ((size_t)o_raw & 7) == 0, IVL_ERR_INVALID_ARG,
              "ivl_gdn_out_linear_small_m_fwd: o_raw / gate rows must be 8-byte aligned");
  GdnOutArgs ga;
  ga.gate = (const bf16_t*)gate; ga.gate_ld = gate_ld; ga.norm_w = (const bf16_t*)norm_weight; ga.eps = eps; ga.H = H;
  ga.proj = (const bf16_t*)proj; ga.proj_ld = proj_ld; ga.col_q = col_q; ga.col_k = col_k;
  ga.cq = (bf16_t*)conv_state_q; ga.ck = (bf16_t*)conv_state_k; ga.Dq = H * 128;
  const dim3 grid((N + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t *op = (const bf16_t*)o_raw, *wp = (const bf16_t*)w, *bp = (const bf16_t*)bias;
  switch (M) {
    case 1: hipLaunchKernelGGL((linear_gdn_out_kernel<1>), grid, dim3(256), 0, st, op, wp, bp, (bf16_t*)y, N, K, ga); break;
    case 2: hipLaunchKernelGGL((linear_gdn_out_kernel<2>), grid, dim3(256), 0, st, op, wp, bp, (bf16_t*)y, N, K, ga); break;
    case 3: hipLaunchKernelGGL((linear_gdn_out_kernel<3>), grid, dim3(256), 0, st, op, wp, bp, (bf16_t*)y, N, K, ga); break;
    default: hipLaunchKernelGGL((linear_gdn_out_kernel<4>), grid, dim3(256), 0, st, op, wp, bp, (bf16_t*)y, N, K, ga); break;
  }
  return check_launch("ivl_gdn_out_linear_small_m_fwd");
}
