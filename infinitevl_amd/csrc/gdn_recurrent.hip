// Gated DeltaNet, token-recurrent form (decode / short calls, T <= 64 in the model).
//
// One workgroup = one (batch, head, 32-column slab of V): the fp32 state slab S[128 x 32] lives in
// registers (16 values per thread: thread = column c, row group rg of 16 rows) for the whole call.
// Per token:  S' = e^g S ;  kv = S'^T k ;  oq = S'^T (q*scale)      (one pass over the registers)
//             d  = beta (v - kv) ;  S = S' + k d^T ;  o = oq + d (k . q*scale)
// (o = S^T q*scale expanded so that both column reductions share one LDS round / one barrier.)
// HBM-bound at T == 1 (state slab read + write); q/k/v/g/beta of a 32-token block are staged in LDS
// by a prologue so that the serial token loop never waits on global memory.
#include "ivl_common.h"

namespace ivl {

constexpr int REC_BV = 32;   // state columns per workgroup
constexpr int REC_TB = 32;   // tokens staged per block
constexpr int REC_K = 128;

// F16: the activations (q, k, v, beta in; o out; the l2norm's rounding point) are IEEE half instead of bf16 -- fla's operators take
// either (chunk.py:352 refuses fp32 only); everything else (fp32 arithmetic, the state's own dtype) is the same code.
template <bool F16> struct Act {
  static __device__ __forceinline__ float lo(unsigned int w) { return F16 ? h2f_bits((unsigned short)(w & 0xffffu)) : bflo(w); }
  static __device__ __forceinline__ float hi(unsigned int w) { return F16 ? h2f_bits((unsigned short)(w >> 16)) : bfhi(w); }
  static __device__ __forceinline__ float from(bf16_t x) { return F16 ? h2f_bits(x) : bf2f(x); }
  static __device__ __forceinline__ bf16_t to(float f) {
    if constexpr (F16) { const _Float16 hh = (_Float16)f; return __builtin_bit_cast(unsigned short, hh); }
    else return f2bf(f);
  }
  static __device__ __forceinline__ float round(float f) { return from(to(f)); }
};
template <bool F16>
__global__ __launch_bounds__(256) void gdn_recurrent_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const float* __restrict__ g, const bf16_t* __restrict__ beta, bf16_t* __restrict__ o,
    const void* h0, int h0_dtype, void* ht, int ht_dtype,
    int T, int H, int V, float scale, int l2norm) {
  __shared__ __attribute__((aligned(16))) float s_q[REC_TB][REC_K];   // q_hat * scale
  __shared__ __attribute__((aligned(16))) float s_k[REC_TB][REC_K];   // k_hat
  __shared__ float s_v[REC_TB][REC_BV];
  __shared__ float s_eg[REC_TB], s_beta[REC_TB], s_kq[REC_TB];
  __shared__ float s_red[2][4][2][REC_BV];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c = tid & 31, rg = tid >> 5;
  const int row0 = rg * 16;
  const int v0 = blockIdx.x * REC_BV;
  const int bh = blockIdx.y;
  const int b = bh / H, h = bh % H;

  float S[16];
  {
    const size_t base = ((size_t)bh * REC_K + row0) * V + v0 + c;
    if (h0 != nullptr && h0_dtype == IVL_F32) {          // dtype branch hoisted: 16 loads in flight
      const float* hp = (const float*)h0 + base;
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = hp[(size_t)r * V];
    } else if (h0 != nullptr) {
      const bf16_t* hp = (const bf16_t*)h0 + base;
      bf16_t raw[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) raw[r] = hp[(size_t)r * V];
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = bf2f(raw[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = 0.f;
    }
  }

  int par = 0;
  for (int t0 = 0; t0 < T; t0 += REC_TB) {
    const int nb = min(REC_TB, T - t0);
    __syncthreads();   // previous block's readers of s_* are done
    // ---- prologue: stage + normalise this token block --------------------------------------
    for (int tt = wave; tt < nb; tt += 4) {
      const size_t tok = ((size_t)b * T + (t0 + tt)) * H + h;
      const unsigned int qw = *(const unsigned int*)(q + tok * REC_K + 2 * lane);
      const unsigned int kw = *(const unsigned int*)(k + tok * REC_K + 2 * lane);
      float q0 = Act<F16>::lo(qw), q1 = Act<F16>::hi(qw), k0 = Act<F16>::lo(kw), k1 = Act<F16>::hi(kw);
      if (l2norm) {
        const float qs = wave_sum(q0 * q0 + q1 * q1);
        const float ks = wave_sum(k0 * k0 + k1 * k1);
        const float rq = 1.0f / sqrtf(qs + 1e-6f), rk = 1.0f / sqrtf(ks + 1e-6f);
        q0 = Act<F16>::round(q0 * rq); q1 = Act<F16>::round(q1 * rq);     // fla l2norm_fwd writes the activation dtype
        k0 = Act<F16>::round(k0 * rk); k1 = Act<F16>::round(k1 * rk);
      }
      q0 *= scale; q1 *= scale;
      const float kq = wave_sum(k0 * q0 + k1 * q1);
      s_q[tt][2 * lane] = q0; s_q[tt][2 * lane + 1] = q1;
      s_k[tt][2 * lane] = k0; s_k[tt][2 * lane + 1] = k1;
      if (lane < REC_BV) s_v[tt][lane] = Act<F16>::from(v[tok * V + v0 + lane]);
      if (lane == 0) {
        s_eg[tt] = __expf(g[tok]);
        s_beta[tt] = Act<F16>::from(beta[tok]);
        s_kq[tt] = kq;
      }
    }
    __syncthreads();
    // ---- serial token loop ------------------------------------------------------------------
    for (int tt = 0; tt < nb; ++tt) {
      const float decay = s_eg[tt];
      float pk = 0.f, pq = 0.f;
      float kk[16];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 kv4 = *(const f32x4*)&s_k[tt][row0 + 4 * r4];
        const f32x4 qv4 = *(const f32x4*)&s_q[tt][row0 + 4 * r4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * r4 + i;
          kk[r] = kv4[i];
          S[r] *= decay;
          pk = fmaf(S[r], kv4[i], pk);
          pq = fmaf(S[r], qv4[i], pq);
        }
      }
      pk += __shfl_xor(pk, 32, 64);
      pq += __shfl_xor(pq, 32, 64);
      if (lane < 32) {
        s_red[par][wave][0][c] = pk;
        s_red[par][wave][1][c] = pq;
      }
      __syncthreads();
      const float kv = s_red[par][0][0][c] + s_red[par][1][0][c] + s_red[par][2][0][c] + s_red[par][3][0][c];
      const float oq = s_red[par][0][1][c] + s_red[par][1][1][c] + s_red[par][2][1][c] + s_red[par][3][1][c];
      const float delta = s_beta[tt] * (s_v[tt][c] - kv);
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = fmaf(kk[r], delta, S[r]);
      if (rg == 0) {
        const size_t tok = ((size_t)b * T + (t0 + tt)) * H + h;
        o[tok * V + v0 + c] = Act<F16>::to(fmaf(delta, s_kq[tt], oq));
      }
      par ^= 1;
    }
  }
  if (ht != nullptr) {
    const size_t base = ((size_t)bh * REC_K + row0) * V + v0 + c;
    if (ht_dtype == IVL_F32) {
      float* hp = (float*)ht + base;
#pragma unroll
      for (int r = 0; r < 16; ++r) hp[(size_t)r * V] = S[r];
    } else {
      bf16_t* hp = (bf16_t*)ht + base;
#pragma unroll
      for (int r = 0; r < 16; ++r) hp[(size_t)r * V] = f2bf(S[r]);
    }
  }
}

}  // namespace ivl

using namespace ivl;

template <bool F16>
static int gdn_recurrent_launch(const char* who, const void* q, const void* k, const void* v, const float* g, const void* beta,
                                void* o, const void* h0, int h0_dtype, void* ht, int ht_dtype,
                                int B, int T, int H, int K, int V, float scale, int use_qk_l2norm, void* stream);

extern "C" int ivl_gdn_recurrent_fwd(const void* q, const void* k, const void* v, const float* g, const void* beta,
                                     void* o, const void* h0, int h0_dtype, void* ht, int ht_dtype,
                                     int B, int T, int H, int K, int V, float scale, int use_qk_l2norm, void* stream) {
  return gdn_recurrent_launch<false>("ivl_gdn_recurrent_fwd", q, k, v, g, beta, o, h0, h0_dtype, ht, ht_dtype, B, T, H, K, V, scale, use_qk_l2norm, stream);
}
extern "C" int ivl_gdn_recurrent_f16_fwd(const void* q, const void* k, const void* v, const float* g, const void* beta,
                                         void* o, const void* h0, int h0_dtype, void* ht, int ht_dtype,
                                         int B, int T, int H, int K, int V, float scale, int use_qk_l2norm, void* stream) {
  return gdn_recurrent_launch<true>("ivl_gdn_recurrent_f16_fwd", q, k, v, g, beta, o, h0, h0_dtype, ht, ht_dtype, B, T, H, K, V, scale, use_qk_l2norm, stream);
}

template <bool F16>
static int gdn_recurrent_launch(const char* who, const void* q, const void* k, const void* v, const float* g, const void* beta,
                                void* o, const void* h0, int h0_dtype, void* ht, int ht_dtype,
                                int B, int T, int H, int K, int V, float scale, int use_qk_l2norm, void* stream) {
  IVL_REQUIRE(q && k && v && g && beta && o, IVL_ERR_INVALID_ARG, "%s: NULL pointer", who);
  IVL_REQUIRE(B > 0 && T > 0 && H > 0, IVL_ERR_INVALID_ARG, "%s: B,T,H must be positive (%d,%d,%d)", who, B, T, H);
  IVL_REQUIRE(K == REC_K, IVL_ERR_UNSUPPORTED, "%s: K=%d unsupported (built for 128)", who, K);
  IVL_REQUIRE(V > 0 && V % REC_BV == 0, IVL_ERR_UNSUPPORTED, "%s: V=%d must be a multiple of %d", who, V, REC_BV);
  IVL_REQUIRE((h0 == nullptr || h0_dtype == IVL_F32 || h0_dtype == IVL_BF16) &&
              (ht == nullptr || ht_dtype == IVL_F32 || ht_dtype == IVL_BF16),
              IVL_ERR_INVALID_ARG, "%s: state dtype must be IVL_F32 or IVL_BF16", who);
  hipLaunchKernelGGL(gdn_recurrent_kernel<F16>, dim3(V / REC_BV, B * H), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, g, (const bf16_t*)beta, (bf16_t*)o,
                     h0, h0_dtype, ht, ht_dtype, T, H, V, scale, use_qk_l2norm);
  return check_launch(who);
}
