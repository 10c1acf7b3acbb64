// Gated DeltaNet mixer core for ONE new token (decode step, std:1215-1347 with q_len == 1), one launch:
//   short conv + SiLU on the q|k|v columns of the fused projection row (conv state shifted in place)
//   -> gate math (g, beta) -> l2norm(q), l2norm(k) -> S = e^g S ; delta rule update ; o = S^T q
//   -> gated RMSNorm with the gate read from the same projection row.
// = ivl_gdn_prologue_fwd + ivl_gdn_recurrent_fwd + ivl_rmsnorm_swish_gate_strided_fwd at T == 1 with the same
// rounding points (conv output, l2norm output, beta, o are rounded to bf16 where those kernels store bf16).
// A decode step is launch-bound (~430 launches per token in the 36-layer stack): this removes two per GDN layer.
//
// One workgroup per (batch, head), 8 waves.  State S[128][256] lives in registers: wave w owns rows 16w..16w+15,
// lane l owns columns 4l..4l+3 (a wave reads/writes one whole 512-byte state row per instruction).
#include "ivl_common.h"

namespace ivl {

constexpr int DK = 128, DV = 256;
constexpr int DNW = 8;                 // waves per workgroup
constexpr int DRW = DK / DNW;          // state rows per wave

struct DecParams {
  const bf16_t* proj; long long ld;
  int col_q, col_k, col_v, col_g, col_a, col_b;
  const bf16_t* wq; const bf16_t* wk; const bf16_t* wv;        // conv taps [D,4]
  bf16_t* cq; bf16_t* ck; bf16_t* cv;                          // conv states [B,D,4], updated in place
  const float* A_log; const float* dt_bias;
  const bf16_t* norm_w; float eps;
  void* state; int state_dtype;                                // [B,H,128,256], updated in place
  bf16_t* y;                                                   // [B, H*256]
  int H; float scale;
};

// one channel of a width-4 causal conv at T == 1: taps over (state[1], state[2], state[3], x); new state = those 4
__device__ __forceinline__ float conv1(const bf16_t* xrow, int col, const bf16_t* w, int wch, bf16_t* st, size_t ch) {
  const u32x2 wv = *(const u32x2*)(w + (size_t)wch * 4);      // taps of channel wch
  const u32x2 sv = *(const u32x2*)(st + ch * 4);              // state of (batch, channel)
  const bf16_t xr = xrow[col];
  const float x = bf2f(xr);
  float a = bflo(wv.x) * bfhi(sv.x);
  a = fmaf(bfhi(wv.x), bflo(sv.y), a);
  a = fmaf(bflo(wv.y), bfhi(sv.y), a);
  a = fmaf(bfhi(wv.y), x, a);
  *(u32x2*)(st + ch * 4) = u32x2{(sv.x >> 16) | (sv.y << 16), (sv.y >> 16) | ((unsigned int)xr << 16)};
  return bf_round(a * sigmoidf_(a));
}

__global__ __launch_bounds__(64 * DNW) void gdn_decode_step_kernel(DecParams p) {
  __shared__ __attribute__((aligned(16))) float s_k[DK], s_q[DK], s_v[DV];
  __shared__ __attribute__((aligned(16))) float s_red[DNW][2][DV];
  __shared__ float s_part[4][2];
  __shared__ float s_sc[4];          // decay, beta, k.q

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  const bf16_t* xrow = p.proj + (long long)b * p.ld;

  // ---- state rows of this wave: issue every load up front --------------------------------------------
  float S[DRW][4];
  const size_t sbase = ((size_t)bh * DK + DRW * wave) * DV + 4 * lane;
  if (p.state_dtype == IVL_F32) {
    const float* sp = (const float*)p.state + sbase;
#pragma unroll
    for (int r = 0; r < DRW; ++r) {
      const f32x4 v4 = *(const f32x4*)(sp + (size_t)r * DV);
      S[r][0] = v4[0]; S[r][1] = v4[1]; S[r][2] = v4[2]; S[r][3] = v4[3];
    }
  } else {
    const bf16_t* sp = (const bf16_t*)p.state + sbase;
    u32x2 raw[DRW];
#pragma unroll
    for (int r = 0; r < DRW; ++r) raw[r] = *(const u32x2*)(sp + (size_t)r * DV);
#pragma unroll
    for (int r = 0; r < DRW; ++r) {
      S[r][0] = bflo(raw[r].x); S[r][1] = bfhi(raw[r].x); S[r][2] = bflo(raw[r].y); S[r][3] = bfhi(raw[r].y);
    }
  }

  // ---- convs: thread t -> q channel t (t < 128) or k channel t-128, and v channel t -------------------
  const int Dq = p.H * DK, Dv = p.H * DV;
  float qk = 0.f;
  if (tid >= 2 * DK) {
    // waves 4..7 only hold state rows; the 256 conv / l2norm lanes are waves 0..3
  } else if (tid < DK) qk = conv1(xrow, p.col_q + h * DK + tid, p.wq, h * DK + tid, p.cq, (size_t)b * Dq + h * DK + tid);
  else qk = conv1(xrow, p.col_k + h * DK + (tid - DK), p.wk, h * DK + tid - DK, p.ck, (size_t)b * Dq + h * DK + (tid - DK));
  if (tid < DV) s_v[tid] = conv1(xrow, p.col_v + h * DV + tid, p.wv, h * DV + tid, p.cv, (size_t)b * Dv + h * DV + tid);
  // l2norm: q in waves 0,1 ; k in waves 2,3
  if (wave < 4) {
    const float ss = wave_sum(qk * qk);
    if (lane == 0) s_part[wave][0] = ss;
  }
  if (tid == 0) {
    // gate math (std:1293-1294) at the prologue kernel's rounding points
    const float av = bf2f(xrow[p.col_a + h]) + p.dt_bias[h];
    const float bv = bf2f(xrow[p.col_b + h]);
    const float sp = av > 20.f ? av : log1pf(expf(av));
    const float g = -expf(p.A_log[h]) * sp;
    s_sc[0] = __expf(g);
    s_sc[1] = bf_round(sigmoid_exact_(bv));
  }
  __syncthreads();
  if (tid < 2 * DK) {
    const float tot = tid < DK ? s_part[0][0] + s_part[1][0] : s_part[2][0] + s_part[3][0];
    const float nrm = bf_round(qk * (1.0f / sqrtf(tot + 1e-6f)));       // fla l2norm_fwd writes bf16
    if (tid < DK) s_q[tid] = nrm * p.scale;
    else s_k[tid - DK] = nrm;
  }
  __syncthreads();
  // k . (q*scale): every wave computes it redundantly (2 elements per lane)
  const float kq = wave_sum(s_k[2 * lane] * s_q[2 * lane] + s_k[2 * lane + 1] * s_q[2 * lane + 1]);

  // ---- decay + the two column reductions over the rows of this wave --------------------------------------
  const float decay = s_sc[0], beta = s_sc[1];
  float pk[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};
  float kk[DRW];
#pragma unroll
  for (int r4 = 0; r4 < DRW / 4; ++r4) {
    const f32x4 k4 = *(const f32x4*)&s_k[DRW * wave + 4 * r4];
    const f32x4 q4 = *(const f32x4*)&s_q[DRW * wave + 4 * r4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * r4 + i;
      kk[r] = k4[i];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        S[r][c] *= decay;
        pk[c] = fmaf(S[r][c], k4[i], pk[c]);
        pq[c] = fmaf(S[r][c], q4[i], pq[c]);
      }
    }
  }
  *(f32x4*)&s_red[wave][0][4 * lane] = f32x4{pk[0], pk[1], pk[2], pk[3]};
  *(f32x4*)&s_red[wave][1][4 * lane] = f32x4{pq[0], pq[1], pq[2], pq[3]};
  __syncthreads();
  float o4[4], delta[4];
  {
    f32x4 kv = *(const f32x4*)&s_red[0][0][4 * lane], oq = *(const f32x4*)&s_red[0][1][4 * lane];
#pragma unroll
    for (int w = 1; w < DNW; ++w) {
      kv += *(const f32x4*)&s_red[w][0][4 * lane];
      oq += *(const f32x4*)&s_red[w][1][4 * lane];
    }
    const f32x4 v4 = *(const f32x4*)&s_v[4 * lane];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      delta[c] = beta * (v4[c] - kv[c]);
      o4[c] = bf_round(fmaf(delta[c], kq, oq[c]));            // the recurrent kernel stores o in bf16
    }
  }
  // ---- state update + write-back -------------------------------------------------------------------
#pragma unroll
  for (int r = 0; r < DRW; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) S[r][c] = fmaf(kk[r], delta[c], S[r][c]);
  if (p.state_dtype == IVL_F32) {
    float* sp = (float*)p.state + sbase;
#pragma unroll
    for (int r = 0; r < DRW; ++r) *(f32x4*)(sp + (size_t)r * DV) = f32x4{S[r][0], S[r][1], S[r][2], S[r][3]};
  } else {
    bf16_t* sp = (bf16_t*)p.state + sbase;
#pragma unroll
    for (int r = 0; r < DRW; ++r) *(u32x2*)(sp + (size_t)r * DV) = u32x2{pack2bf(S[r][0], S[r][1]), pack2bf(S[r][2], S[r][3])};
  }
  // ---- gated RMSNorm over the head's 256 outputs (every wave holds all of them, 4 per lane) -----------
  if (wave == 0) {
    const float ss = wave_sum(o4[0] * o4[0] + o4[1] * o4[1] + o4[2] * o4[2] + o4[3] * o4[3]);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / 256.0f) + p.eps);
    const u32x2 wv = *(const u32x2*)(p.norm_w + 4 * lane);
    const u32x2 gv = *(const u32x2*)(xrow + p.col_g + h * DV + 4 * lane);
    const float wf[4] = {bflo(wv.x), bfhi(wv.x), bflo(wv.y), bfhi(wv.y)};
    const float gf[4] = {bflo(gv.x), bfhi(gv.x), bflo(gv.y), bfhi(gv.y)};
    float y4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) y4[c] = o4[c] * rstd * wf[c] * gf[c] * sigmoidf_(gf[c]);
    *(u32x2*)(p.y + ((size_t)b * p.H + h) * DV + 4 * lane) = u32x2{pack2bf(y4[0], y4[1]), pack2bf(y4[2], y4[3])};
  }
}

}  // namespace ivl

using namespace ivl;

extern "C" int ivl_gdn_decode_step_fwd(const void* proj, int64_t ld, int col_q, int col_k, int col_v, int col_g,
                                       int col_a, int col_b, const void* conv_wq, const void* conv_wk,
                                       const void* conv_wv, void* conv_state_q, void* conv_state_k, void* conv_state_v,
                                       const float* A_log, const float* dt_bias, const void* norm_weight, float eps,
                                       void* state, int state_dtype, void* y, int B, int H, int K, int V, float scale,
                                       void* stream) {
  IVL_REQUIRE(proj && conv_wq && conv_wk && conv_wv && conv_state_q && conv_state_k && conv_state_v && A_log && dt_bias &&
              norm_weight && state && y, IVL_ERR_INVALID_ARG, "ivl_gdn_decode_step_fwd: NULL pointer");
  IVL_REQUIRE(B > 0 && H > 0, IVL_ERR_INVALID_ARG, "ivl_gdn_decode_step_fwd: B,H must be positive (%d,%d)", B, H);
  IVL_REQUIRE(K == DK && V == DV, IVL_ERR_UNSUPPORTED, "ivl_gdn_decode_step_fwd: built for K=128,V=256 (got %d,%d)", K, V);
  IVL_REQUIRE(state_dtype == IVL_F32 || state_dtype == IVL_BF16, IVL_ERR_INVALID_ARG,
              "ivl_gdn_decode_step_fwd: state dtype must be IVL_F32 or IVL_BF16");
  IVL_REQUIRE(col_g % 4 == 0 && ld % 4 == 0, IVL_ERR_INVALID_ARG, "ivl_gdn_decode_step_fwd: gate columns must be 8-byte aligned");
  DecParams p;
  p.proj = (const bf16_t*)proj; p.ld = ld;
  p.col_q = col_q; p.col_k = col_k; p.col_v = col_v; p.col_g = col_g; p.col_a = col_a; p.col_b = col_b;
  p.wq = (const bf16_t*)conv_wq; p.wk = (const bf16_t*)conv_wk; p.wv = (const bf16_t*)conv_wv;
  p.cq = (bf16_t*)conv_state_q; p.ck = (bf16_t*)conv_state_k; p.cv = (bf16_t*)conv_state_v;
  p.A_log = A_log; p.dt_bias = dt_bias; p.norm_w = (const bf16_t*)norm_weight; p.eps = eps;
  p.state = state; p.state_dtype = state_dtype; p.y = (bf16_t*)y; p.H = H; p.scale = scale;
  hipLaunchKernelGGL(gdn_decode_step_kernel, dim3(B * H), dim3(64 * DNW), 0, (hipStream_t)stream, p);
  return check_launch("ivl_gdn_decode_step_fwd");
}
