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

// The fields a workgroup needs for its FIRST loads (state rows, the token's projection row, conv taps) are passed a second time as
// leading scalar / pointer parameters: those are preloaded into SGPRs at wave launch (Makefile: -amdgpu-kernarg-preload-count=16,
// exactly these 16 dwords), a by-value struct is not -- the kernel's first loads no longer wait for a cold read of its own kernarg
// segment.  DEC_HEAD_APPLY overrides the struct's copies, so the rest of the kernel reads `p` as before.
#define DEC_HEAD_PARAMS void* hd_state, int hd_state_dtype, int hd_H, const bf16_t* hd_proj, long long hd_ld, int hd_col_q, int hd_col_k, \
                        int hd_col_v, const bf16_t* hd_wq, const bf16_t* hd_wk
#define DEC_HEAD_ARGS(p) (p).state, (p).state_dtype, (p).H, (p).proj, (p).ld, (p).col_q, (p).col_k, (p).col_v, (p).wq, (p).wk
#define DEC_HEAD_APPLY(p)                                                                                                    \
  (p).state = hd_state; (p).state_dtype = hd_state_dtype; (p).H = hd_H; (p).proj = hd_proj; (p).ld = hd_ld; (p).col_q = hd_col_q; \
  (p).col_k = hd_col_k; (p).col_v = hd_col_v; (p).wq = hd_wq; (p).wk = hd_wk

__global__ __launch_bounds__(64 * DNW) void gdn_decode_step_kernel(DEC_HEAD_PARAMS, DecParams p) {
  DEC_HEAD_APPLY(p);
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

// ------------------------------------------------------------------------------------------------------------------------------
// Round 5: the same step over 4 x as many workgroups.  The 16 workgroups of the kernel above stream 2.1 MB of
// state in ~6 us (0.04 of the HBM roof: 16 CUs' worth of memory parallelism).  Here a workgroup = (batch, head, QUARTER of the 256
// value columns): its state slab S[128][64], the delta rule and o = S^T q are column-local, so the quarters never exchange
// anything.  What spans a head is moved out:
//   * the gated RMSNorm (a 256-wide row statistic) -> the prologue of the o_proj weight stream that follows (linear_small_m.hip,
//     ivl_gdn_out_linear_small_m_fwd): this kernel writes the delta rule's bf16 output rows un-normalised;
//   * the conv STATES of q and k (every quarter reads them, nobody may write them while a quarter may still read) -> shifted by
//     that same next launch; the v conv state is column-local and is shifted here.
// Thread t: column pair cp = t & 31 of the quarter, row group rg = t >> 5 (16 rows): the per-column sums over the rows run in
// the order of the kernel above (16-row fma chains, then the eight groups in order): o, the state and the conv states are
// bit-identical to it.
constexpr int DSQ = 4;                 // column quarters per head
constexpr int DSC = DV / DSQ;          // 64 columns per workgroup
__device__ __forceinline__ float conv1_ro(const bf16_t* xrow, int col, const bf16_t* w, int wch, const bf16_t* st, size_t ch) {
  const u32x2 wv = *(const u32x2*)(w + (size_t)wch * 4);
  const u32x2 sv = *(const u32x2*)(st + ch * 4);
  const float x = bf2f(xrow[col]);
  float a = bflo(wv.x) * bfhi(sv.x);
  a = fmaf(bfhi(wv.x), bflo(sv.y), a);
  a = fmaf(bflo(wv.y), bfhi(sv.y), a);
  a = fmaf(bfhi(wv.y), x, a);
  return bf_round(a * sigmoidf_(a));
}

__global__ __launch_bounds__(256) void gdn_decode_split_kernel(DEC_HEAD_PARAMS, DecParams p) {
  DEC_HEAD_APPLY(p);
  __shared__ __attribute__((aligned(16))) float s_k[DK], s_q[DK], s_v[DSC];
  __shared__ __attribute__((aligned(16))) float s_red[8][2][DSC];
  __shared__ float s_part[4][2];
  __shared__ float s_sc[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = blockIdx.x % DSQ, bh = blockIdx.x / DSQ, b = bh / p.H, h = bh % p.H;
  const bf16_t* xrow = p.proj + (long long)b * p.ld;
  const int cp = tid & 31, rg = tid >> 5;
  const int c0 = DSC * qt + 2 * cp;                                  // first of the thread's two columns (of the head's 256)

  float S[16][2];
  const size_t sbase = ((size_t)bh * DK + 16 * rg) * DV + c0;
  if (p.state_dtype == IVL_F32) {
    const float* sp = (const float*)p.state + sbase;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float2 v2 = *(const float2*)(sp + (size_t)r * DV);
      S[r][0] = v2.x; S[r][1] = v2.y;
    }
  } else {
    const bf16_t* sp = (const bf16_t*)p.state + sbase;
    unsigned int raw[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) raw[r] = *(const unsigned int*)(sp + (size_t)r * DV);
#pragma unroll
    for (int r = 0; r < 16; ++r) { S[r][0] = bflo(raw[r]); S[r][1] = bfhi(raw[r]); }
  }
  // convs: q channel tid (tid < 128) | k channel tid - 128: the states are READ here and shifted by the next launch;
  // v channel 64 qt + tid (tid < 64): column-local, shifted in place
  const int Dq = p.H * DK, Dv = p.H * DV;
  float qk;
  if (tid < DK) qk = conv1_ro(xrow, p.col_q + h * DK + tid, p.wq, h * DK + tid, p.cq, (size_t)b * Dq + h * DK + tid);
  else qk = conv1_ro(xrow, p.col_k + h * DK + (tid - DK), p.wk, h * DK + tid - DK, p.ck, (size_t)b * Dq + h * DK + (tid - DK));
  if (tid < DSC) s_v[tid] = conv1(xrow, p.col_v + h * DV + DSC * qt + tid, p.wv, h * DV + DSC * qt + tid, p.cv, (size_t)b * Dv + h * DV + DSC * qt + tid);
  {
    const float ss = wave_sum(qk * qk);
    if (lane == 0) s_part[wave][0] = ss;
  }
  if (tid == 0) {
    const float av = bf2f(xrow[p.col_a + h]) + p.dt_bias[h];
    const float bv = bf2f(xrow[p.col_b + h]);
    const float sp = av > 20.f ? av : log1pf(expf(av));
    const float g = -expf(p.A_log[h]) * sp;
    s_sc[0] = __expf(g);
    s_sc[1] = bf_round(sigmoid_exact_(bv));
  }
  __syncthreads();
  {
    const float tot = tid < DK ? s_part[0][0] + s_part[1][0] : s_part[2][0] + s_part[3][0];
    const float nrm = bf_round(qk * (1.0f / sqrtf(tot + 1e-6f)));
    if (tid < DK) s_q[tid] = nrm * p.scale;
    else s_k[tid - DK] = nrm;
  }
  __syncthreads();
  const float kq = wave_sum(s_k[2 * lane] * s_q[2 * lane] + s_k[2 * lane + 1] * s_q[2 * lane + 1]);
  const float decay = s_sc[0], beta = s_sc[1];
  float pk[2] = {0.f, 0.f}, pq[2] = {0.f, 0.f};
  float kk[16];
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const f32x4 k4 = *(const f32x4*)&s_k[16 * rg + 4 * r4];
    const f32x4 q4 = *(const f32x4*)&s_q[16 * rg + 4 * r4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * r4 + i;
      kk[r] = k4[i];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        S[r][c] *= decay;
        pk[c] = fmaf(S[r][c], k4[i], pk[c]);
        pq[c] = fmaf(S[r][c], q4[i], pq[c]);
      }
    }
  }
  *(float2*)&s_red[rg][0][2 * cp] = float2{pk[0], pk[1]};
  *(float2*)&s_red[rg][1][2 * cp] = float2{pq[0], pq[1]};
  __syncthreads();
  float delta[2], o2[2];
  {
    float2 kv = *(const float2*)&s_red[0][0][2 * cp], oq = *(const float2*)&s_red[0][1][2 * cp];
#pragma unroll
    for (int w = 1; w < 8; ++w) {
      const float2 a = *(const float2*)&s_red[w][0][2 * cp], c = *(const float2*)&s_red[w][1][2 * cp];
      kv.x += a.x; kv.y += a.y; oq.x += c.x; oq.y += c.y;
    }
    const float2 v2 = *(const float2*)&s_v[2 * cp];
    delta[0] = beta * (v2.x - kv.x); delta[1] = beta * (v2.y - kv.y);
    o2[0] = bf_round(fmaf(delta[0], kq, oq.x)); o2[1] = bf_round(fmaf(delta[1], kq, oq.y));
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) S[r][c] = fmaf(kk[r], delta[c], S[r][c]);
  if (p.state_dtype == IVL_F32) {
    float* sp = (float*)p.state + sbase;
#pragma unroll
    for (int r = 0; r < 16; ++r) *(float2*)(sp + (size_t)r * DV) = float2{S[r][0], S[r][1]};
  } else {
    bf16_t* sp = (bf16_t*)p.state + sbase;
#pragma unroll
    for (int r = 0; r < 16; ++r) *(unsigned int*)(sp + (size_t)r * DV) = pack2bf(S[r][0], S[r][1]);
  }
  if (rg == 0) *(unsigned int*)(p.y + ((size_t)b * p.H + h) * DV + c0) = pack2bf(o2[0], o2[1]);      // un-normalised bf16 o (exact: already rounded)
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
  hipLaunchKernelGGL(gdn_decode_step_kernel, dim3(B * H), dim3(64 * DNW), 0, (hipStream_t)stream, DEC_HEAD_ARGS(p), p);
  return check_launch("ivl_gdn_decode_step_fwd");
}

// The split form of the step (above): y receives the UN-NORMALISED bf16 delta-rule output [B, H*256]; the conv states of q and k are
// read, not written; ivl_gdn_out_linear_small_m_fwd (the o_proj launch) applies the gated RMSNorm and shifts those two states.
extern "C" int ivl_gdn_decode_split_fwd(const void* proj, int64_t ld, int col_q, int col_k, int col_v, int col_a, int col_b,
                                        const void* conv_wq, const void* conv_wk, const void* conv_wv, const void* conv_state_q,
                                        const void* conv_state_k, void* conv_state_v, const float* A_log, const float* dt_bias,
                                        void* state, int state_dtype, void* o_raw, int B, int H, int K, int V, float scale,
                                        void* stream) {
  IVL_REQUIRE(proj && conv_wq && conv_wk && conv_wv && conv_state_q && conv_state_k && conv_state_v && A_log && dt_bias && state &&
              o_raw, IVL_ERR_INVALID_ARG, "ivl_gdn_decode_split_fwd: NULL pointer");
  IVL_REQUIRE(B > 0 && H > 0, IVL_ERR_INVALID_ARG, "ivl_gdn_decode_split_fwd: B,H must be positive (%d,%d)", B, H);
  IVL_REQUIRE(K == DK && V == DV, IVL_ERR_UNSUPPORTED, "ivl_gdn_decode_split_fwd: built for K=128,V=256 (got %d,%d)", K, V);
  IVL_REQUIRE(state_dtype == IVL_F32 || state_dtype == IVL_BF16, IVL_ERR_INVALID_ARG,
              "ivl_gdn_decode_split_fwd: state dtype must be IVL_F32 or IVL_BF16");
  DecParams p;
  p.proj = (const bf16_t*)proj; p.ld = ld;
  p.col_q = col_q; p.col_k = col_k; p.col_v = col_v; p.col_g = 0; p.col_a = col_a; p.col_b = col_b;
  p.wq = (const bf16_t*)conv_wq; p.wk = (const bf16_t*)conv_wk; p.wv = (const bf16_t*)conv_wv;
  p.cq = (bf16_t*)conv_state_q; p.ck = (bf16_t*)conv_state_k; p.cv = (bf16_t*)conv_state_v;
  p.A_log = A_log; p.dt_bias = dt_bias; p.norm_w = nullptr; p.eps = 0.f;
  p.state = state; p.state_dtype = state_dtype; p.y = (bf16_t*)o_raw; p.H = H; p.scale = scale;
  hipLaunchKernelGGL(gdn_decode_split_kernel, dim3(B * H * DSQ), dim3(256), 0, (hipStream_t)stream, DEC_HEAD_ARGS(p), p);
  return check_launch("ivl_gdn_decode_split_fwd");
}
