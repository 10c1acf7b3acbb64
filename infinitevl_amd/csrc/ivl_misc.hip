// Bandwidth-bound kernels either side of the Gated DeltaNet / SWA kernels:
// short causal conv (+SiLU, state carry-in), gated RMSNorm, gate math, M-RoPE, and the
// device-side position counter.  All are HBM-bound: 16-byte vector loads, one pass.
#include <stdarg.h>

#include "ivl_common.h"

namespace ivl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return IVL_ERR_LAUNCH;
  }
  return IVL_OK;
}

// ------------------------------------------------------------------------------------------------
// short conv: y[b,t,d] = act(sum_j w[d,j] * ext[t+1+j, d]), ext = concat(state_in^T (W rows), x).
// One thread = 8 channels x TCH consecutive tokens; the thread of token-chunk 0 is the only reader
// of state_in for its channels and also the writer of state_out, so state_out may alias state_in.
// ------------------------------------------------------------------------------------------------
constexpr int CONV_W = 4;
constexpr int CONV_TCH = 8;

// bias (fla's ShortConvolution(bias=True), not used by InfiniteVL): the accumulator starts from it, as in causal_conv1d.
__global__ __launch_bounds__(256) void short_conv_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias, const bf16_t* state_in,
    bf16_t* __restrict__ y, bf16_t* state_out, int B, int T, int D, int apply_silu) {
  const int DG = D / 8;
  const int NCH = (T + CONV_TCH - 1) / CONV_TCH;
  const long long total = (long long)B * NCH * DG;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int dg = (int)(idx % DG);
    const int ch = (int)((idx / DG) % NCH);
    const int b = (int)(idx / ((long long)DG * NCH));
    const int d0 = dg * 8;
    const int t0 = ch * CONV_TCH;

    // weights: 8 channels x 4 taps
    float wf[8][CONV_W];
    {
      const u32x4* wp = (const u32x4*)(w + (size_t)d0 * CONV_W);
      u32x4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
      unsigned int ww[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w,
                             w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        wf[c][0] = bflo(ww[2 * c]);
        wf[c][1] = bfhi(ww[2 * c]);
        wf[c][2] = bflo(ww[2 * c + 1]);
        wf[c][3] = bfhi(ww[2 * c + 1]);
      }
    }
    // history state for these 8 channels (only needed by chunk 0)
    float st[8][CONV_W];
    if (ch == 0) {
      if (state_in != nullptr) {
        const u32x4* sp = (const u32x4*)(state_in + ((size_t)b * D + d0) * CONV_W);
        u32x4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
        unsigned int ss[16] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w,
                               s2.x, s2.y, s2.z, s2.w, s3.x, s3.y, s3.z, s3.w};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          st[c][0] = bflo(ss[2 * c]);
          st[c][1] = bfhi(ss[2 * c]);
          st[c][2] = bflo(ss[2 * c + 1]);
          st[c][3] = bfhi(ss[2 * c + 1]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int j = 0; j < CONV_W; ++j) st[c][j] = 0.f;
      }
    }
    // sliding window of the last 3 inputs: win[k][c] = input at time (t0-3+k)
    float win[3][8];
    const bf16_t* xb = x + (size_t)b * T * D + d0;
    if (ch == 0) {
      // times -3,-2,-1 come from the carried state (newest last): state[...,1..3]
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 8; ++c) win[k][c] = st[c][k + 1];
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        u32x4 v = *(const u32x4*)(xb + (size_t)(t0 - 3 + k) * D);
        win[k][0] = bflo(v.x); win[k][1] = bfhi(v.x); win[k][2] = bflo(v.y); win[k][3] = bfhi(v.y);
        win[k][4] = bflo(v.z); win[k][5] = bfhi(v.z); win[k][6] = bflo(v.w); win[k][7] = bfhi(v.w);
      }
    }
    float bf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) unpack8(*(const u32x4*)(bias + d0), bf);
    const int tend = min(t0 + CONV_TCH, T);
    for (int t = t0; t < tend; ++t) {
      u32x4 v = *(const u32x4*)(xb + (size_t)t * D);
      float cur[8] = {bflo(v.x), bfhi(v.x), bflo(v.y), bfhi(v.y), bflo(v.z), bfhi(v.z), bflo(v.w), bfhi(v.w)};
      float out[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float a = bias != nullptr ? fmaf(wf[c][0], win[0][c], bf[c]) : wf[c][0] * win[0][c];
        a = fmaf(wf[c][1], win[1][c], a);
        a = fmaf(wf[c][2], win[2][c], a);
        a = fmaf(wf[c][3], cur[c], a);
        if (apply_silu) a = a * sigmoidf_(a);
        out[c] = a;
        win[0][c] = win[1][c];
        win[1][c] = win[2][c];
        win[2][c] = cur[c];
      }
      u32x4 o;
      o.x = pack2bf(out[0], out[1]); o.y = pack2bf(out[2], out[3]);
      o.z = pack2bf(out[4], out[5]); o.w = pack2bf(out[6], out[7]);
      *(u32x4*)(y + ((size_t)b * T + t) * D + d0) = o;
    }
    if (ch == 0 && state_out != nullptr) {
      // new_state[c][j] = ext[T + j], ext = [state(4), x(T)]
      float ns[8][CONV_W];
#pragma unroll
      for (int j = 0; j < CONV_W; ++j) {
        const int e = T + j;            // index into ext
        if (e >= CONV_W) {
          u32x4 v = *(const u32x4*)(xb + (size_t)(e - CONV_W) * D);
          ns[0][j] = bflo(v.x); ns[1][j] = bfhi(v.x); ns[2][j] = bflo(v.y); ns[3][j] = bfhi(v.y);
          ns[4][j] = bflo(v.z); ns[5][j] = bfhi(v.z); ns[6][j] = bflo(v.w); ns[7][j] = bfhi(v.w);
        } else {
#pragma unroll
          for (int e2 = 0; e2 < CONV_W; ++e2)
            if (e == e2) {
#pragma unroll
              for (int c = 0; c < 8; ++c) ns[c][j] = st[c][e2];
            }
        }
      }
      u32x4* op = (u32x4*)(state_out + ((size_t)b * D + d0) * CONV_W);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4 o;
        o.x = pack2bf(ns[2 * i][0], ns[2 * i][1]);
        o.y = pack2bf(ns[2 * i][2], ns[2 * i][3]);
        o.z = pack2bf(ns[2 * i + 1][0], ns[2 * i + 1][1]);
        o.w = pack2bf(ns[2 * i + 1][2], ns[2 * i + 1][3]);
        op[i] = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gated RMSNorm over rows of 256: half a wavefront (32 lanes x 8 bf16) per row.
// ------------------------------------------------------------------------------------------------
// GATED = false: the plain RMSNorm of fla (fla:modules/layernorm.py: y = x * rstd * w in fp32, rounded ONCE at the store) -- the
// output norm of a GatedDeltaNet built with use_gate=False (std:1213).
// RES (fla's residual= / prenorm= / residual_in_fp32= options, fla:modules/fused_norm_gate.py:54-60; not used by InfiniteVL):
// the row is x + residual in fp32 (residual: bf16 or fp32), stored to residual_out (bf16 or fp32) when wanted, and the
// statistics and the output are those of the UNROUNDED fp32 row, as in the reference kernel.
struct NormRes {
  const void* residual; int residual_dtype;      // NULL: none
  void* residual_out; int residual_out_dtype;    // NULL: not stored
};
template <bool GATED, bool RES = false>
__global__ __launch_bounds__(256) void rmsnorm_gate_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ gate, const bf16_t* __restrict__ weight,
    bf16_t* __restrict__ y, int rows, float eps, NormRes nr = NormRes{nullptr, 0, nullptr, 0}) {
  const int lane32 = threadIdx.x & 31;
  const long long row0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const long long stride = ((long long)gridDim.x * blockDim.x) >> 5;
  u32x4 wv = *(const u32x4*)(weight + lane32 * 8);
  const float wf[8] = {bflo(wv.x), bfhi(wv.x), bflo(wv.y), bfhi(wv.y), bflo(wv.z), bfhi(wv.z), bflo(wv.w), bfhi(wv.w)};
  for (long long r = row0; r < rows; r += stride) {
    u32x4 xv = *(const u32x4*)(x + r * 256 + lane32 * 8);
    u32x4 gv = u32x4{0u, 0u, 0u, 0u};
    if constexpr (GATED) gv = *(const u32x4*)(gate + r * 256 + lane32 * 8);
    float xf[8] = {bflo(xv.x), bfhi(xv.x), bflo(xv.y), bfhi(xv.y), bflo(xv.z), bfhi(xv.z), bflo(xv.w), bfhi(xv.w)};
    float gf[8] = {bflo(gv.x), bfhi(gv.x), bflo(gv.y), bfhi(gv.y), bflo(gv.z), bfhi(gv.z), bflo(gv.w), bfhi(gv.w)};
    if constexpr (RES) {
      const size_t e0 = (size_t)r * 256 + lane32 * 8;
      if (nr.residual != nullptr) {
        float rf[8];
        if (nr.residual_dtype == IVL_F32) {
          const f32x4 a = *(const f32x4*)((const float*)nr.residual + e0), b = *(const f32x4*)((const float*)nr.residual + e0 + 4);
          rf[0] = a[0]; rf[1] = a[1]; rf[2] = a[2]; rf[3] = a[3]; rf[4] = b[0]; rf[5] = b[1]; rf[6] = b[2]; rf[7] = b[3];
        } else {
          unpack8(*(const u32x4*)((const bf16_t*)nr.residual + e0), rf);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) xf[i] += rf[i];
      }
      if (nr.residual_out != nullptr) {
        if (nr.residual_out_dtype == IVL_F32) {
          *(f32x4*)((float*)nr.residual_out + e0) = f32x4{xf[0], xf[1], xf[2], xf[3]};
          *(f32x4*)((float*)nr.residual_out + e0 + 4) = f32x4{xf[4], xf[5], xf[6], xf[7]};
        } else {
          *(u32x4*)((bf16_t*)nr.residual_out + e0) = pack8(xf);
        }
      }
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss = fmaf(xf[i], xf[i], ss);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / 256.0f) + eps);
    float o8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o8[i] = GATED ? xf[i] * rstd * wf[i] * gf[i] * sigmoidf_(gf[i]) : xf[i] * rstd * wf[i];
    u32x4 ov;
    ov.x = pack2bf(o8[0], o8[1]); ov.y = pack2bf(o8[2], o8[3]);
    ov.z = pack2bf(o8[4], o8[5]); ov.w = pack2bf(o8[6], o8[7]);
    *(u32x4*)(y + r * 256 + lane32 * 8) = ov;
  }
}

// ------------------------------------------------------------------------------------------------
// gate math: beta = sigmoid(b) -> bf16 ; g = -exp(A_log[h]) * softplus(a + dt_bias[h]) -> fp32
// softplus follows torch (threshold 20).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gdn_gate_kernel(
    const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const float* __restrict__ A_log,
    const float* __restrict__ dt_bias, float* __restrict__ g, bf16_t* __restrict__ beta, long long n, int H) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int h = (int)(i % H);
    const float av = bf2f(a[i]) + dt_bias[h];
    const float sp = av > 20.f ? av : log1pf(expf(av));
    g[i] = -expf(A_log[h]) * sp;
    beta[i] = f2bf(sigmoid_exact_(bf2f(b[i])));
  }
}

// ------------------------------------------------------------------------------------------------
// M-RoPE in place on time-major q [B,T,Hq,d], k [B,T,Hkv,d]; cos/sin bf16 [3,B,T,d].
// One thread = one head x 8 channels of the first half and their partners in the second half.
// bf16 rounding after each product and after the sum (== torch eager bf16 arithmetic, std:982-983).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mrope_kernel(
    bf16_t* q, bf16_t* k, const bf16_t* __restrict__ cosp, const bf16_t* __restrict__ sinp,
    int B, int T, int Hq, int Hkv, int d, int s0, int s1) {
  const int half = d / 2;
  const int CG = half / 8;                 // channel groups per head (first half)
  const int HT = Hq + Hkv;
  const long long total = (long long)B * T * HT * CG;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % CG);
    const int h = (int)((idx / CG) % HT);
    const long long bt = idx / ((long long)CG * HT);
    const int c0 = cg * 8;
    bf16_t* base = h < Hq ? q + (bt * Hq + h) * d : k + (bt * Hkv + (h - Hq)) * d;
    u32x4 lo = *(const u32x4*)(base + c0);
    u32x4 hi = *(const u32x4*)(base + c0 + half);
    float x1[8] = {bflo(lo.x), bfhi(lo.x), bflo(lo.y), bfhi(lo.y), bflo(lo.z), bfhi(lo.z), bflo(lo.w), bfhi(lo.w)};
    float x2[8] = {bflo(hi.x), bfhi(hi.x), bflo(hi.y), bfhi(hi.y), bflo(hi.z), bfhi(hi.z), bflo(hi.w), bfhi(hi.w)};
    float o1[8], o2[8];
    const long long plane = (long long)B * T * d;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i;                       // channel in the first half; partner c+half has the same section
      const int sec = c < s0 ? 0 : (c < s0 + s1 ? 1 : 2);
      const long long off = sec * plane + bt * d;
      const float c1 = bf2f(cosp[off + c]), sn1 = bf2f(sinp[off + c]);
      const float c2 = bf2f(cosp[off + c + half]), sn2 = bf2f(sinp[off + c + half]);
      o1[i] = bf_round(bf_round(x1[i] * c1) + bf_round(-x2[i] * sn1));
      o2[i] = bf_round(bf_round(x2[i] * c2) + bf_round(x1[i] * sn2));
    }
    u32x4 w1, w2;
    w1.x = pack2bf(o1[0], o1[1]); w1.y = pack2bf(o1[2], o1[3]); w1.z = pack2bf(o1[4], o1[5]); w1.w = pack2bf(o1[6], o1[7]);
    w2.x = pack2bf(o2[0], o2[1]); w2.y = pack2bf(o2[2], o2[3]); w2.z = pack2bf(o2[4], o2[5]); w2.w = pack2bf(o2[6], o2[7]);
    *(u32x4*)(base + c0) = w1;
    *(u32x4*)(base + c0 + half) = w2;
  }
}

// 3-D rotary tables (std:896-930): row r = one (axis, batch, token) position; cos / sin of position * inv_freq[j] in fp32,
// the 64 frequencies repeated over both halves of the 128 channels (emb = cat(freqs, freqs)), scaled, rounded to bf16.
// One launch instead of the eager chain (cast, outer product, cat, cos, sin, two scalings, two casts).
__global__ __launch_bounds__(256) void rope_tables_kernel(const long long* __restrict__ pos, const float* __restrict__ inv_freq,
                                                         bf16_t* __restrict__ cos_o, bf16_t* __restrict__ sin_o, int rows,
                                                         int half_dim, float scaling) {
  const long long total = (long long)rows * half_dim;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % half_dim);
    const long long r = i / half_dim;
    const float f = (float)pos[r] * inv_freq[j];
    const bf16_t c = f2bf(cosf(f) * scaling), s_ = f2bf(sinf(f) * scaling);
    cos_o[r * 2 * half_dim + j] = c;
    cos_o[r * 2 * half_dim + half_dim + j] = c;
    sin_o[r * 2 * half_dim + j] = s_;
    sin_o[r * 2 * half_dim + half_dim + j] = s_;
  }
}

__global__ void counter_add_kernel(long long* c, long long delta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c += delta;
}

static inline int grid_for(long long work_items, int block = 256, int cap = 256 * 8) {
  long long g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace ivl

using namespace ivl;

extern "C" int ivl_abi_version(void) { return IVL_ABI_VERSION; }
#ifdef IVL_TRACE
// Developer build only (libivl_hip_trace.so, never shipped): device buffer of >= 64 int64 slots that the instrumented
// kernels fill with phase durations (NULL switches the timeline off).
namespace ivl {
void trace_set_gdn(void* p);
void trace_set_swa(void* p);
void trace_set_vis(void* p);
extern int g_scan_ncw;
}
extern "C" IVL_API void ivl_debug_set_trace(void* device_buffer) {
  ivl::trace_set_gdn(device_buffer);
  ivl::trace_set_swa(device_buffer);
  ivl::trace_set_vis(device_buffer);
}
extern "C" IVL_API void ivl_debug_set_scan_waves(int ncw) { ivl::g_scan_ncw = ncw; }
#endif
extern "C" const char* ivl_last_error(void) { return g_err; }

extern "C" int ivl_short_conv_fwd(const void* x, const void* weight, const void* state_in, void* y, void* state_out,
                                  int B, int T, int D, int W, int apply_silu, void* stream) {
  IVL_REQUIRE(x && weight && y, IVL_ERR_INVALID_ARG, "ivl_short_conv_fwd: NULL x/weight/y");
  IVL_REQUIRE(B > 0 && T > 0 && D > 0, IVL_ERR_INVALID_ARG, "ivl_short_conv_fwd: B,T,D must be positive (got %d,%d,%d)", B, T, D);
  IVL_REQUIRE(W == CONV_W, IVL_ERR_UNSUPPORTED, "ivl_short_conv_fwd: kernel size %d unsupported (built for 4)", W);
  IVL_REQUIRE(D % 8 == 0, IVL_ERR_UNSUPPORTED, "ivl_short_conv_fwd: D=%d must be a multiple of 8", D);
  const long long items = (long long)B * ((T + CONV_TCH - 1) / CONV_TCH) * (D / 8);
  hipLaunchKernelGGL(short_conv_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)weight, (const bf16_t*)nullptr, (const bf16_t*)state_in, (bf16_t*)y, (bf16_t*)state_out,
                     B, T, D, apply_silu);
  return check_launch("ivl_short_conv_fwd");
}

extern "C" int ivl_rmsnorm_swish_gate_fwd(const void* x, const void* gate, const void* weight, void* y,
                                          int rows, int N, float eps, void* stream) {
  IVL_REQUIRE(x && weight && y, IVL_ERR_INVALID_ARG, "ivl_rmsnorm_swish_gate_fwd: NULL pointer");
  IVL_REQUIRE(rows > 0, IVL_ERR_INVALID_ARG, "ivl_rmsnorm_swish_gate_fwd: rows=%d", rows);
  IVL_REQUIRE(N == 256, IVL_ERR_UNSUPPORTED, "ivl_rmsnorm_swish_gate_fwd: N=%d unsupported (built for head_v_dim 256)", N);
  if (gate != nullptr)
    hipLaunchKernelGGL(rmsnorm_gate_kernel<true>, dim3(grid_for((long long)rows * 32)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (const bf16_t*)gate, (const bf16_t*)weight, (bf16_t*)y, rows, eps);
  else
    hipLaunchKernelGGL(rmsnorm_gate_kernel<false>, dim3(grid_for((long long)rows * 32)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (const bf16_t*)nullptr, (const bf16_t*)weight, (bf16_t*)y, rows, eps);
  return check_launch("ivl_rmsnorm_swish_gate_fwd");
}

extern "C" int ivl_rmsnorm_swish_gate_res_fwd(const void* x, const void* gate, const void* weight, const void* residual, int residual_dtype,
                                              void* residual_out, int residual_out_dtype, void* y, int rows, int N, float eps, void* stream) {
  IVL_REQUIRE(x && gate && weight && y, IVL_ERR_INVALID_ARG, "ivl_rmsnorm_swish_gate_res_fwd: NULL pointer");
  IVL_REQUIRE(rows > 0, IVL_ERR_INVALID_ARG, "ivl_rmsnorm_swish_gate_res_fwd: rows=%d", rows);
  IVL_REQUIRE(N == 256, IVL_ERR_UNSUPPORTED, "ivl_rmsnorm_swish_gate_res_fwd: N=%d unsupported (built for head_v_dim 256)", N);
  IVL_REQUIRE((residual == nullptr || residual_dtype == IVL_F32 || residual_dtype == IVL_BF16) &&
              (residual_out == nullptr || residual_out_dtype == IVL_F32 || residual_out_dtype == IVL_BF16),
              IVL_ERR_INVALID_ARG, "ivl_rmsnorm_swish_gate_res_fwd: residual dtype must be IVL_F32 or IVL_BF16");
  hipLaunchKernelGGL((rmsnorm_gate_kernel<true, true>), dim3(grid_for((long long)rows * 32)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)gate, (const bf16_t*)weight, (bf16_t*)y, rows, eps,
                     NormRes{residual, residual_dtype, residual_out, residual_out_dtype});
  return check_launch("ivl_rmsnorm_swish_gate_res_fwd");
}

extern "C" int ivl_short_conv_bias_fwd(const void* x, const void* weight, const void* bias, const void* state_in, void* y, void* state_out,
                                       int B, int T, int D, int W, int apply_silu, void* stream) {
  IVL_REQUIRE(x && weight && y, IVL_ERR_INVALID_ARG, "ivl_short_conv_bias_fwd: NULL x/weight/y");
  IVL_REQUIRE(B > 0 && T > 0 && D > 0, IVL_ERR_INVALID_ARG, "ivl_short_conv_bias_fwd: B,T,D must be positive (got %d,%d,%d)", B, T, D);
  IVL_REQUIRE(W == CONV_W, IVL_ERR_UNSUPPORTED, "ivl_short_conv_bias_fwd: kernel size %d unsupported (built for 4)", W);
  IVL_REQUIRE(D % 8 == 0, IVL_ERR_UNSUPPORTED, "ivl_short_conv_bias_fwd: D=%d must be a multiple of 8", D);
  const long long items = (long long)B * ((T + CONV_TCH - 1) / CONV_TCH) * (D / 8);
  hipLaunchKernelGGL(short_conv_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)weight, (const bf16_t*)bias, (const bf16_t*)state_in, (bf16_t*)y, (bf16_t*)state_out,
                     B, T, D, apply_silu);
  return check_launch("ivl_short_conv_bias_fwd");
}

extern "C" int ivl_gdn_gate_fwd(const void* a, const void* b, const float* A_log, const float* dt_bias,
                                float* g, void* beta, int rows, int H, void* stream) {
  IVL_REQUIRE(a && b && A_log && dt_bias && g && beta, IVL_ERR_INVALID_ARG, "ivl_gdn_gate_fwd: NULL pointer");
  IVL_REQUIRE(rows > 0 && H > 0, IVL_ERR_INVALID_ARG, "ivl_gdn_gate_fwd: rows=%d H=%d", rows, H);
  const long long n = (long long)rows * H;
  hipLaunchKernelGGL(gdn_gate_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)a, (const bf16_t*)b, A_log, dt_bias, g, (bf16_t*)beta, n, H);
  return check_launch("ivl_gdn_gate_fwd");
}

extern "C" int ivl_mrope_fwd(void* q, void* k, const void* cos, const void* sin,
                             int B, int T, int Hq, int Hkv, int d, int s0, int s1, int s2, void* stream) {
  IVL_REQUIRE(q && k && cos && sin, IVL_ERR_INVALID_ARG, "ivl_mrope_fwd: NULL pointer");
  IVL_REQUIRE(B > 0 && T > 0 && Hq > 0 && Hkv > 0, IVL_ERR_INVALID_ARG, "ivl_mrope_fwd: bad sizes");
  IVL_REQUIRE(d % 16 == 0 && s0 + s1 + s2 == d / 2, IVL_ERR_UNSUPPORTED,
              "ivl_mrope_fwd: need d%%16==0 and s0+s1+s2==d/2 (d=%d, sections %d,%d,%d)", d, s0, s1, s2);
  const long long items = (long long)B * T * (Hq + Hkv) * (d / 16);
  hipLaunchKernelGGL(mrope_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)q, (bf16_t*)k, (const bf16_t*)cos, (const bf16_t*)sin, B, T, Hq, Hkv, d, s0, s1);
  return check_launch("ivl_mrope_fwd");
}

extern "C" int ivl_rope_tables_fwd(const int64_t* position_ids, const float* inv_freq, void* cos_out, void* sin_out, int rows,
                                   int half_dim, float attention_scaling, void* stream) {
  IVL_REQUIRE(position_ids && inv_freq && cos_out && sin_out, IVL_ERR_INVALID_ARG, "ivl_rope_tables_fwd: NULL pointer");
  IVL_REQUIRE(rows > 0 && half_dim > 0 && half_dim % 2 == 0, IVL_ERR_INVALID_ARG, "ivl_rope_tables_fwd: bad sizes rows=%d half_dim=%d",
              rows, half_dim);
  const long long total = (long long)rows * half_dim;
  long long gb = (total + 255) / 256;
  if (gb > 1024) gb = 1024;
  hipLaunchKernelGGL(rope_tables_kernel, dim3((int)gb), dim3(256), 0, (hipStream_t)stream, (const long long*)position_ids, inv_freq,
                     (bf16_t*)cos_out, (bf16_t*)sin_out, rows, half_dim, attention_scaling);
  return check_launch("ivl_rope_tables_fwd");
}

extern "C" int ivl_counter_add(int64_t* counter, int64_t delta, void* stream) {
  IVL_REQUIRE(counter, IVL_ERR_INVALID_ARG, "ivl_counter_add: NULL counter");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)counter, (long long)delta);
  return check_launch("ivl_counter_add");
}
