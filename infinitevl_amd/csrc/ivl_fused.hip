// Prologue / epilogue fusion around the Gated DeltaNet and SWA kernels:
// the steps immediately either side of the path that the reference runs as a dozen separate launches.
//
//   ivl_gdn_prologue_fwd     3 short convs (+SiLU, state carry-in, in-place state) + gate math, reading the
//                            q|k|v|..|a|b columns of ONE fused projection output (row stride ld)
//   ivl_rmsnorm_swish_gate_strided_fwd   gated RMSNorm with the gate read in place from that same buffer
//   ivl_mrope_strided_fwd    M-RoPE in place on q|k column blocks of a fused qkv projection output
//   ivl_add_rmsnorm_fwd      (residual add +) RMSNorm of the decoder layer in one pass
//   ivl_silu_mul_fwd         SwiGLU gate: silu(a) * b over the two halves of a fused gate|up projection
// All HBM-bound, 16-byte vector accesses, rounding points identical to the unfused torch bf16 ops.
#include "ivl_common.h"

namespace ivl {

constexpr int FC_W = 4;       // conv taps
// tokens per conv thread: 8 for long calls (3 halo rows re-read per 8 tokens); 4 for the 256-token step, where 8
// would leave half the chip idle (128 workgroups; 7.8 -> 6.0 us).  Not below 4: only the thread of chunk 0 may touch
// the carried state (it is updated in place), so chunk 1 must start at a token >= 3.

struct ConvSeg {
  const bf16_t* w;            // [D,4]
  const bf16_t* state_in;     // [B,D,4] or NULL
  bf16_t* state_out;          // [B,D,4] or NULL (may alias state_in)
  bf16_t* y;                  // [B,T,D] contiguous
  int col0;                   // first column of this segment in the fused projection row
  int D;
};

struct ProParams {
  const bf16_t* proj;         // [B*T, ld]
  long long ld;
  ConvSeg seg[3];
  int col_a, col_b;           // columns of the a / b projections (H each)
  const float* A_log; const float* dt_bias;
  float* g; bf16_t* beta;     // [B,T,H]
  int B, T, H, conv_blocks, apply_silu;
};


template <int FC_TCH>      // >= 3 (see above)
__global__ __launch_bounds__(256) void gdn_prologue_kernel(ProParams p) {
  if ((int)blockIdx.x >= p.conv_blocks) {
    // ---- gate math: beta = sigmoid(b) ; g = -exp(A_log) softplus(a + dt_bias) (std:1293-1294) ---------
    const long long n = (long long)p.B * p.T * p.H;
    const long long i0 = ((long long)blockIdx.x - p.conv_blocks) * blockDim.x + threadIdx.x;
    const long long step = ((long long)gridDim.x - p.conv_blocks) * blockDim.x;
    for (long long i = i0; i < n; i += step) {
      const int h = (int)(i % p.H);
      const long long row = i / p.H;
      const float av = bf2f(p.proj[row * p.ld + p.col_a + h]) + p.dt_bias[h];
      const float bv = bf2f(p.proj[row * p.ld + p.col_b + h]);
      const float sp = av > 20.f ? av : log1pf(expf(av));
      p.g[i] = -expf(p.A_log[h]) * sp;
      p.beta[i] = f2bf(sigmoid_exact_(bv));
    }
    return;
  }
  // ---- short convs: thread = 8 channels x FC_TCH tokens; chunk 0 owns the state of its channels -------
  const int Dtot = p.seg[0].D + p.seg[1].D + p.seg[2].D;
  const int DG = Dtot / 8;
  const int NCH = (p.T + FC_TCH - 1) / FC_TCH;
  const long long total = (long long)p.B * NCH * DG;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)p.conv_blocks * blockDim.x) {
    const int dgt = (int)(idx % DG);
    const int ch = (int)((idx / DG) % NCH);
    const int b = (int)(idx / ((long long)DG * NCH));
    int dall = dgt * 8;
    const int s = dall < p.seg[0].D ? 0 : (dall < p.seg[0].D + p.seg[1].D ? 1 : 2);
    const ConvSeg sg = s == 0 ? p.seg[0] : (s == 1 ? p.seg[1] : p.seg[2]);
    const int d0 = dall - (s == 0 ? 0 : (s == 1 ? p.seg[0].D : p.seg[0].D + p.seg[1].D));
    const int t0 = ch * FC_TCH;
    const int T = p.T;
    const bf16_t* xb = p.proj + (long long)b * T * p.ld + sg.col0 + d0;     // row t at xb + t*ld

    // issue every load first (branch-free): taps, up to 3 halo rows, the chunk's rows, the state
    const u32x4* wp = (const u32x4*)(sg.w + (size_t)d0 * FC_W);
    const u32x4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
    u32x4 xr[FC_TCH + 3];
#pragma unroll
    for (int k = 0; k < FC_TCH + 3; ++k) {
      int tt = t0 - 3 + k;
      tt = tt < 0 ? 0 : (tt > T - 1 ? T - 1 : tt);
      xr[k] = *(const u32x4*)(xb + (long long)tt * p.ld);
    }
    u32x4 s0 = u32x4{0u, 0u, 0u, 0u}, s1 = s0, s2 = s0, s3 = s0;
    if (ch == 0 && sg.state_in != nullptr) {
      const u32x4* sp = (const u32x4*)(sg.state_in + ((size_t)b * sg.D + d0) * FC_W);
      s0 = sp[0]; s1 = sp[1]; s2 = sp[2]; s3 = sp[3];
    }
    float wf[8][FC_W], st[8][FC_W];
    {
      const unsigned int ww[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
      const unsigned int ss[16] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w, s3.x, s3.y, s3.z, s3.w};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        wf[c][0] = bflo(ww[2 * c]); wf[c][1] = bfhi(ww[2 * c]); wf[c][2] = bflo(ww[2 * c + 1]); wf[c][3] = bfhi(ww[2 * c + 1]);
        st[c][0] = bflo(ss[2 * c]); st[c][1] = bfhi(ss[2 * c]); st[c][2] = bflo(ss[2 * c + 1]); st[c][3] = bfhi(ss[2 * c + 1]);
      }
    }
    float win[3][8];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      unpack8(xr[k], win[k]);
      if (ch == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) win[k][c] = st[c][k + 1];    // times -3,-2,-1 = state[...,1..3]
      }
    }
    bf16_t* yb = sg.y + ((size_t)b * T) * sg.D + d0;
#pragma unroll
    for (int k = 0; k < FC_TCH; ++k) {
      const int t = t0 + k;
      float cur[8], out[8];
      unpack8(xr[k + 3], cur);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float a = wf[c][0] * win[0][c];
        a = fmaf(wf[c][1], win[1][c], a);
        a = fmaf(wf[c][2], win[2][c], a);
        a = fmaf(wf[c][3], cur[c], a);
        if (p.apply_silu) a = a * sigmoidf_(a);
        out[c] = a;
        win[0][c] = win[1][c]; win[1][c] = win[2][c]; win[2][c] = cur[c];
      }
      if (t < T) *(u32x4*)(yb + (size_t)t * sg.D) = pack8(out);
    }
    if (ch == 0 && sg.state_out != nullptr) {
      // new_state[c][j] = ext[T + j], ext = [state(4), x(T)]
      float ns[8][FC_W];
#pragma unroll
      for (int j = 0; j < FC_W; ++j) {
        const int e = T + j;
        if (e >= FC_W) {
          float xv[8];
          unpack8(*(const u32x4*)(xb + (long long)(e - FC_W) * p.ld), xv);
#pragma unroll
          for (int c = 0; c < 8; ++c) ns[c][j] = xv[c];
        } else {
#pragma unroll
          for (int e2 = 0; e2 < FC_W; ++e2)
            if (e == e2) {
#pragma unroll
              for (int c = 0; c < 8; ++c) ns[c][j] = st[c][e2];
            }
        }
      }
      u32x4* op = (u32x4*)(sg.state_out + ((size_t)b * sg.D + d0) * FC_W);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        op[i] = u32x4{pack2bf(ns[2 * i][0], ns[2 * i][1]), pack2bf(ns[2 * i][2], ns[2 * i][3]),
                      pack2bf(ns[2 * i + 1][0], ns[2 * i + 1][1]), pack2bf(ns[2 * i + 1][2], ns[2 * i + 1][3])};
    }
  }
}

// gated RMSNorm, rows of 256; gate read in place: element (token, head, c) at gate + token*gate_ld + head*256 + c
__global__ __launch_bounds__(256) void rmsnorm_gate_strided_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ gate, long long gate_ld, int H,
    const bf16_t* __restrict__ weight, bf16_t* __restrict__ y, int rows, float eps) {
  const int lane32 = threadIdx.x & 31;
  const long long row0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const long long stride = ((long long)gridDim.x * blockDim.x) >> 5;
  float wf[8];
  unpack8(*(const u32x4*)(weight + lane32 * 8), wf);
  for (long long r = row0; r < rows; r += stride) {
    int h;
    const long long tok = divmod_idx(r, H, h);
    float xf[8], gf[8], o8[8];
    unpack8(*(const u32x4*)(x + r * 256 + lane32 * 8), xf);
    unpack8(*(const u32x4*)(gate + tok * gate_ld + h * 256 + lane32 * 8), gf);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss = fmaf(xf[i], xf[i], ss);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / 256.0f) + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) o8[i] = xf[i] * rstd * wf[i] * gf[i] * sigmoidf_(gf[i]);
    store_out16(y + r * 256 + lane32 * 8, pack8(o8));
  }
}

// M-RoPE in place; q element (b,t,h,c) at q + (b*T+t)*q_ld + h*d + c, same for k
__global__ __launch_bounds__(256) void mrope_strided_kernel(
    bf16_t* q, bf16_t* k, long long q_ld, long long k_ld, const bf16_t* __restrict__ cosp,
    const bf16_t* __restrict__ sinp, int B, int T, int Hq, int Hkv, int d, int s0, int s1) {
  const int half = d / 2;
  const int CG = half / 8;
  const int HT = Hq + Hkv;
  const long long total = (long long)B * T * HT * CG;
  const long long plane = (long long)B * T * d;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % CG);
    const int h = (int)((idx / CG) % HT);
    const long long bt = idx / ((long long)CG * HT);
    const int c0 = cg * 8;
    bf16_t* base = h < Hq ? q + bt * q_ld + (long long)h * d : k + bt * k_ld + (long long)(h - Hq) * d;
    // channel block c0..c0+7 lies inside one section (sections are multiples of 8 for d = 128: 16|24|24)
    const int sec = c0 < s0 ? 0 : (c0 < s0 + s1 ? 1 : 2);
    const long long off = sec * plane + bt * d;
    float x1[8], x2[8], c1[8], n1[8], c2[8], n2[8], o1[8], o2[8];
    unpack8(*(const u32x4*)(base + c0), x1);
    unpack8(*(const u32x4*)(base + c0 + half), x2);
    unpack8(*(const u32x4*)(cosp + off + c0), c1);
    unpack8(*(const u32x4*)(sinp + off + c0), n1);
    unpack8(*(const u32x4*)(cosp + off + c0 + half), c2);
    unpack8(*(const u32x4*)(sinp + off + c0 + half), n2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o1[i] = bf_round(bf_round(x1[i] * c1[i]) + bf_round(-x2[i] * n1[i]));
      o2[i] = bf_round(bf_round(x2[i] * c2[i]) + bf_round(x1[i] * n2[i]));
    }
    *(u32x4*)(base + c0) = pack8(o1);
    *(u32x4*)(base + c0 + half) = pack8(o2);
  }
}

// (residual add +) RMSNorm, one 256-thread workgroup per row, N % 8 == 0, N <= 8192.
//   h = bf16(x + residual)            (written to h_out when residual != NULL)
//   y = bf16(weight * bf16(h * rsqrt(mean(h^2) + eps)))        (Qwen2RMSNorm rounding points)
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ residual, const bf16_t* __restrict__ weight,
    bf16_t* __restrict__ y, bf16_t* __restrict__ h_out, int N, float eps) {
  __shared__ float s_part[4];
  const long long row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nvec = N / 8;
  float hv[4][8];
  float ss = 0.f;
  // the weight vectors are requested with the row (behind the barrier they were a second memory round trip per launch,
  // and the decoder step issues this kernel 72 times)
  u32x4 wraw[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = tid + it * 256;
    if (v < nvec) wraw[it] = *(const u32x4*)(weight + v * 8);
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = tid + it * 256;
    if (v < nvec) {
      unpack8(*(const u32x4*)(x + row * N + v * 8), hv[it]);
      if (residual != nullptr) {
        float rv[8];
        unpack8(*(const u32x4*)(residual + row * N + v * 8), rv);
#pragma unroll
        for (int i = 0; i < 8; ++i) hv[it][i] = bf_round(hv[it][i] + rv[i]);
        store_out16(h_out + row * N + v * 8, pack8(hv[it]));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) ss = fmaf(hv[it][i], hv[it][i], ss);
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) s_part[wave] = ss;
  __syncthreads();
  const float tot = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  const float rstd = rsqrtf(tot / (float)N + eps);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = tid + it * 256;
    if (v < nvec) {
      float wv[8], o8[8];
      unpack8(wraw[it], wv);
#pragma unroll
      for (int i = 0; i < 8; ++i) o8[i] = wv[i] * bf_round(hv[it][i] * rstd);
      store_out16(y + row * N + v * 8, pack8(o8));
    }
  }
}

// y[r, i] = bf16( bf16(silu(a)) * b ), a = gu[r, i], b = gu[r, I + i]  (two halves of a fused gate|up GEMM)
__global__ __launch_bounds__(256) void silu_mul_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ y,
                                                      long long rows, int I) {
  const int nvec = I / 8;
  const long long total = rows * nvec;
  // up to 32 MB of output (the 256-token step: 5.6 MB) is written through (store_out16: -0.1 us in step); the 90 MB of a
  // 4096-token call back up behind the memory side when written through (42.8 -> 45.7 us in call): plain stores there
  const bool write_through = total * 16 <= (32ll << 20);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int v;
    const long long r = divmod_idx(idx, nvec, v);
    float a[8], b[8], o[8];
    unpack8(*(const u32x4*)(gu + r * 2 * I + v * 8), a);
    unpack8(*(const u32x4*)(gu + r * 2 * I + I + v * 8), b);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bf_round(a[i] * sigmoidf_(a[i])) * b[i];
    if (write_through) store_out16(y + r * I + v * 8, pack8(o));
    else *(u32x4*)(y + r * I + v * 8) = pack8(o);
  }
}

static inline int grid_cap(long long items, int block = 256, int cap = 2048) {
  long long g = (items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace ivl

using namespace ivl;

extern "C" int ivl_gdn_prologue_fwd(const void* proj, int64_t ld, int col_q, int col_k, int col_v, int col_a, int col_b,
                                    const void* w_q, const void* w_k, const void* w_v,
                                    const void* sq_in, const void* sk_in, const void* sv_in,
                                    void* sq_out, void* sk_out, void* sv_out,
                                    const float* A_log, const float* dt_bias,
                                    void* q, void* k, void* v, float* g, void* beta,
                                    int B, int T, int H, int Dq, int Dk, int Dv, int W, int apply_silu, void* stream) {
  IVL_REQUIRE(proj && w_q && w_k && w_v && A_log && dt_bias && q && k && v && g && beta, IVL_ERR_INVALID_ARG,
              "ivl_gdn_prologue_fwd: NULL pointer");
  IVL_REQUIRE(B > 0 && T > 0 && H > 0 && Dq > 0 && Dk > 0 && Dv > 0, IVL_ERR_INVALID_ARG, "ivl_gdn_prologue_fwd: bad sizes");
  IVL_REQUIRE(W == FC_W, IVL_ERR_UNSUPPORTED, "ivl_gdn_prologue_fwd: kernel size %d unsupported (built for 4)", W);
  IVL_REQUIRE(Dq % 8 == 0 && Dk % 8 == 0 && Dv % 8 == 0 && ld % 8 == 0 && col_q % 8 == 0 && col_k % 8 == 0 && col_v % 8 == 0,
              IVL_ERR_UNSUPPORTED, "ivl_gdn_prologue_fwd: channel counts / offsets / row stride must be multiples of 8");
  ProParams p;
  p.proj = (const bf16_t*)proj; p.ld = ld;
  p.seg[0] = ConvSeg{(const bf16_t*)w_q, (const bf16_t*)sq_in, (bf16_t*)sq_out, (bf16_t*)q, col_q, Dq};
  p.seg[1] = ConvSeg{(const bf16_t*)w_k, (const bf16_t*)sk_in, (bf16_t*)sk_out, (bf16_t*)k, col_k, Dk};
  p.seg[2] = ConvSeg{(const bf16_t*)w_v, (const bf16_t*)sv_in, (bf16_t*)sv_out, (bf16_t*)v, col_v, Dv};
  p.col_a = col_a; p.col_b = col_b; p.A_log = A_log; p.dt_bias = dt_bias; p.g = g; p.beta = (bf16_t*)beta;
  p.B = B; p.T = T; p.H = H; p.apply_silu = apply_silu;
  const int tch = T <= 1024 ? 4 : 8;
  const long long conv_items = (long long)B * ((T + tch - 1) / tch) * ((Dq + Dk + Dv) / 8);
  p.conv_blocks = grid_cap(conv_items);
  const int gate_blocks = grid_cap((long long)B * T * H, 256, 64);
  if (tch == 4) hipLaunchKernelGGL(gdn_prologue_kernel<4>, dim3(p.conv_blocks + gate_blocks), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gdn_prologue_kernel<8>, dim3(p.conv_blocks + gate_blocks), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("ivl_gdn_prologue_fwd");
}

extern "C" int ivl_rmsnorm_swish_gate_strided_fwd(const void* x, const void* gate, int64_t gate_ld, int H,
                                                  const void* weight, void* y, int rows, int N, float eps, void* stream) {
  IVL_REQUIRE(x && gate && weight && y, IVL_ERR_INVALID_ARG, "ivl_rmsnorm_swish_gate_strided_fwd: NULL pointer");
  IVL_REQUIRE(rows > 0 && H > 0 && rows % H == 0, IVL_ERR_INVALID_ARG, "ivl_rmsnorm_swish_gate_strided_fwd: rows=%d H=%d", rows, H);
  IVL_REQUIRE(N == 256 && gate_ld % 8 == 0, IVL_ERR_UNSUPPORTED, "ivl_rmsnorm_swish_gate_strided_fwd: N=%d (built for 256)", N);
  hipLaunchKernelGGL(rmsnorm_gate_strided_kernel, dim3(grid_cap((long long)rows * 32)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)gate, (long long)gate_ld, H, (const bf16_t*)weight, (bf16_t*)y, rows, eps);
  return check_launch("ivl_rmsnorm_swish_gate_strided_fwd");
}

extern "C" int ivl_mrope_strided_fwd(void* q, void* k, int64_t q_ld, int64_t k_ld, const void* cos, const void* sin,
                                     int B, int T, int Hq, int Hkv, int d, int s0, int s1, int s2, void* stream) {
  IVL_REQUIRE(q && k && cos && sin, IVL_ERR_INVALID_ARG, "ivl_mrope_strided_fwd: NULL pointer");
  IVL_REQUIRE(B > 0 && T > 0 && Hq > 0 && Hkv > 0, IVL_ERR_INVALID_ARG, "ivl_mrope_strided_fwd: bad sizes");
  IVL_REQUIRE(d % 16 == 0 && s0 + s1 + s2 == d / 2 && s0 % 8 == 0 && s1 % 8 == 0 && q_ld % 8 == 0 && k_ld % 8 == 0,
              IVL_ERR_UNSUPPORTED, "ivl_mrope_strided_fwd: need d%%16==0, sections multiple of 8 summing to d/2");
  const long long items = (long long)B * T * (Hq + Hkv) * (d / 16);
  hipLaunchKernelGGL(mrope_strided_kernel, dim3(grid_cap(items)), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)q, (bf16_t*)k, (long long)q_ld, (long long)k_ld, (const bf16_t*)cos, (const bf16_t*)sin,
                     B, T, Hq, Hkv, d, s0, s1);
  return check_launch("ivl_mrope_strided_fwd");
}

extern "C" int ivl_add_rmsnorm_fwd(const void* x, const void* residual, const void* weight, void* y, void* h_out,
                                   int rows, int N, float eps, void* stream) {
  IVL_REQUIRE(x && weight && y, IVL_ERR_INVALID_ARG, "ivl_add_rmsnorm_fwd: NULL pointer");
  IVL_REQUIRE(residual == nullptr || h_out != nullptr, IVL_ERR_INVALID_ARG, "ivl_add_rmsnorm_fwd: residual needs h_out");
  IVL_REQUIRE(rows > 0, IVL_ERR_INVALID_ARG, "ivl_add_rmsnorm_fwd: rows=%d", rows);
  IVL_REQUIRE(N > 0 && N % 8 == 0 && N <= 8192, IVL_ERR_UNSUPPORTED, "ivl_add_rmsnorm_fwd: N=%d (need N%%8==0, N<=8192)", N);
  hipLaunchKernelGGL(add_rmsnorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)residual, (const bf16_t*)weight, (bf16_t*)y, (bf16_t*)h_out, N, eps);
  return check_launch("ivl_add_rmsnorm_fwd");
}

extern "C" int ivl_silu_mul_fwd(const void* gate_up, void* y, int64_t rows, int I, void* stream) {
  IVL_REQUIRE(gate_up && y, IVL_ERR_INVALID_ARG, "ivl_silu_mul_fwd: NULL pointer");
  IVL_REQUIRE(rows > 0 && I > 0 && I % 8 == 0, IVL_ERR_INVALID_ARG, "ivl_silu_mul_fwd: rows=%lld I=%d", (long long)rows, I);
  hipLaunchKernelGGL(silu_mul_kernel, dim3(grid_cap((long long)rows * (I / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)gate_up, (bf16_t*)y, (long long)rows, I);
  return check_launch("ivl_silu_mul_fwd");
}
