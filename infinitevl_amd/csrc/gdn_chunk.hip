// Gated DeltaNet, chunkwise form (chunk C = 64 tokens) for gfx950, K = 128, V = 256.
//
// Two launches (the reference uses six and materialises the per-chunk state h[B,NT,H,K,V] in HBM):
//
//  (1) gdn_chunk_prepare_kernel -- chunk-PARALLEL, one workgroup per (chunk, batch*head):
//        l2norm(q), l2norm(k) -> bf16;  gamma = cumsum(g);  L = tril(bf16(beta k) k^T, -1) (MFMA bf16);
//        Tw = (I+L)^-1 in fp32: 16x16 diagonal blocks by forward substitution, off-diagonal blocks by
//        block elimination on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32); Tu = Tw * e^{gamma_i-gamma_j};
//        w = bf16(Tw) bf16(beta k),  u = bf16(Tu) bf16(beta v)  (MFMA bf16);
//        A = tril((q k^T) * Gamma) -> bf16.
//      It leaves in the workspace, per chunk, exactly the operands the serial pass needs, already decayed
//      and laid out K-contiguous ("NT" operands) so that the scan issues plain 16-byte fragment loads:
//        Wg[64][128] = bf16(w * e^gamma)   Qh[64][128] = q_hat     KdT[128][64] = bf16(k_hat * e^{gl-gamma})^T
//        UT[256][64] = u^T                 Aqk[64][64]             eg[64] = e^gamma, egl = e^{gamma_last}
//  (2) gdn_chunk_scan_kernel -- SERIAL over chunks, one workgroup per (32-column slab of V, batch*head):
//        the fp32 state slab S[128x32] lives in MFMA accumulators (wave w owns rows 32w..32w+31);
//        per chunk:  v_new = u - Wg S ;  o = scale((Qh S) * e^gamma + Aqk v_new) ;  S = egl S + KdT v_new
//        with bf16 MFMA operands / fp32 accumulation at the reference's rounding points.  State and
//        v_new cross waves through two small LDS tiles (S^T bf16, v_new^T bf16); nothing else leaves the CU.
//
// All matrix products are "NT" products on v_mfma_f32_32x32x16_bf16: lane l holds A[i=l&31][k=8(l>>5)..+7]
// and B^T[j=l&31][k=8(l>>5)..+7] (16 contiguous bytes each), C[i=(r&3)+8(r>>2)+4(l>>5)][j=l&31].
#include "ivl_common.h"

namespace ivl {

typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int GC = 64;        // chunk length
constexpr int GK = 128;       // key head dim
constexpr int GV = 256;       // value head dim
constexpr int G_BV = 32;      // state columns per scan workgroup
constexpr int G_SEG_CHUNKS = 64;   // chunks per workspace segment (4096 tokens)

// workspace record per (batch*head, chunk): byte offsets
constexpr size_t WS_WG = 0;                       // bf16 [64][128]
constexpr size_t WS_QH = WS_WG + GC * GK * 2;     // bf16 [64][128]
constexpr size_t WS_KDT = WS_QH + GC * GK * 2;    // bf16 [128][64]
constexpr size_t WS_UT = WS_KDT + GK * GC * 2;    // bf16 [256][64]
constexpr size_t WS_AQK = WS_UT + GV * GC * 2;    // bf16 [64][64]
constexpr size_t WS_EG = WS_AQK + GC * GC * 2;    // f32  [64]
constexpr size_t WS_EGL = WS_EG + GC * 4;         // f32  [1] (+pad)
constexpr size_t WS_STRIDE = WS_EG + 1024;        // 91136 (the e^gamma block is fetched as one 1 KB piece)

__device__ __forceinline__ mfma_bf16x8 mf(u32x4 v) {
  mfma_bf16x8 r;
  __builtin_memcpy(&r, &v, 16);
  return r;
}
__device__ __forceinline__ int crow32(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ==================================================================================================
// (1) chunk-parallel pre-pass
// ==================================================================================================
// LDS map (bytes).  Row strides are padded so 16-byte fragment reads of 32 consecutive rows spread banks.
constexpr int P_LDK = 136;                         // bf16 elements per row of kh/qh/kb/stage (272 B)
constexpr int P_LDV = 264;                         // bf16 elements per row of vb (528 B)
constexpr int P_LDF = 68;                          // f32 elements per row of L / T (16-byte aligned rows)
constexpr int P_LDY = 33;
constexpr int P_KH = 0;                            // k_hat            [64][136] bf16
constexpr int P_QH = P_KH + GC * P_LDK * 2;        // q_hat            [64][136]
constexpr int P_VB = 0;                            // bf16(beta v)     [64][264] row-major, written over k_hat/q_hat
                                                   //   once they are dead (B operands via the LDS transpose read)
constexpr int P_KB = P_QH + GC * P_LDK * 2;        // bf16(beta k_hat) [64][136]; later: Wg output staging
constexpr int P_L = P_KB + GC * P_LDK * 2;         // L, inverted IN PLACE to T = (I+L)^-1   [64][68] f32
constexpr int P_Y = P_L + GC * P_LDF * 4;
constexpr int P_SM = P_Y + 32 * P_LDY * 4;         // gam[64], beta[64], eg[64], dec[64]
constexpr int P_BYTES = P_SM + 4 * GC * 4;         // 74,880: two workgroups per CU
static_assert(GC * P_LDV * 2 <= 2 * GC * P_LDK * 2, "vb must fit in the k_hat/q_hat region");
static_assert(2 * P_BYTES <= 160 * 1024, "pre-pass LDS budget (2 workgroups per CU)");

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// 32x32x16 MFMA B fragment taken from a ROW-MAJOR LDS tile X[k][n] (row stride ld elements) with the gfx950
// transpose read: lane needs X[k0 + 8(lane>>5) + e][n0 + (lane&31)], e = 0..7.  Each 16-lane group reads a
// 4x16 block (lane i supplies the address of row i>>2, columns 4(i&3)..+3 and receives column i).
__device__ __forceinline__ u32x4 bfrag_tr(const bf16_t* X, int ld, int k0, int n0, int lane) {
  const int i = lane & 15, gq = lane >> 4;
  const bf16_t* p = X + (k0 + 8 * (gq >> 1) + (i >> 2)) * ld + n0 + 16 * (gq & 1) + 4 * (i & 3);
  const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * ld));
  u32x2 w0, w1;
  __builtin_memcpy(&w0, &a0, 8);
  __builtin_memcpy(&w1, &a1, 8);
  return u32x4{w0.x, w0.y, w1.x, w1.y};
}

// fp32 16x16 tile product on v_mfma_f32_16x16x4_f32 from LDS operands: acc += A[a_r0.., a_c0..] * B[b_r0.., b_c0..]
__device__ __forceinline__ f32x4 tile16_f32(f32x4 acc, const float* A, int lda, int a_r0, int a_c0,
                                            const float* Bm, int ldb, int b_r0, int b_c0, int K, int lane) {
  const int i = lane & 15, kq = lane >> 4;
  for (int kk = 0; kk < K; kk += 4) {
    const float a = A[(a_r0 + i) * lda + a_c0 + kk + kq];
    const float b = Bm[(b_r0 + kk + kq) * ldb + b_c0 + i];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  return acc;
}
__device__ __forceinline__ void store16_f32(float* Cm, int ldc, int r0, int c0, f32x4 acc, float sign, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) Cm[(r0 + 4 * g + r) * ldc + c0 + j] = sign * acc[r];
}

__global__ __launch_bounds__(256, 2) void gdn_chunk_prepare_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const float* __restrict__ g, const bf16_t* __restrict__ beta, unsigned char* __restrict__ ws,
    int T, int H, int t_seg0, int nt_seg, int l2norm, long long* trace) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* s_kh = (bf16_t*)(smem + P_KH);
  bf16_t* s_qh = (bf16_t*)(smem + P_QH);
  bf16_t* s_kb = (bf16_t*)(smem + P_KB);
  bf16_t* s_vb = (bf16_t*)(smem + P_VB);
  float* s_L = (float*)(smem + P_L);
  float* s_T = s_L;                          // inverted in place
  float* s_Y = (float*)(smem + P_Y);
  float* s_gam = (float*)(smem + P_SM);
  float* s_beta = s_gam + GC;
  float* s_eg = s_beta + GC;
  float* s_dec = s_eg + GC;
  bf16_t* s_stage = (bf16_t*)(smem + P_KB);

  trace_stamp(trace, 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int ci = blockIdx.x;                  // chunk within the segment
  const int bh = blockIdx.y;
  const int b = bh / H, h = bh % H;
  const int t0 = t_seg0 + ci * GC;            // first token of the chunk
  const int nvalid = min(GC, T - t0);
  unsigned char* rec = ws + ((size_t)bh * nt_seg + ci) * WS_STRIDE;

  // ---- issue every global load of the chunk up front (clamped rows, zeroed later): 16 x 16 B per thread
  //      in flight at once instead of one HBM round trip per conditional row ------------------------
  const int oct = tid & 15, rg = tid >> 4;         // q,k: 16 column octets x 16 row groups of 4 rows
  const int voct = tid & 31, vrg = tid >> 5;       // v  : 32 column octets x  8 row groups of 8 rows
  u32x4 kraw[4], qraw[4], vraw[8];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(4 * rg + r, nvalid - 1);
    const size_t tok = ((size_t)b * T + t0 + row) * H + h;
    kraw[r] = *(const u32x4*)(k + tok * GK + 8 * oct);
    qraw[r] = *(const u32x4*)(q + tok * GK + 8 * oct);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = min(8 * vrg + r, nvalid - 1);
    const size_t tok = ((size_t)b * T + t0 + row) * H + h;
    vraw[r] = *(const u32x4*)(v + tok * GV + 8 * voct);
  }
  // ---- S0: g, beta; chunk-local inclusive cumsum (wave 0) -------------------------------------
  if (wave == 0) {
    float gv = 0.f, bv = 0.f;
    {
      const size_t tok = ((size_t)b * T + t0 + min(lane, nvalid - 1)) * H + h;
      const float g_ld = g[tok];
      const float b_ld = bf2f(beta[tok]);
      if (lane < nvalid) { gv = g_ld; bv = b_ld; }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float up = __shfl_up(gv, o, 64);
      if (lane >= o) gv += up;
    }
    const float gl = __shfl(gv, nvalid - 1, 64);     // gamma at the last VALID token
    s_gam[lane] = gv;
    s_beta[lane] = bv;
    const float e = __expf(gv);
    s_eg[lane] = e;
    s_dec[lane] = __expf(gl - gv);                   // e^{gamma_last - gamma_t}
    ((float*)(rec + WS_EG))[lane] = e;
    if (lane == 0) *(float*)(rec + WS_EGL) = __expf(gl);
  }
  trace_stamp(trace, 8);
  trace_stamp(trace, 9);

  // ---- S1a (independent of beta/gamma, overlaps S0): l2norm -> k_hat, q_hat (bf16) to LDS -------------
  float kf[4][8];
  {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * rg + r;
      const bool ok = row < nvalid;
      float qf[8];
      const u32x4 kv = kraw[r], qv = qraw[r];
      kf[r][0] = bflo(kv.x); kf[r][1] = bfhi(kv.x); kf[r][2] = bflo(kv.y); kf[r][3] = bfhi(kv.y);
      kf[r][4] = bflo(kv.z); kf[r][5] = bfhi(kv.z); kf[r][6] = bflo(kv.w); kf[r][7] = bfhi(kv.w);
      if (r == 0 && __float_as_uint(kf[0][0]) != 0x7fc12345u) trace_stamp(trace, 10);
      qf[0] = bflo(qv.x); qf[1] = bfhi(qv.x); qf[2] = bflo(qv.y); qf[3] = bfhi(qv.y);
      qf[4] = bflo(qv.z); qf[5] = bfhi(qv.z); qf[6] = bflo(qv.w); qf[7] = bfhi(qv.w);
#pragma unroll
      for (int c = 0; c < 8; ++c) { kf[r][c] = ok ? kf[r][c] : 0.f; qf[c] = ok ? qf[c] : 0.f; }
      if (l2norm) {
        float ks = 0.f, qs = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) { ks = fmaf(kf[r][c], kf[r][c], ks); qs = fmaf(qf[c], qf[c], qs); }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { ks += __shfl_xor(ks, o, 64); qs += __shfl_xor(qs, o, 64); }
        const float rk = 1.0f / sqrtf(ks + 1e-6f), rq = 1.0f / sqrtf(qs + 1e-6f);
#pragma unroll
        for (int c = 0; c < 8; ++c) { kf[r][c] = bf_round(kf[r][c] * rk); qf[c] = qf[c] * rq; }
      }
      *(u32x4*)(s_kh + row * P_LDK + 8 * oct) =
          u32x4{pack2bf(kf[r][0], kf[r][1]), pack2bf(kf[r][2], kf[r][3]), pack2bf(kf[r][4], kf[r][5]), pack2bf(kf[r][6], kf[r][7])};
      *(u32x4*)(s_qh + row * P_LDK + 8 * oct) =
          u32x4{pack2bf(qf[0], qf[1]), pack2bf(qf[2], qf[3]), pack2bf(qf[4], qf[5]), pack2bf(qf[6], qf[7])};
    }
  }
  trace_stamp(trace, 11);
  __syncthreads();

  trace_stamp(trace, 1);
  // ---- S1b: bf16(beta k_hat) row-major to LDS (beta v follows once k_hat/q_hat are dead) -------------
  {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * rg + r;
      const float bt = s_beta[row];
      *(u32x4*)(s_kb + row * P_LDK + 8 * oct) =
          u32x4{pack2bf(kf[r][0] * bt, kf[r][1] * bt), pack2bf(kf[r][2] * bt, kf[r][3] * bt),
                pack2bf(kf[r][4] * bt, kf[r][5] * bt), pack2bf(kf[r][6] * bt, kf[r][7] * bt)};
    }
  }
  __syncthreads();

  trace_stamp(trace, 2);
  // ---- S2: L = tril(kb kh^T, -1) -> s_L (fp32);  Aqk = tril((qh kh^T) * Gamma) -> global bf16 ------
  //          wave w -> 32x32 tile (mi = w>>1, ni = w&1); tile (0,1) lies above the diagonal.
  {
    const int mi = wave >> 1, ni = wave & 1;
    if (!(mi == 0 && ni == 1)) {
      f32x16 accL, accA;
#pragma unroll
      for (int r = 0; r < 16; ++r) { accL[r] = 0.f; accA[r] = 0.f; }
      const bf16_t* arow_kb = s_kb + (32 * mi + l31) * P_LDK + 8 * hi;
      const bf16_t* arow_qh = s_qh + (32 * mi + l31) * P_LDK + 8 * hi;
      const bf16_t* brow = s_kh + (32 * ni + l31) * P_LDK + 8 * hi;
#pragma unroll
      for (int ks = 0; ks < GK / 16; ++ks) {
        const u32x4 bfr = *(const u32x4*)(brow + 16 * ks);
        const u32x4 a1 = *(const u32x4*)(arow_kb + 16 * ks);
        const u32x4 a2 = *(const u32x4*)(arow_qh + 16 * ks);
        accL = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(a1), mf(bfr), accL, 0, 0, 0);
        accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(a2), mf(bfr), accA, 0, 0, 0);
      }
      const int j = 32 * ni + l31;
      const float gj = s_gam[j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 32 * mi + crow32(r, hi);
        s_L[i * P_LDF + j] = i > j ? accL[r] : 0.f;
        const float a = i >= j ? accA[r] * __expf(s_gam[i] - gj) : 0.f;
        ((bf16_t*)(rec + WS_AQK))[i * GC + j] = f2bf(a);
      }
    } else {
      // zero the strictly-upper tile of L and Aqk
      const int j = 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = crow32(r, hi);
        s_L[i * P_LDF + j] = 0.f;
        ((bf16_t*)(rec + WS_AQK))[i * GC + j] = 0;
      }
    }
  }
  __syncthreads();

  trace_stamp(trace, 3);
  // ---- Qh and KdT leave now (their stores overlap the solve); afterwards k_hat/q_hat are dead -----------
  for (int idx = tid; idx < GC * (GK / 8); idx += 256) {
    const int row = idx >> 4, ch = idx & 15;
    *(u32x4*)(rec + WS_QH + ((size_t)row * GK + 8 * ch) * 2) = *(const u32x4*)(s_qh + row * P_LDK + 8 * ch);
  }
  // KdT[kidx][time] = bf16(k_hat[time][kidx] * e^{gamma_last - gamma_time}): thread = (kidx, 32-token half),
  // one 64-byte run per thread
  {
    const int c = tid & 127, half = tid >> 7;
    unsigned int pk[16];
#pragma unroll
    for (int t2 = 0; t2 < 16; ++t2) {
      const int t = 32 * half + 2 * t2;
      const float a0 = bf2f(s_kh[t * P_LDK + c]) * s_dec[t];
      const float a1 = bf2f(s_kh[(t + 1) * P_LDK + c]) * s_dec[t + 1];
      pk[t2] = pack2bf(a0, a1);
    }
    u32x4* dst = (u32x4*)(rec + WS_KDT + ((size_t)c * GC + 32 * half) * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = u32x4{pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3]};
  }
  // ---- S3: T = (I + L)^-1, in place ------------------------------------------------------------
  // (a) diagonal 16x16 blocks by forward substitution: wave w -> block w, lane c<16 -> column c.
  if (lane < 16) {
    const int r0 = 16 * wave, c = lane;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float s = (i == c) ? 1.f : 0.f;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj)
        if (jj < i) s = fmaf(-s_L[(r0 + i) * P_LDF + r0 + jj], x[jj], s);
      x[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s_T[(r0 + i) * P_LDF + r0 + c] = x[i];
  }
  __syncthreads();
  // bf16(beta v) row-major over the dead k_hat/q_hat region (read by S6, several barriers later)
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = 8 * vrg + r;
    const u32x4 vv = vraw[r];
    const float bt = s_beta[row];               // 0 for padded rows
    *(u32x4*)(s_vb + row * P_LDV + 8 * voct) =
        u32x4{pack2bf(bflo(vv.x) * bt, bfhi(vv.x) * bt), pack2bf(bflo(vv.y) * bt, bfhi(vv.y) * bt),
              pack2bf(bflo(vv.z) * bt, bfhi(vv.z) * bt), pack2bf(bflo(vv.w) * bt, bfhi(vv.w) * bt)};
  }
  // (b) 16->32: X21 = -X22 (L21 X11) for block pairs (0,1) [wave 0] and (2,3) [wave 1]
  if (wave < 2) {
    const int base = 32 * wave;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    acc = tile16_f32(acc, s_L, P_LDF, base + 16, base, s_T, P_LDF, base, base, 16, lane);
    store16_f32(s_Y, P_LDY, 16 * wave, 0, acc, 1.f, lane);
  }
  __syncthreads();
  if (wave < 2) {
    const int base = 32 * wave;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    acc = tile16_f32(acc, s_T, P_LDF, base + 16, base + 16, s_Y, P_LDY, 16 * wave, 0, 16, lane);
    store16_f32(s_T, P_LDF, base + 16, base, acc, -1.f, lane);
  }
  __syncthreads();
  // (c) 32->64: T21 = -T22 (L21 T11), 32x32 blocks; wave w -> 16x16 tile (w>>1, w&1)
  {
    const int ti = wave >> 1, tj = wave & 1;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    acc = tile16_f32(acc, s_L, P_LDF, 32 + 16 * ti, 0, s_T, P_LDF, 0, 16 * tj, 32, lane);
    store16_f32(s_Y, P_LDY, 16 * ti, 16 * tj, acc, 1.f, lane);
  }
  __syncthreads();
  {
    const int ti = wave >> 1, tj = wave & 1;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    acc = tile16_f32(acc, s_T, P_LDF, 32 + 16 * ti, 32, s_Y, P_LDY, 0, 16 * tj, 32, lane);
    store16_f32(s_T, P_LDF, 32 + 16 * ti, 16 * tj, acc, -1.f, lane);
  }
  __syncthreads();

  trace_stamp(trace, 4);
  trace_stamp(trace, 5);
  // ---- S5: w = bf16(Tw) kb  (64x64 . 64x128): wave w -> columns 32w..32w+31, both row tiles.
  //          A fragments are built straight from the fp32 T in LDS (rounded to bf16 in registers:
  //          the reference stores Aw/Au in bf16, wy_fast.py:341-343). ------------------------------
  u32x4 twf[2][4];          // bf16(Tw)[row 32mi + l31][16ks + 8hi .. +7]
  u32x4 tuf[2][4];          // bf16(Tw * e^{gamma_i - gamma_j}) same positions (Tu)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int i = 32 * mi + l31;
    const float gi = s_gam[i];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 t0 = *(const f32x4*)(s_T + i * P_LDF + 16 * ks + 8 * hi);
      const f32x4 t1 = *(const f32x4*)(s_T + i * P_LDF + 16 * ks + 8 * hi + 4);
      const f32x4 g0 = *(const f32x4*)(s_gam + 16 * ks + 8 * hi);
      const f32x4 g1 = *(const f32x4*)(s_gam + 16 * ks + 8 * hi + 4);
      float tw[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
      float gj[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
      float tu[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 16 * ks + 8 * hi + e;
        tu[e] = i >= j ? tw[e] * __expf(gi - gj[e]) : 0.f;
      }
      twf[mi][ks] = u32x4{pack2bf(tw[0], tw[1]), pack2bf(tw[2], tw[3]), pack2bf(tw[4], tw[5]), pack2bf(tw[6], tw[7])};
      tuf[mi][ks] = u32x4{pack2bf(tu[0], tu[1]), pack2bf(tu[2], tu[3]), pack2bf(tu[4], tu[5]), pack2bf(tu[6], tu[7])};
    }
  }
  {
    f32x16 acc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < GC / 16; ++ks) {
      const u32x4 bfr = bfrag_tr(s_kb, P_LDK, 16 * ks, 32 * wave, lane);     // (beta k)[time][col 32w + l31]
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(twf[mi][ks]), mf(bfr), acc[mi], 0, 0, 0);
    }
    __syncthreads();          // every wave has read its kb fragments before the region is reused for staging
    // Wg = bf16(bf16(w) * e^gamma_i) staged row-major in the (dead) kb region
    const int j = 32 * wave + l31;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 32 * mi + crow32(r, hi);
        s_stage[i * P_LDK + j] = f2bf(bf_round(acc[mi][r]) * s_eg[i]);
      }
  }
  __syncthreads();
  // coalesced copy-out of Wg (64 rows x 256 B)
  for (int idx = tid; idx < GC * (GK / 8); idx += 256) {
    const int row = idx >> 4, ch = idx & 15;
    *(u32x4*)(rec + WS_WG + ((size_t)row * GK + 8 * ch) * 2) = *(const u32x4*)(s_stage + row * P_LDK + 8 * ch);
  }

  trace_stamp(trace, 6);
  // ---- S6: u = Tu vb  (64x64 . 64x256): wave w -> columns 64w..64w+63 ; UT[col][time] to global --
  {
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
      f32x16 acc[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
      const int col = 64 * wave + 32 * nj + l31;
#pragma unroll
      for (int ks = 0; ks < GC / 16; ++ks) {
        const u32x4 bfr = bfrag_tr(s_vb, P_LDV, 16 * ks, 64 * wave + 32 * nj, lane);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(tuf[mi][ks]), mf(bfr), acc[mi], 0, 0, 0);
      }
      bf16_t* ut = (bf16_t*)(rec + WS_UT) + (size_t)col * GC;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          u32x2 w;
          w.x = pack2bf(acc[mi][4 * r4 + 0], acc[mi][4 * r4 + 1]);
          w.y = pack2bf(acc[mi][4 * r4 + 2], acc[mi][4 * r4 + 3]);
          *(u32x2*)(ut + 32 * mi + 8 * r4 + 4 * hi) = w;      // times 32mi + 8r4 + 4hi + 0..3
        }
    }
  }
  trace_stamp(trace, 7);
}

// ==================================================================================================
// (2) serial scan + output
// ==================================================================================================
constexpr int S_LDS = 128;     // bf16 per row of S^T  [32 cols][128 k]   (256 B, 16-byte pieces XOR-swizzled by the row)
constexpr int S_LDV = 64;      // bf16 per row of v_new^T [32 cols][64 t] (128 B, pieces swizzled like the KdT image)
constexpr int S_LDO = 40;      // bf16 per row of the output staging tile [64 t][32 cols] (80 B)

// LDS operand image of one chunk (bytes).  Every region is an image of the workspace record region with the
// 16-byte piece index XOR-swizzled by the row, so that the 16-byte MFMA fragment reads (32 consecutive rows,
// same piece) are bank-conflict free although the rows are 256 / 128 bytes apart.  The image is filled by
// LDS-DMA (global_load_lds_dwordx4): one wave instruction lands 1 KB lane-linearly (dest = M0 + 16*lane), so
// the swizzle is applied to each lane's SOURCE address; two images alternate so that the DMA of chunk c+1
// runs underneath the MFMAs of chunk c without occupying VGPRs or ds_write issue slots.
constexpr int OP_WG = 0;                   // [64][16 pieces]   piece' = c ^ (row & 15)
constexpr int OP_QH = OP_WG + 16384;       // [64][16 pieces]
constexpr int OP_KDT = OP_QH + 16384;      // [128][8 pieces]   piece' = c ^ ((row >> 1) & 7)
constexpr int OP_AQK = OP_KDT + 16384;     // [64][8 pieces]
constexpr int OP_UT = OP_AQK + 8192;       // [32][8 pieces]    (this workgroup's 32 columns of u^T)
constexpr int OP_EG = OP_UT + 4096;        // f32 e^gamma[64], e^gamma_last, pad (1 KB, linear)
constexpr int OP_BYTES = OP_EG + 1024;     // 62464
constexpr int SC_ST = 2 * OP_BYTES;                        // S^T   bf16 [32][136]
constexpr int SC_VN = SC_ST + G_BV * S_LDS * 2;            // v_new^T bf16 [32][72]
constexpr int SC_O = SC_VN + G_BV * S_LDV * 2;             // o tile bf16 [64][40]
constexpr int SC_BYTES = SC_O + GC * S_LDO * 2;            // 143360
static_assert(SC_BYTES <= 160 * 1024, "scan LDS budget");
static_assert(WS_STRIDE >= WS_EG + 1024, "the e^gamma block is fetched as one 1 KB piece");

__device__ __forceinline__ int swz16(int row, int c) { return (c ^ (row & 15)) << 4; }          // 256-byte rows
__device__ __forceinline__ int swz8(int row, int c) { return (c ^ ((row >> 1) & 7)) << 4; }     // 128-byte rows

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [lds_dst + 16*lane).  hipcc does not
// count this operation: completion is awaited with dma_wait_all() and published by the following barrier.
__device__ __forceinline__ void dma_piece(const unsigned char* gsrc, unsigned int lds_dst) {
  unsigned int keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst));
  // no "memory" clobber: volatile asm statements keep their order among themselves (barrier -> DMA issue -> wait),
  // and the LDS reads of the CURRENT image may be scheduled freely around the issue of the next one.
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// workgroup barrier that waits for this wave's LDS traffic only (a __syncthreads() would also be correct; the
// explicit form documents that no vector-memory drain is wanted here)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(256) void gdn_chunk_scan_kernel(
    const unsigned char* __restrict__ ws, bf16_t* __restrict__ o,
    const void* h0, int h0_dtype, void* ht, int ht_dtype,
    int T, int H, int t_seg0, int nt_seg, float scale, long long* trace) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  bf16_t* s_st = (bf16_t*)(smem + SC_ST);
  bf16_t* s_vn = (bf16_t*)(smem + SC_VN);
  bf16_t* s_o = (bf16_t*)(smem + SC_O);

  trace_stamp(trace, 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int l31 = lane & 31, hi = lane >> 5;
  // grid = (B*H, V/32): linear block id = bh + B*H*slab, so with B*H % 8 == 0 the 8 V-slabs of one head run on
  // the same XCD (id % 8) and share its L2 for the operands they all read (Wg/Qh/KdT/Aqk).
  const int v0 = blockIdx.y * G_BV;
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh % H;
  const unsigned int lds0 = __builtin_amdgcn_readfirstlane((unsigned int)(size_t)smem);

  // ---- operand staging.  Piece p (1 KB) of the image <-> image bytes [1024 p, +1024).  Wave w issues pieces
  //      4i + w, i = 0..14 (i < 4: Wg, < 8: Qh, < 12: KdT, < 14: Aqk, 14: u^T slab); wave 0 also the e^gamma block.
  //      Per-lane source offsets inside the record are chunk-invariant: computed once. ----------------------
  unsigned int src_off[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) {
    if (i < 8) {                       // 256-byte rows, 4 rows per piece
      const int q = (4 * (i & 3) + wave), row = 4 * q + (lane >> 4), c = (lane & 15) ^ (row & 15);
      src_off[i] = (unsigned int)((i < 4 ? WS_WG : WS_QH) + row * 256 + c * 16);
    } else if (i < 14) {               // 128-byte rows, 8 rows per piece
      const int q = (i < 12 ? 4 * (i - 8) : 4 * (i - 12)) + wave, row = 8 * q + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      src_off[i] = (unsigned int)((i < 12 ? WS_KDT : WS_AQK) + row * 128 + c * 16);
    } else {
      const int row = 8 * wave + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
      src_off[i] = (unsigned int)(WS_UT + (size_t)(v0 + row) * 128 + c * 16);
    }
  }
  auto issue_dma = [&](int ci, int parity, int first, int last) {     // pieces first..last-1 of chunk ci
    const unsigned char* rec = ws + ((size_t)bh * nt_seg + ci) * WS_STRIDE;
    const unsigned int img = lds0 + (unsigned int)(parity * OP_BYTES);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i < first || i >= last) continue;
      if (i < 15) {
        dma_piece(rec + src_off[i], img + (unsigned int)((4 * i + wave_u) * 1024));
      } else if (wave_u == 0) {
        dma_piece(rec + WS_EG + lane * 16, img + (unsigned int)OP_EG);
      }
    }
  };

  trace_stamp(trace, 17);
  issue_dma(0, 0, 0, 16);      // first chunk's operands and the state slab are fetched concurrently

  // state slab rows 32*wave + crow32(r,hi), column v0 + l31 (dtype branch hoisted: 16 loads in flight)
  f32x16 S;
  {
    const size_t base = ((size_t)bh * GK + 32 * wave) * GV + v0 + l31;
    if (h0 == nullptr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = 0.f;
    } else if (h0_dtype == IVL_F32) {
      const float* hp = (const float*)h0 + base;
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = hp[(size_t)crow32(r, hi) * GV];
    } else {
      const bf16_t* hp = (const bf16_t*)h0 + base;
      bf16_t raw[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) raw[r] = hp[(size_t)crow32(r, hi) * GV];
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = bf2f(raw[r]);
    }
  }

  // lane-constant image offsets of the fragments this wave reads every chunk.  [Wg;Qh] S and the output use
  // 16-row tiles (v_mfma_f32_16x16x32_bf16: A[i=l&15][k=8(l>>4)..+7], C[i=4(l>>4)+r][j=l&15]) so that every wave
  // owns 16 rows of v_new AND 16 rows of the output: the four waves do identical work between barriers.
  const int l15 = lane & 15, g4 = lane >> 4;
  const int arow = 16 * wave + l15;                              // Wg / Qh / Aqk row of this lane's A fragments
  const int w_off = OP_WG + arow * 256, qh_off = OP_QH + arow * 256, q_off = OP_AQK + arow * 128;
  const int krow = 32 * wave + l31;                              // KdT row (state row owned by this lane's wave)
  const int k_off = OP_KDT + krow * 128;
  const int trow = 16 * wave + 4 * g4;                           // first of this lane's 4 C rows (times)

  // coalesced store of a finished 64x32 output tile from LDS: thread -> (row tid>>2, 8 columns), 16 bytes
  auto flush_o = [&](int tc0) {
    const int row = tid >> 2, part = tid & 3;
    const int t = tc0 + row;
    if (t < T) *(u32x4*)(o + (((size_t)b * T + t) * H + h) * GV + v0 + 8 * part) = *(const u32x4*)(s_o + row * S_LDO + 8 * part);
  };

  // Per chunk: publish S -> wait for this chunk's image -> barrier -> [DMA of the next chunk interleaved with]
  // (ii) [Wg;Qh] S  -> barrier -> (iii) state update / output.  Two barriers per chunk; the DMA of chunk c+1 is
  // issued after the first barrier of chunk c (every wave has left chunk c-1, whose image it overwrites) and is
  // awaited with vmcnt(0) at the top of chunk c+1, a whole chunk of MFMA work later.
  for (int ci = 0; ci < nt_seg; ++ci) {
    const int tc0 = t_seg0 + ci * GC;
    const unsigned char* img = smem + (ci & 1) * OP_BYTES;
    const int cn = min(ci + 1, nt_seg - 1), pn = (ci + 1) & 1;      // next record (clamped) and its image
    if (ci < 4) trace_stamp(trace, 18 + 4 * ci);
    // ---- (i) publish the state slab as bf16 S^T[col][k] ------------------------------------------------
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      u32x2 w;
      w.x = pack2bf(S[4 * r4 + 0], S[4 * r4 + 1]);
      w.y = pack2bf(S[4 * r4 + 2], S[4 * r4 + 3]);
      // piece (k/8) = 4 wave + r4, stored at piece ^ (row & 15): the 16-row fragment reads below (ds_read_b128 lane
      // groups {0-3,12-15,20-27}, ...) are then conflict-free; a padded 272-byte stride was not (33 % of the LDS cycles)
      *(u32x2*)((unsigned char*)s_st + l31 * 256 + swz16(l31, 4 * wave + r4) + 8 * hi) = w;
    }
    dma_wait_all();                         // this wave's pieces of chunk ci have landed
    if (ci < 4) trace_stamp(trace, 40 + 4 * ci);
    lds_barrier();                          // ... and so have everyone else's; S^T visible
    if (ci < 4) trace_stamp(trace, 19 + 4 * ci);
    // every LDS operand of phase (ii) is requested before the first MFMA; the DMA pieces of the next chunk are
    // issued branch-free in between (past the last chunk they re-fetch it into the idle image: harmless), so the
    // whole phase is one basic block the scheduler can interleave freely.
    u32x4 aw[4], aq[4], bs[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      aw[ks] = *(const u32x4*)(img + w_off + swz16(arow, 4 * ks + g4));
      aq[ks] = *(const u32x4*)(img + qh_off + swz16(arow, 4 * ks + g4));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        bs[ks][nt] = *(const u32x4*)((const unsigned char*)s_st + (16 * nt + l15) * 256 + swz16(l15, 4 * ks + g4));
    }
    u32x2 uu[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = 16 * nt + l15;
      uu[nt] = *(const u32x2*)(img + OP_UT + col * 128 + swz8(col, 2 * wave + (g4 >> 1)) + 8 * (g4 & 1));
    }
    const f32x4 eg = *(const f32x4*)(img + OP_EG + trow * 4);
    const float egl = *(const float*)(img + OP_EG + 256);
    if (ci > 0) flush_o(tc0 - GC);          // previous chunk's output tile (written before this barrier)
    issue_dma(cn, pn, 0, 4);

    // ---- (ii) [Wg ; Qh] S : every wave 16 rows of each, K = 128 (four independent accumulation chains) ---
    f32x4 accW[2], accQ[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { accW[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; accQ[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        accW[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf(aw[ks]), mf(bs[ks][nt]), accW[nt], 0, 0, 0);
        accQ[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf(aq[ks]), mf(bs[ks][nt]), accQ[nt], 0, 0, 0);
      }
      issue_dma(cn, pn, 4 + 2 * ks, 6 + 2 * ks);                // pieces 4..11 under the MFMAs
    }
    if (ci < 4) trace_stamp(trace, 41 + 4 * ci);
    // v_new = u - Wg S  -> bf16 -> v_new^T[col][time];   (Qh S) * e^gamma_i
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      u32x2 w;
      w.x = pack2bf(bflo(uu[nt].x) - accW[nt][0], bfhi(uu[nt].x) - accW[nt][1]);
      w.y = pack2bf(bflo(uu[nt].y) - accW[nt][2], bfhi(uu[nt].y) - accW[nt][3]);
      *(u32x2*)((unsigned char*)s_vn + (16 * nt + l15) * 128 + swz8(16 * nt + l15, 2 * wave + (g4 >> 1)) + 8 * (g4 & 1)) = w;
#pragma unroll
      for (int r = 0; r < 4; ++r) accQ[nt][r] *= eg[r];
    }
    // operands of phase (iii) that do not depend on v_new: requested before the barrier
    u32x4 kd[4], aa[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kd[ks] = *(const u32x4*)(img + k_off + swz8(krow, 2 * ks + hi));
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) aa[k2] = *(const u32x4*)(img + q_off + swz8(arow, 4 * k2 + g4));
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] *= egl;
    if (ci < 4) trace_stamp(trace, 42 + 4 * ci);
    lds_barrier();
    if (ci < 4) trace_stamp(trace, 20 + 4 * ci);

    // ---- (iii) state update (32x32 tile per wave) ; output rows 16w..16w+15: + Aqk v_new -------------------
    {
      const unsigned char* vp = (const unsigned char*)s_vn + l31 * 128;
      u32x4 vfr[4], bv[2][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) vfr[ks] = *(const u32x4*)(vp + swz8(l31, 2 * ks + hi));
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          bv[k2][nt] = *(const u32x4*)((const unsigned char*)s_vn + (16 * nt + l15) * 128 + swz8(16 * nt + l15, 4 * k2 + g4));
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(kd[ks]), mf(vfr[ks]), S, 0, 0, 0);
        if (ks < 2) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            accQ[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf(aa[ks]), mf(bv[ks][nt]), accQ[nt], 0, 0, 0);
        }
        issue_dma(cn, pn, 12 + ks, 13 + ks);                    // pieces 12..15
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_o[(trow + r) * S_LDO + 16 * nt + l15] = f2bf(accQ[nt][r] * scale);
    if (ci < 4) trace_stamp(trace, 21 + 4 * ci);
  }
  dma_wait_all();              // the clamped re-fetch issued during the last chunk
  lds_barrier();
  flush_o(t_seg0 + (nt_seg - 1) * GC);
  trace_stamp(trace, 34);

  if (ht != nullptr) {
    const size_t base = ((size_t)bh * GK + 32 * wave) * GV + v0 + l31;
    if (ht_dtype == IVL_F32) {
      float* hp = (float*)ht + base;
#pragma unroll
      for (int r = 0; r < 16; ++r) hp[(size_t)crow32(r, hi) * GV] = S[r];
    } else {
      bf16_t* hp = (bf16_t*)ht + base;
#pragma unroll
      for (int r = 0; r < 16; ++r) hp[(size_t)crow32(r, hi) * GV] = f2bf(S[r]);
    }
  }
  trace_stamp(trace, 35);
}

}  // namespace ivl

using namespace ivl;

static inline int seg_chunks(int NT) { return NT < G_SEG_CHUNKS ? NT : G_SEG_CHUNKS; }

extern "C" size_t ivl_gdn_chunk_workspace_bytes(int B, int T, int H, int K, int V) {
  if (B <= 0 || T <= 0 || H <= 0 || K != GK || V != GV) return 0;
  const int NT = (T + GC - 1) / GC;
  size_t bytes = (size_t)B * H * seg_chunks(NT) * WS_STRIDE;
  if (NT > G_SEG_CHUNKS) bytes += (size_t)B * H * GK * GV * sizeof(float);   // fp32 state carried between segments
  return bytes;
}

extern "C" int ivl_gdn_chunk_fwd(const void* q, const void* k, const void* v, const float* g, const void* beta,
                                 void* o, const void* h0, int h0_dtype, void* ht, int ht_dtype,
                                 int B, int T, int H, int K, int V, float scale, int use_qk_l2norm,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  IVL_REQUIRE(q && k && v && g && beta && o, IVL_ERR_INVALID_ARG, "ivl_gdn_chunk_fwd: NULL pointer");
  IVL_REQUIRE(B > 0 && T > 0 && H > 0, IVL_ERR_INVALID_ARG, "ivl_gdn_chunk_fwd: B,T,H must be positive (%d,%d,%d)", B, T, H);
  IVL_REQUIRE(K == GK && V == GV, IVL_ERR_UNSUPPORTED, "ivl_gdn_chunk_fwd: built for K=128,V=256 (got %d,%d)", K, V);
  IVL_REQUIRE((h0 == nullptr || h0_dtype == IVL_F32 || h0_dtype == IVL_BF16) &&
              (ht == nullptr || ht_dtype == IVL_F32 || ht_dtype == IVL_BF16),
              IVL_ERR_INVALID_ARG, "ivl_gdn_chunk_fwd: state dtype must be IVL_F32 or IVL_BF16");
  const size_t need = ivl_gdn_chunk_workspace_bytes(B, T, H, K, V);
  IVL_REQUIRE(workspace != nullptr && workspace_bytes >= need, IVL_ERR_WORKSPACE,
              "ivl_gdn_chunk_fwd: workspace %zu bytes < required %zu", workspace_bytes, need);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gdn_chunk_prepare_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P_BYTES);
    (void)hipFuncSetAttribute((const void*)gdn_chunk_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SC_BYTES);
    attr_set = true;
  }
  hipStream_t st = (hipStream_t)stream;
  const int NT = (T + GC - 1) / GC;
  const int segc = seg_chunks(NT);
  unsigned char* wsb = (unsigned char*)workspace;
  float* carry = NT > G_SEG_CHUNKS ? (float*)(wsb + (size_t)B * H * segc * WS_STRIDE) : nullptr;
  for (int c0 = 0; c0 < NT; c0 += segc) {
    const int nseg = (NT - c0) < segc ? (NT - c0) : segc;
    const bool first = c0 == 0, last = c0 + nseg >= NT;
    hipLaunchKernelGGL(gdn_chunk_prepare_kernel, dim3(nseg, B * H), dim3(256), P_BYTES, st,
                       (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, g, (const bf16_t*)beta, wsb,
                       T, H, c0 * GC, nseg, use_qk_l2norm, debug_trace_buffer());
    int rc = check_launch("ivl_gdn_chunk_fwd(prepare)");
    if (rc != IVL_OK) return rc;
    const void* hin = first ? h0 : (const void*)carry;
    const int hin_dt = first ? h0_dtype : IVL_F32;
    void* hout = last ? ht : (void*)carry;
    const int hout_dt = last ? ht_dtype : IVL_F32;
    hipLaunchKernelGGL(gdn_chunk_scan_kernel, dim3(B * H, GV / G_BV), dim3(256), SC_BYTES, st,
                       (const unsigned char*)wsb, (bf16_t*)o, hin, hin_dt, hout, hout_dt, T, H, c0 * GC, nseg, scale, debug_trace_buffer());
    rc = check_launch("ivl_gdn_chunk_fwd(scan)");
    if (rc != IVL_OK) return rc;
  }
  return IVL_OK;
}
