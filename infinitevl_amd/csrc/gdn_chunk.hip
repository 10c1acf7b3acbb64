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
constexpr size_t WS_STRIDE = WS_EGL + 256;        // 90624

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
    int T, int H, int t_seg0, int nt_seg, int l2norm, int dbg_stop, long long* trace) {
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
  if (dbg_stop == 1) return;   // timing ladder (IVL_DEBUG_PREP_STOP)
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
  if (dbg_stop == 2) return;   // timing ladder (IVL_DEBUG_PREP_STOP)
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
  if (dbg_stop == 3) return;   // timing ladder (IVL_DEBUG_PREP_STOP)
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
  if (dbg_stop == 4) return;   // timing ladder (IVL_DEBUG_PREP_STOP)
  trace_stamp(trace, 5);
  if (dbg_stop == 5) return;   // timing ladder (IVL_DEBUG_PREP_STOP)
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
  if (dbg_stop == 6) return;   // timing ladder (IVL_DEBUG_PREP_STOP)
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
constexpr int S_LDS = 136;     // bf16 per row of S^T  [32 cols][128 k]   (272 B)
constexpr int S_LDV = 72;      // bf16 per row of v_new^T [32 cols][64 t] (144 B)
constexpr int S_LDO = 40;      // bf16 per row of the output staging tile [64 t][32 cols] (80 B)

// LDS operand buffer of one chunk (bytes).  Every region is an image of the workspace record region with the
// 16-byte chunk index XOR-swizzled by the row, so that the 16-byte MFMA fragment reads (32 consecutive rows,
// same chunk) are bank-conflict free although the rows are 256 / 128 bytes apart.
constexpr int OP_WG = 0;                   // [64][16 chunks]   chunk' = c ^ (row & 15)
constexpr int OP_QH = OP_WG + 16384;       // [64][16 chunks]
constexpr int OP_KDT = OP_QH + 16384;      // [128][8 chunks]   chunk' = c ^ ((row >> 1) & 7)
constexpr int OP_AQK = OP_KDT + 16384;     // [64][8 chunks]
constexpr int OP_UT = OP_AQK + 8192;       // [32][8 chunks]    (this workgroup's 32 columns of u^T)
constexpr int OP_BYTES = OP_UT + 4096;     // 61440
constexpr int SC_ST = OP_BYTES;                            // S^T   bf16 [32][136]
constexpr int SC_VN = SC_ST + G_BV * S_LDS * 2;            // v_new^T bf16 [32][72]
constexpr int SC_O = SC_VN + G_BV * S_LDV * 2;             // o tile bf16 [64][40]
constexpr int SC_BYTES = SC_O + GC * S_LDO * 2;            // 79872
constexpr int SC_PIECES = OP_BYTES / 16 / 256;             // 15 sixteen-byte pieces per thread per chunk
static_assert(SC_PIECES * 256 * 16 == OP_BYTES, "operand image must split evenly over 256 threads");

__device__ __forceinline__ int swz16(int row, int c) { return (c ^ (row & 15)) << 4; }          // 256-byte rows
__device__ __forceinline__ int swz8(int row, int c) { return (c ^ ((row >> 1) & 7)) << 4; }     // 128-byte rows

__global__ __launch_bounds__(256) void gdn_chunk_scan_kernel(
    const unsigned char* __restrict__ ws, bf16_t* __restrict__ o,
    const void* h0, int h0_dtype, void* ht, int ht_dtype,
    int T, int H, int t_seg0, int nt_seg, float scale, long long* trace) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* s_st = (bf16_t*)(smem + SC_ST);
  bf16_t* s_vn = (bf16_t*)(smem + SC_VN);
  bf16_t* s_o = (bf16_t*)(smem + SC_O);

  trace_stamp(trace, 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // grid = (B*H, V/32): linear block id = bh + B*H*slab, so with B*H % 8 == 0 the 8 V-slabs of one head run on
  // the same XCD (id % 8) and share its L2 for the operands they all read (Wg/Qh/KdT/Aqk).
  const int v0 = blockIdx.y * G_BV;
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh % H;
  const bool is_p = wave < 2;                 // waves 0,1: v_new rows 32*wave.. ; waves 2,3: output rows 32*(wave-2)..
  const int mrow0 = 32 * (wave & 1);

  // ---- operand staging: the record of a chunk is fetched as whole 1 KB wavefront loads (fully coalesced; the
  //      earlier fragment-shaped loads touched 32 cache lines per instruction and were bound by the per-CU
  //      load path at ~25 GB/s) into registers one chunk ahead, then written to the swizzled LDS image ------
  // piece i of thread tid covers record bytes [(i*256 + tid)*16, +16) of the concatenation WG|QH|KDT|AQK|UTslab
  struct Stage { u32x4 v[SC_PIECES]; f32x4 egv[4]; float egl; };
  auto issue_loads = [&](Stage& st, int ci) {
    const unsigned char* rec = ws + ((size_t)bh * nt_seg + ci) * WS_STRIDE;
#pragma unroll
    for (int i = 0; i < SC_PIECES; ++i) {
      const unsigned char* src;
      if (i < 4) src = rec + WS_WG + (size_t)(i * 256 + tid) * 16;
      else if (i < 8) src = rec + WS_QH + (size_t)((i - 4) * 256 + tid) * 16;
      else if (i < 12) src = rec + WS_KDT + (size_t)((i - 8) * 256 + tid) * 16;
      else if (i < 14) src = rec + WS_AQK + (size_t)((i - 12) * 256 + tid) * 16;
      else src = rec + WS_UT + (size_t)v0 * GC * 2 + (size_t)tid * 16;
      st.v[i] = *(const u32x4*)src;
    }
    if (!is_p) {
      const float* ep = (const float*)(rec + WS_EG) + mrow0 + 4 * hi;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) st.egv[r4] = *(const f32x4*)(ep + 8 * r4);
    }
    st.egl = *(const float*)(rec + WS_EGL);
  };
  auto write_stage = [&](const Stage& st) {
#pragma unroll
    for (int i = 0; i < SC_PIECES; ++i) {
      int off;
      if (i < 8) {                      // WG / QH: 16 chunks per row
        const int q = (i & 3) * 256 + tid, row = q >> 4, c = q & 15;
        off = (i < 4 ? OP_WG : OP_QH) + row * 256 + swz16(row, c);
      } else if (i < 14) {              // KDT / AQK: 8 chunks per row
        const int q = (i < 12 ? (i - 8) : (i - 12)) * 256 + tid, row = q >> 3, c = q & 7;
        off = (i < 12 ? OP_KDT : OP_AQK) + row * 128 + swz8(row, c);
      } else {                          // UT slab: 32 rows x 8 chunks
        const int row = tid >> 3, c = tid & 7;
        off = OP_UT + row * 128 + swz8(row, c);
      }
      *(u32x4*)(smem + off) = st.v[i];
    }
  };

  // L2 warm-up: the record of a chunk was written by another CU (usually another XCD), so its first touch
  // on this XCD misses L2 and pays the MALL/HBM latency (~3k cycles, all 8 slab workgroups of the head stall on
  // the same lines).  Touch one dword of each of its 480 cache lines three chunks ahead; the staged 1 KB loads
  // issued one chunk ahead then hit L2.
  unsigned int sink = 0;       // keeps the warm-up loads alive; consumed long after they were issued
  auto warm_l2 = [&](int ci, unsigned int& w0, unsigned int& w1) {
    w0 = w1 = 0;
    if (ci >= nt_seg) return;
    const unsigned char* rec = ws + ((size_t)bh * nt_seg + ci) * WS_STRIDE;
    {
      const int l = tid;                                   // lines 0..255 of WG|QH|KDT
      w0 = *(const unsigned int*)(rec + (size_t)l * 128);
    }
    {
      const int l = 256 + tid;                             // lines 256..479: rest of KDT, AQK, this slab of UT
      if (l < 480) {
        const size_t off = l < 384 ? (size_t)l * 128
                                   : (l < 448 ? WS_AQK + (size_t)(l - 384) * 128
                                              : WS_UT + (size_t)v0 * GC * 2 + (size_t)(l - 448) * 128);
        w1 = *(const unsigned int*)(rec + off);
      }
    }
  };

  trace_stamp(trace, 17);
  Stage sa, sb;
  issue_loads(sa, 0);          // first chunk's operands and the state slab are fetched concurrently
  unsigned int wa0, wa1, wb0, wb1;
  warm_l2(1, wa0, wa1);
  warm_l2(2, wb0, wb1);
  sink ^= wa0 ^ wa1 ^ wb0 ^ wb1;

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

  // lane-constant LDS addresses of the fragments this wave reads every chunk
  const int arow = mrow0 + l31;                                  // Wg / Qh / Aqk row
  const unsigned char* a_base = smem + (is_p ? OP_WG : OP_QH) + arow * 256;
  const int krow = 32 * wave + l31;                              // KdT row (state row owned by this lane's wave)
  const unsigned char* k_base = smem + OP_KDT + krow * 128;
  const unsigned char* q_base = smem + OP_AQK + arow * 128;
  const unsigned char* u_base = smem + OP_UT + l31 * 128 + 8 * hi;

  // coalesced store of a finished 64x32 output tile from LDS: thread -> (row tid>>2, 8 columns), 16 bytes
  auto flush_o = [&](int tc0) {
    const int row = tid >> 2, part = tid & 3;
    const int t = tc0 + row;
    if (t < T) *(u32x4*)(o + (((size_t)b * T + t) * H + h) * GV + v0 + 8 * part) = *(const u32x4*)(s_o + row * S_LDO + 8 * part);
  };

  // one chunk of the recurrence; its operands are in the LDS image, egv/egl in `st` ----------------------
  auto chunk_step = [&](const Stage& st, int ci) {
    const int tc0 = t_seg0 + ci * GC;
    if (ci < 4) trace_stamp(trace, 18 + 4 * ci);
    // ---- (i) publish the state slab as bf16 S^T[col][k]; flush the previous chunk's output tile --------
    if (ci > 0) flush_o(tc0 - GC);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      u32x2 w;
      w.x = pack2bf(S[4 * r4 + 0], S[4 * r4 + 1]);
      w.y = pack2bf(S[4 * r4 + 2], S[4 * r4 + 3]);
      *(u32x2*)(s_st + l31 * S_LDS + 32 * wave + 8 * r4 + 4 * hi) = w;
    }
    __syncthreads();
    if (ci < 4) trace_stamp(trace, 19 + 4 * ci);

    // ---- (ii) [Wg ; Qh] S : every wave one 32x32 tile over K = 128 (two independent accumulation chains) --
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    {
      const bf16_t* bp = s_st + l31 * S_LDS + 8 * hi;
#pragma unroll
      for (int ks = 0; ks < 8; ks += 2) {
        const u32x4 a0 = *(const u32x4*)(a_base + swz16(arow, 2 * ks + hi));
        const u32x4 a1 = *(const u32x4*)(a_base + swz16(arow, 2 * ks + 2 + hi));
        const u32x4 b0 = *(const u32x4*)(bp + 16 * ks);
        const u32x4 b1 = *(const u32x4*)(bp + 16 * ks + 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(a0), mf(b0), acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(a1), mf(b1), acc2, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    }
    if (is_p) {
      // v_new = u - Wg S  -> bf16 -> v_new^T[col][time]
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const u32x2 uu = *(const u32x2*)(u_base + swz8(l31, mrow0 / 8 + r4));
        const float u0 = bflo(uu.x), u1 = bfhi(uu.x), u2 = bflo(uu.y), u3 = bfhi(uu.y);
        u32x2 w;
        w.x = pack2bf(u0 - acc[4 * r4 + 0], u1 - acc[4 * r4 + 1]);
        w.y = pack2bf(u2 - acc[4 * r4 + 2], u3 - acc[4 * r4 + 3]);
        *(u32x2*)(s_vn + l31 * S_LDV + mrow0 + 8 * r4 + 4 * hi) = w;
      }
    } else {
      // (Qh S) * e^gamma_i
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * r4 + i] *= st.egv[r4][i];
    }
    __syncthreads();
    if (ci < 4) trace_stamp(trace, 20 + 4 * ci);

    // ---- (iii) state update (all waves) ; output rows (waves 2,3): + Aqk v_new -----------------------------
    const bf16_t* vp = s_vn + l31 * S_LDV + 8 * hi;
    u32x4 vfr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) vfr[ks] = *(const u32x4*)(vp + 16 * ks);
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] *= st.egl;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4 kd = *(const u32x4*)(k_base + swz8(krow, 2 * ks + hi));
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(kd), mf(vfr[ks]), S, 0, 0, 0);
    }
    if (!is_p) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4 aq = *(const u32x4*)(q_base + swz8(arow, 2 * ks + hi));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(aq), mf(vfr[ks]), acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s_o[(mrow0 + crow32(r, hi)) * S_LDO + l31] = f2bf(acc[r] * scale);
    }
    if (ci < 4) trace_stamp(trace, 21 + 4 * ci);
  };

  // pipeline: while chunk c is computed from the LDS image, chunk c+1 is in flight into registers; it is
  // written to the (single) LDS image when every wave is done with chunk c.  Unrolled by two (no copies).
  write_stage(sa);
  for (int ci = 0; ci < nt_seg; ci += 2) {
    if (ci + 1 < nt_seg) issue_loads(sb, ci + 1);
    warm_l2(ci + 3, wa0, wa1);
    __syncthreads();                       // image of chunk ci complete; previous readers of s_st / s_vn done
    chunk_step(sa, ci);
    sink ^= wa0 ^ wa1;
    if (ci + 1 < nt_seg) {
      if (ci < 4) trace_stamp(trace, 40 + 4 * ci);
      __syncthreads();                     // every wave has finished reading the image of chunk ci
      if (ci < 4) trace_stamp(trace, 41 + 4 * ci);
      write_stage(sb);
      if (ci < 4) trace_stamp(trace, 42 + 4 * ci);
      if (ci + 2 < nt_seg) issue_loads(sa, ci + 2);
      if (ci < 4) trace_stamp(trace, 43 + 4 * ci);
      warm_l2(ci + 4, wb0, wb1);
      __syncthreads();
      chunk_step(sb, ci + 1);
      sink ^= wb0 ^ wb1;
      if (ci + 2 < nt_seg) {
        __syncthreads();
        write_stage(sa);
      }
    }
  }
  __syncthreads();
  flush_o(t_seg0 + (nt_seg - 1) * GC);
  if (sink == 0x9e3779b9u && trace != nullptr) trace[63] = sink;     // never true in practice; keeps `sink` live
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
  static int dbg_stop = -1;
  if (dbg_stop < 0) {
    const char* e = getenv("IVL_DEBUG_PREP_STOP");
    dbg_stop = e ? atoi(e) : 0;
  }
  static int dbg_skip_scan = -1;
  if (dbg_skip_scan < 0) {
    const char* e = getenv("IVL_DEBUG_SKIP_SCAN");
    dbg_skip_scan = e ? atoi(e) : 0;
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
                       T, H, c0 * GC, nseg, use_qk_l2norm, dbg_stop, debug_trace_buffer());
    int rc = check_launch("ivl_gdn_chunk_fwd(prepare)");
    if (rc != IVL_OK) return rc;
    const void* hin = first ? h0 : (const void*)carry;
    const int hin_dt = first ? h0_dtype : IVL_F32;
    void* hout = last ? ht : (void*)carry;
    const int hout_dt = last ? ht_dtype : IVL_F32;
    if (dbg_skip_scan) continue;
    hipLaunchKernelGGL(gdn_chunk_scan_kernel, dim3(B * H, GV / G_BV), dim3(256), SC_BYTES, st,
                       (const unsigned char*)wsb, (bf16_t*)o, hin, hin_dt, hout, hout_dt, T, H, c0 * GC, nseg, scale, debug_trace_buffer());
    rc = check_launch("ivl_gdn_chunk_fwd(scan)");
    if (rc != IVL_OK) return rc;
  }
  return IVL_OK;
}
