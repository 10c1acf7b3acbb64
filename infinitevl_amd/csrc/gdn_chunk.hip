// Gated DeltaNet, chunkwise form (chunk C = 64 tokens) for gfx950, K = 128, V = 256.
// Replaces fla's six-kernel pipeline (fla:ops/gated_delta_rule/chunk.py:18-71: l2norm, cumsum, WY transform
// wy_fast.py:114-238, state scan chunk_delta_h.py:32-124, output chunk_o.py:92-113) with two launches.
//
//  (1) gdn_chunk_prepare_kernel -- chunk-PARALLEL, one 512-thread workgroup per (chunk, batch*head).  Everything
//      that depends on q, k, g, beta only (the "K side"):
//        q_hat, k_hat = l2norm -> bf16;  gamma = cumsum(g);  L = tril(bf16(beta k_hat) k_hat^T, -1);
//        Tw = (I+L)^-1 in fp32 (16x16 diagonal blocks by column-parallel substitution in registers, the rest by
//        block elimination on the exact-fp32 MFMA v_mfma_f32_16x16x4_f32, intermediates chained in registers);
//        Tu = Tw * e^{gamma_i-gamma_j};  w = bf16(Tw) bf16(beta k_hat);  A = tril((q_hat k_hat^T) * Gamma).
//      It leaves one 65 KB record per chunk holding the five A-operand matrices of the serial pass, already
//      decayed / negated and stored as a sequence of 1 KB MFMA FRAGMENT BLOCKS (below), plus e^gamma and beta.
//  (2) gdn_chunk_scan_kernel -- SERIAL over chunks.  One WAVE owns a 16-column slab of the state for all 128
//      rows: S[128 x 16] fp32 lives in 32 accumulator registers for the whole call and never visits LDS.  The
//      MFMA C layout (lane = column, registers = rows) IS the B-operand layout of the next product once the
//      contraction index is permuted inside each block of 32 (slot 8g+e <-> index 4g+e | 16+4g+(e-4)); the
//      records are written with that permutation, so per chunk a wave runs
//          u      = Tu (beta v)                  ( 8 MFMA, beta v from an LDS transpose read of the raw v slab)
//          v_new  = u - (w e^gamma) S            (16 MFMA, S as B operand straight from the accumulators)
//          S      = e^{gamma_L} S + (k_hat e^{gamma_L-gamma})^T v_new      (16 MFMA, v_new straight from accumulators)
//          o^T    = scale ((S^T q_hat^T) e^gamma + v_new^T A^T)            (24 MFMA, transposed: 8-byte row stores)
//      with no barrier and no LDS traffic on the S -> v_new -> S chain.  The 4 waves of a workgroup share the
//      chunk's operand image, which arrives by LDS-DMA (global_load_lds_dwordx4, four 1 KB pieces per issue)
//      into one of two alternating images; one barrier per chunk.
//
// Fragment block (v_mfma_f32_16x16x32_bf16): 16 rows x 32 contraction slots = 64 x 16 bytes, piece (g, i) at byte
// 16 (16 g + i) holds row i, slots 8g..8g+7  <->  contraction indices 4g+e (e < 4), 16+4g+(e-4) (e >= 4).  A wave
// reads a block lane-linearly (lane l = 16 g + i reads bytes [16 l, 16 l + 16)): conflict-free ds_read_b128, and
// the DMA that fills it is a plain linear copy of the record.
#include <mutex>

#include "ivl_common.h"

namespace ivl {
IVL_TRACE_DECL(gdn)

typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int GC = 64;        // chunk length
constexpr int GK = 128;       // key head dim
constexpr int GV = 256;       // value head dim
constexpr int G_SEG_CHUNKS = 64;   // chunks per workspace segment (4096 tokens)

// ---- workspace record per (batch*head, chunk): byte offsets; [0, 65536) is a sequence of 64 fragment blocks ----
constexpr int REC_WN = 0;          // -bf16(bf16(w) e^gamma)       blocks (m, s)  = 4 m + s    rows: time   contraction: k
constexpr int REC_QH = 16384;      // q_hat                        blocks (m, s)  = 4 m + s    rows: time   contraction: k
constexpr int REC_KDT = 32768;     // (k_hat e^{gl-gamma})^T       blocks (t, s2) = 2 t + s2   rows: k      contraction: time
constexpr int REC_AQK = 49152;     // tril((q k^T) Gamma)          blocks (m, s2) = 2 m + s2   rows: time   contraction: time
constexpr int REC_TU = 57344;      // bf16(Tw e^{gamma_i-gamma_j}) blocks (m, s2) = 2 m + s2   rows: time   contraction: time
constexpr int REC_EG = 65536;      // f32 e^gamma[64]
constexpr int REC_EGL = REC_EG + 256;    // f32 e^gamma_last
constexpr int REC_BETA = REC_EG + 512;   // f32 beta[64]
constexpr size_t REC_STRIDE = 66560;     // 65 pieces of 1 KB

__device__ __forceinline__ mfma_bf16x8 mf(u32x4 v) {
  mfma_bf16x8 r;
  __builtin_memcpy(&r, &v, 16);
  return r;
}
__device__ __forceinline__ int crow32(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ __forceinline__ u32x4 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
  return u32x4{pack2bf(a0, a1), pack2bf(a2, a3), pack2bf(a4, a5), pack2bf(a6, a7)};
}

// ==================================================================================================
// (1) chunk-parallel pre-pass
// ==================================================================================================
constexpr int P_LDK = 136;                         // bf16 elements per row of k_hat / q_hat / beta k_hat (272 B)
constexpr int P_LDF = 68;                          // f32 elements per row of L / T
constexpr int P_KH = 0;                            // k_hat            [64][136] bf16
constexpr int P_QH = P_KH + GC * P_LDK * 2;        // q_hat            [64][136]
constexpr int P_KB = P_QH + GC * P_LDK * 2;        // bf16(beta k_hat) [64][136]
constexpr int P_L = P_KB + GC * P_LDK * 2;         // L, inverted IN PLACE to T = (I+L)^-1   [64][68] f32
constexpr int P_SM = P_L + GC * P_LDF * 4;         // gam[64], eg[64], dec[64]
constexpr int P_BYTES = P_SM + 4 * GC * 4;         // 70,656: two workgroups per CU
static_assert(2 * P_BYTES <= 160 * 1024, "pre-pass LDS budget (2 workgroups per CU)");

// sum over the 16 lanes of a DPP row (the 16 threads that share one q / k row), result in every lane
__device__ __forceinline__ float dpp_add(float x, const int ctrl_sel) {
  const int xi = __builtin_bit_cast(int, x);
  int y;
  if (ctrl_sel == 0) y = __builtin_amdgcn_update_dpp(0, xi, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
  else if (ctrl_sel == 1) y = __builtin_amdgcn_update_dpp(0, xi, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  else if (ctrl_sel == 2) y = __builtin_amdgcn_update_dpp(0, xi, 0x141, 0xF, 0xF, true);  // row_half_mirror
  else y = __builtin_amdgcn_update_dpp(0, xi, 0x140, 0xF, 0xF, true);                     // row_mirror
  return x + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float row16_sum(float x) {
  x = dpp_add(x, 0);
  x = dpp_add(x, 1);
  x = dpp_add(x, 2);
  return dpp_add(x, 3);
}

// 32x32x16 MFMA operand fragment taken from a ROW-MAJOR LDS tile X[k][n] (row stride ld elements) with the gfx950
// transpose read: lane receives X[k0 + 8(lane>>5) + e][n0 + (lane&31)], e = 0..7.  Each 16-lane group reads a
// 4x16 block (lane i supplies the address of row i>>2, columns 4(i&3)..+3 and receives column i).
__device__ __forceinline__ u32x4 frag_tr32(const bf16_t* X, int ld, int k0, int n0, int lane) {
  const int i = lane & 15, gq = lane >> 4;
  const bf16_t* p = X + (k0 + 8 * (gq >> 1) + (i >> 2)) * ld + n0 + 16 * (gq & 1) + 4 * (i & 3);
  const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * ld));
  u32x2 w0, w1;
  __builtin_memcpy(&w0, &a0, 8);
  __builtin_memcpy(&w1, &a1, 8);
  return u32x4{w0.x, w0.y, w1.x, w1.y};
}

__global__ __launch_bounds__(512, 2) void gdn_chunk_prepare_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const float* __restrict__ g,
    const bf16_t* __restrict__ beta, unsigned char* __restrict__ ws, int T, int H, int t_seg0, int nt_seg, int l2norm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* s_kh = (bf16_t*)(smem + P_KH);
  bf16_t* s_qh = (bf16_t*)(smem + P_QH);
  bf16_t* s_kb = (bf16_t*)(smem + P_KB);
  float* s_L = (float*)(smem + P_L);          // L, then T in place
  float* s_gam = (float*)(smem + P_SM);
  float* s_eg = s_gam + GC;
  float* s_dec = s_eg + GC;

  IVL_T(tp0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int l31 = lane & 31, hi = lane >> 5;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int ci = blockIdx.x;                  // chunk within the segment
  const int bh = blockIdx.y;
  const int b = bh / H, h = bh % H;
  const int t0 = t_seg0 + ci * GC;            // first token of the chunk
  const int nvalid = min(GC, T - t0);
  unsigned char* rec = ws + ((size_t)bh * nt_seg + ci) * REC_STRIDE;

  // ---- P0: every global load of the chunk is issued up front (clamped rows, zeroed later) ------------------
  const int oct = tid & 15, r0 = tid >> 4;         // thread -> rows r0, r0 + 32; 16-byte column octet
  u32x4 kraw[2], qraw[2];
  bf16_t braw[2];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int row = min(r0 + 32 * rr, nvalid - 1);
    const size_t tok = ((size_t)b * T + t0 + row) * H + h;
    kraw[rr] = *(const u32x4*)(k + tok * GK + 8 * oct);
    qraw[rr] = *(const u32x4*)(q + tok * GK + 8 * oct);
    braw[rr] = beta[tok];
  }
  // ---- P1a (wave 0): g -> chunk-local inclusive cumsum -> e^gamma, decay to the chunk end; beta ---------------
  if (wave_u == 0) {
    const size_t tok = ((size_t)b * T + t0 + min(lane, nvalid - 1)) * H + h;
    const float g_ld = g[tok];
    const float b_ld = bf2f(beta[tok]);
    float gv = lane < nvalid ? g_ld : 0.f;
    const float bv = lane < nvalid ? b_ld : 0.f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float up = __shfl_up(gv, o, 64);
      if (lane >= o) gv += up;
    }
    const float gl = __shfl(gv, nvalid - 1, 64);     // gamma at the last VALID token
    const float e = __expf(gv);
    s_gam[lane] = gv;
    s_eg[lane] = e;
    s_dec[lane] = __expf(gl - gv);                   // e^{gamma_last - gamma_t}
    ((float*)(rec + REC_EG))[lane] = e;
    ((float*)(rec + REC_BETA))[lane] = bv;
    if (lane == 0) *(float*)(rec + REC_EGL) = __expf(gl);
  }
  // ---- P1b: l2norm -> k_hat, q_hat (bf16);  bf16(beta k_hat) -------------------------------------------------
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int row = r0 + 32 * rr;
    const bool ok = row < nvalid;
    const u32x4 kv = kraw[rr], qv = qraw[rr];
    float kf[8] = {bflo(kv.x), bfhi(kv.x), bflo(kv.y), bfhi(kv.y), bflo(kv.z), bfhi(kv.z), bflo(kv.w), bfhi(kv.w)};
    float qf[8] = {bflo(qv.x), bfhi(qv.x), bflo(qv.y), bfhi(qv.y), bflo(qv.z), bfhi(qv.z), bflo(qv.w), bfhi(qv.w)};
    float rk = 1.f, rq = 1.f;
    if (l2norm) {
      float ks = 0.f, qs = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) { ks = fmaf(kf[c], kf[c], ks); qs = fmaf(qf[c], qf[c], qs); }
      ks = row16_sum(ks);
      qs = row16_sum(qs);
      rk = __builtin_amdgcn_rsqf(ks + 1e-6f);
      rq = __builtin_amdgcn_rsqf(qs + 1e-6f);
    }
    rk = ok ? rk : 0.f;
    rq = ok ? rq : 0.f;
    const float bt = ok ? bf2f(braw[rr]) : 0.f;
    float kb[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      kf[c] = bf_round(kf[c] * rk);
      qf[c] = qf[c] * rq;
      kb[c] = kf[c] * bt;
    }
    *(u32x4*)(s_kh + row * P_LDK + 8 * oct) = pack8(kf[0], kf[1], kf[2], kf[3], kf[4], kf[5], kf[6], kf[7]);
    *(u32x4*)(s_qh + row * P_LDK + 8 * oct) = pack8(qf[0], qf[1], qf[2], qf[3], qf[4], qf[5], qf[6], qf[7]);
    *(u32x4*)(s_kb + row * P_LDK + 8 * oct) = pack8(kb[0], kb[1], kb[2], kb[3], kb[4], kb[5], kb[6], kb[7]);
  }
  __syncthreads();                                   // B1
  IVL_T(tp1);

  // Copy-out of the two operands that are plain re-orderings of the LDS tiles, 16-byte piece `idx` of 2048:
  //   [0,1024)    QH : piece (block 4m+s, g, i) = q_hat[16m+i][32s + {4g..4g+3, 16+4g..+3}]
  //   [1024,2048) KDT: piece (block 2t+s2, g, i) = k_hat[32s2 + {4g.., 16+4g..}][16t+i] * dec[time]  (LDS transpose read)
  auto copy_piece = [&](int idx) {
    const int i = idx & 15, gg = (idx >> 4) & 3;
    if (idx < 1024) {
      const int blk = idx >> 6, m = blk >> 2, s = blk & 3;
      const bf16_t* src = s_qh + (16 * m + i) * P_LDK + 32 * s + 4 * gg;
      const u32x2 lo = *(const u32x2*)src, hi2 = *(const u32x2*)(src + 16);
      *(u32x4*)(rec + REC_QH + idx * 16) = u32x4{lo.x, lo.y, hi2.x, hi2.y};
    } else {
      const int id2 = idx - 1024, blk = id2 >> 6, t = blk >> 1, s2 = blk & 1;
      const int time0 = 32 * s2 + 4 * gg;
      const bf16_t* p = s_kh + (time0 + (i >> 2)) * P_LDK + 16 * t + 4 * (i & 3);
      const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
      const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * P_LDK));
      const f32x4 d0 = *(const f32x4*)(s_dec + time0), d1 = *(const f32x4*)(s_dec + time0 + 16);
      u32x2 w0, w1;
      __builtin_memcpy(&w0, &a0, 8);
      __builtin_memcpy(&w1, &a1, 8);
      *(u32x4*)(rec + REC_KDT + id2 * 16) =
          pack8(bflo(w0.x) * d0[0], bfhi(w0.x) * d0[1], bflo(w0.y) * d0[2], bfhi(w0.y) * d0[3],
                bflo(w1.x) * d1[0], bfhi(w1.x) * d1[1], bflo(w1.y) * d1[2], bfhi(w1.y) * d1[3]);
    }
  };

  // ---- P2: waves 0-2: L = tril(kb kh^T, -1) -> s_L;  waves 3-5: A^T = kh qh^T -> Aqk blocks;  waves 6-7: zero
  //          blocks + first copy-outs ---------------------------------------------------------------------------
  if (wave_u < 3) {
    const int mi = wave_u == 0 ? 0 : 1, ni = wave_u == 2 ? 1 : 0;       // tiles (0,0), (1,0), (1,1)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bf16_t* arow = s_kb + (32 * mi + l31) * P_LDK + 8 * hi;
    const bf16_t* brow = s_kh + (32 * ni + l31) * P_LDK + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < GK / 16; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(*(const u32x4*)(arow + 16 * ks)), mf(*(const u32x4*)(brow + 16 * ks)), acc, 0, 0, 0);
    const int j = 32 * ni + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 32 * mi + crow32(r, hi);
      s_L[i * P_LDF + j] = i > j ? acc[r] : 0.f;
    }
  } else if (wave_u < 6) {
    const int nj = wave_u == 5 ? 1 : 0, mi = wave_u == 3 ? 0 : 1;       // (key tile, query tile) = (0,0), (0,1), (1,1)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bf16_t* arow = s_kh + (32 * nj + l31) * P_LDK + 8 * hi;
    const bf16_t* brow = s_qh + (32 * mi + l31) * P_LDK + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < GK / 16; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(*(const u32x4*)(arow + 16 * ks)), mf(*(const u32x4*)(brow + 16 * ks)), acc, 0, 0, 0);
    const int i = 32 * mi + l31;                     // query row owned by this lane
    const float gi = s_gam[i];
    float val[16];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const f32x4 gj = *(const f32x4*)(s_gam + 32 * nj + 8 * a + 4 * hi);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = 32 * nj + 8 * a + 4 * hi + c;
        val[4 * a + c] = i >= j ? acc[4 * a + c] * __expf(gi - gj[c]) : 0.f;
      }
    }
    unsigned char* blk = rec + REC_AQK + ((2 * mi + (l31 >> 4)) * 2 + nj) * 1024;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      *(u32x4*)(blk + ((hi + 2 * p) * 16 + (l31 & 15)) * 16) =
          pack8(val[4 * p], val[4 * p + 1], val[4 * p + 2], val[4 * p + 3], val[8 + 4 * p], val[9 + 4 * p], val[10 + 4 * p], val[11 + 4 * p]);
  } else {
    // blocks (m < 2, s2 = 1) of Aqk and Tu lie strictly above the diagonal: zeros
    const int t2 = tid - 384;                        // 0..127
#pragma unroll
    for (int z = 0; z < 2; ++z) {
      const int idx = t2 + 128 * z;                  // 0..255: (matrix, m, piece)
      const int mat = idx >> 7, m = (idx >> 6) & 1, pc = idx & 63;
      *(u32x4*)(rec + (mat ? REC_TU : REC_AQK) + (2 * m + 1) * 1024 + pc * 16) = u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int z = 0; z < 4; ++z) copy_piece(t2 + 128 * z);                 // QH pieces 0..511
  }
  __syncthreads();                                   // B2: s_L complete
  IVL_T(tp2);

  // ---- P3: T = (I + L)^-1 in place (waves 0-3); waves 4-7 finish the copy-outs meanwhile -----------------------
  // level 0: 16x16 diagonal block `wave` by column-parallel substitution, the solution vector in registers:
  //          lane j (every 16-lane group redundantly) owns column j of the inverse (forward substitution, row by row).
  if (wave_u < 4) {
    const int bb = 16 * wave_u;
    float x[16];
    x[0] = l15 == 0 ? 1.f : 0.f;
#pragma unroll
    for (int i = 1; i < 16; ++i) {                   // row i of (I + L) x = e_j: four partial dot products in flight
      float pz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (4 * c >= i) continue;
        const f32x4 lr = *(const f32x4*)(s_L + (bb + i) * P_LDF + bb + 4 * c);   // wave-uniform address: broadcast
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * c + e < i) pz[c] = fmaf(lr[e], x[4 * c + e], pz[c]);
      }
      x[i] = (i == l15 ? 1.f : 0.f) - ((pz[0] + pz[1]) + (pz[2] + pz[3]));
    }
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s_L[(bb + i) * P_LDF + bb + l15] = x[i];
    }
  } else {
    const int t2 = tid - 256;                        // 0..255
#pragma unroll
    for (int z = 0; z < 6; ++z) copy_piece(512 + t2 + 256 * z);           // QH 512..1023, KDT 1024..2047
  }
  __syncthreads();                                   // B3
  IVL_T(tp3a);
  // level 1: X[hb][lb] = -D_hb (L[hb][lb] D_lb) for the block pairs (1,0) and (3,2); the intermediate product stays in
  //          the accumulator registers: with the contraction order k = 4g + s (lane group g, instruction s) register s
  //          of a 16x16x4 result IS the B operand of instruction s of the next product.
  if (wave_u < 2) {
    const int lb = 32 * wave_u, hb = lb + 16;
    const f32x4 a = *(const f32x4*)(s_L + (hb + l15) * P_LDF + lb + 4 * g4);
    float bq[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bq[s] = s_L[(lb + 4 * g4 + s) * P_LDF + lb + l15];
    const f32x4 a2 = *(const f32x4*)(s_L + (hb + l15) * P_LDF + hb + 4 * g4);
    f32x4 Y = f32x4{0.f, 0.f, 0.f, 0.f}, X = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) Y = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bq[s], Y, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) X = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[s], Y[s], X, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) s_L[(hb + 4 * g4 + r) * P_LDF + lb + l15] = -X[r];
  }
  __syncthreads();                                   // B4
  IVL_T(tp3b);
  // level 2: Z = -Q (L21 P), 32x32 blocks as 2x2 tiles of 16x16; wave (ib, jb) -> tile Z[ib][jb].
  //          P = T[0:32,0:32] and Q = T[32:64,32:64] are lower triangular: P[0][1] = Q[0][1] = 0.
  f32x4 Z = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ib = (wave_u >> 1) & 1, jb = wave_u & 1;
  if (wave_u < 4) {
    f32x4 M[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      M[a] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a > ib) continue;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (c < jb) continue;
        const f32x4 af = *(const f32x4*)(s_L + (32 + 16 * a + l15) * P_LDF + 16 * c + 4 * g4);
        float bq[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) bq[s] = s_L[(16 * c + 4 * g4 + s) * P_LDF + 16 * jb + l15];
#pragma unroll
        for (int s = 0; s < 4; ++s) M[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bq[s], M[a], 0, 0, 0);
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if (a > ib) continue;
      const f32x4 qf = *(const f32x4*)(s_L + (32 + 16 * ib + l15) * P_LDF + 32 + 16 * a + 4 * g4);
#pragma unroll
      for (int s = 0; s < 4; ++s) Z = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], M[a][s], Z, 0, 0, 0);
    }
  }
  __syncthreads();                                   // B5: every L21 tile has been read
  if (wave_u < 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s_L[(32 + 16 * ib + 4 * g4 + r) * P_LDF + 16 * jb + l15] = -Z[r];
  }
  __syncthreads();                                   // B6: T complete
  IVL_T(tp3);

  // ---- P4a: Tu = Tw * e^{gamma_i - gamma_j} -> bf16 fragment blocks, one 16-byte piece per thread ------------------
  {
    const int i = tid & 15, gg = (tid >> 4) & 3, blk = tid >> 6, m = blk >> 1, s2 = blk & 1;
    if (!(m < 2 && s2 == 1)) {                       // those two blocks were zero-filled above
      const int row = 16 * m + i, c0 = 32 * s2 + 4 * gg;
      const float gi = s_gam[row];
      float tv[8];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x4 tq = *(const f32x4*)(s_L + row * P_LDF + c0 + 16 * hh);
        const f32x4 gj = *(const f32x4*)(s_gam + c0 + 16 * hh);
#pragma unroll
        for (int c = 0; c < 4; ++c) tv[4 * hh + c] = row >= c0 + 16 * hh + c ? tq[c] * __expf(gi - gj[c]) : 0.f;
      }
      *(u32x4*)(rec + REC_TU + tid * 16) = pack8(tv[0], tv[1], tv[2], tv[3], tv[4], tv[5], tv[6], tv[7]);
    }
  }
  // ---- P4b: w^T = kb^T Tw^T  (transposed so that a lane owns a time row and 32 k-columns): wave -> 32x32 tile
  //           (time tile mi, k tile s);  Wn = -bf16(bf16(w) e^gamma_i) -> fragment blocks ---------------------------
  {
    const int mi = wave_u >> 2, s = wave_u & 3;
    const int i = 32 * mi + l31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks > 2 * mi + 1) continue;                 // Tw is lower triangular: columns > 32 mi + 31 are zero
      const f32x4 ta = *(const f32x4*)(s_L + i * P_LDF + 16 * ks + 8 * hi);
      const f32x4 tb = *(const f32x4*)(s_L + i * P_LDF + 16 * ks + 8 * hi + 4);
      const int j0 = 16 * ks + 8 * hi;
      const u32x4 twf = pack8(i >= j0 ? ta[0] : 0.f, i >= j0 + 1 ? ta[1] : 0.f, i >= j0 + 2 ? ta[2] : 0.f, i >= j0 + 3 ? ta[3] : 0.f,
                              i >= j0 + 4 ? tb[0] : 0.f, i >= j0 + 5 ? tb[1] : 0.f, i >= j0 + 6 ? tb[2] : 0.f, i >= j0 + 7 ? tb[3] : 0.f);
      const u32x4 kbf = frag_tr32(s_kb, P_LDK, 16 * ks, 32 * s, lane);        // A[n = 32s + l31][time 16ks + 8hi + e]
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(kbf), mf(twf), acc, 0, 0, 0);
    }
    const float negeg = -s_eg[i];
    unsigned char* blk = rec + REC_WN + ((2 * mi + (l31 >> 4)) * 4 + s) * 1024;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      *(u32x4*)(blk + ((hi + 2 * p) * 16 + (l31 & 15)) * 16) =
          pack8(bf_round(acc[4 * p]) * negeg, bf_round(acc[4 * p + 1]) * negeg, bf_round(acc[4 * p + 2]) * negeg, bf_round(acc[4 * p + 3]) * negeg,
                bf_round(acc[8 + 4 * p]) * negeg, bf_round(acc[9 + 4 * p]) * negeg, bf_round(acc[10 + 4 * p]) * negeg, bf_round(acc[11 + 4 * p]) * negeg);
  }
  IVL_T(tp4);
  IVL_TOUT(0, tp0); IVL_TOUT(1, tp1 - tp0); IVL_TOUT(2, tp2 - tp1); IVL_TOUT(3, tp3a - tp2); IVL_TOUT(4, tp3b - tp3a);
  IVL_TOUT(5, tp3 - tp3b); IVL_TOUT(6, tp4 - tp3); IVL_TOUT(7, tp4);
}

// ==================================================================================================
// (2) serial scan + output
// ==================================================================================================
constexpr int IMG_V = 66560;                               // per-wave raw v slab [64 t][16 cols] bf16 (2 KB each) follows the record image
__host__ __device__ constexpr int img_bytes(int nw) { return IMG_V + nw * 2048; }

// LDS-DMA, four consecutive 1 KB pieces: global [gsrc + 1024 p + 16 lane] -> LDS [lds_dst + 1024 p + 16 lane], p = 0..3
// (the instruction offset is added to both addresses).  gsrc and lds_dst are wave-uniform (SGPRs); hipcc does not count
// these operations: completion is awaited with an explicit s_waitcnt vmcnt and published by the following barrier.
__device__ __forceinline__ void dma4(const unsigned char* gsrc, unsigned int lds_dst, unsigned int lane16) {
  unsigned int keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
      "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
      "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(lane16), "s"(lds_dst), "s"(gsrc) : "memory");
}
// one piece with per-lane source addresses
__device__ __forceinline__ void dma1(const unsigned char* gsrc_lane, unsigned int lds_dst) {
  unsigned int keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_dst) : "memory");
}
// workgroup barrier that waits for this wave's LDS traffic only (no vector-memory drain)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NW>
__global__ __launch_bounds__(64 * NW) void gdn_chunk_scan_kernel(
    const unsigned char* __restrict__ ws, const bf16_t* __restrict__ v, bf16_t* __restrict__ o,
    const void* h0, int h0_dtype, void* ht, int ht_dtype,
    int T, int H, int t_seg0, int nt_seg, float scale) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int IMG = img_bytes(NW);

  IVL_T(ts0);
  IVL_TVAR(t_wait); IVL_TVAR(t_issue); IVL_TVAR(t_comp);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int j = lane & 15, g = lane >> 4;
  // grid = (B*H, 16/NW): linear block id = bh + B*H*slab, so with B*H % 8 == 0 all slabs of one head run on the same
  // XCD (id % 8) and share its L2 for the record they all read.
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh % H;
  const int v0 = (blockIdx.y * NW + wave_u) * 16;               // first state column of this wave
  const unsigned int lds0 = __builtin_amdgcn_readfirstlane((unsigned int)(size_t)smem);
  const unsigned int lane16 = lane * 16;

  auto issue = [&](int ci) {                                    // stage chunk ci into image ci & 1
    const unsigned char* rec = ws + ((size_t)bh * nt_seg + ci) * REC_STRIDE;
    const unsigned int img = lds0 + (unsigned int)((ci & 1) * IMG);
#pragma unroll
    for (int qq = 0; qq < 16 / NW; ++qq) {
      const int grp = qq * NW + wave_u;                         // 4 KB groups of the 64 KB operand part
      dma4(rec + grp * 4096, img + (unsigned int)(grp * 4096), lane16);
    }
    if (wave_u == 0) dma1(rec + REC_EG + lane16, img + (unsigned int)REC_EG);
    const int tc = t_seg0 + ci * GC;
#pragma unroll
    for (int p = 0; p < 2; ++p) {                               // raw v slab: token 32p + lane/2, 16-byte half lane&1
      const int tok = min(tc + 32 * p + (lane >> 1), T - 1);
      const bf16_t* src = v + (((size_t)b * T + tok) * H + h) * GV + v0 + 8 * (lane & 1);
      dma1((const unsigned char*)src, img + (unsigned int)(IMG_V + wave_u * 2048 + p * 1024));
    }
  };
  issue(0);

  // state slab: tile t = rows 16t..16t+15, lane (g, j) register r <-> S[16t + 4g + r][v0 + j]
  f32x4 S[8];
  {
    const size_t base = ((size_t)bh * GK + 4 * g) * GV + v0 + j;
    if (h0 == nullptr) {
#pragma unroll
      for (int t = 0; t < 8; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if (h0_dtype == IVL_F32) {
      const float* hp = (const float*)h0 + base;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[t][r] = hp[(size_t)(16 * t + r) * GV];
    } else {
      const bf16_t* hp = (const bf16_t*)h0 + base;
      bf16_t raw[32];
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) raw[4 * t + r] = hp[(size_t)(16 * t + r) * GV];
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[t][r] = bf2f(raw[4 * t + r]);
    }
  }

  for (int ci = 0; ci < nt_seg; ++ci) {
    const unsigned char* img = smem + (ci & 1) * IMG;
    const int tc0 = t_seg0 + ci * GC;
    IVL_T(tc_a);
    // This wave's pieces of chunk ci have landed: everything but the 4 output stores of the previous chunk, which
    // were issued after them (vmcnt retires in issue order).
    if (ci == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    lds_barrier();                       // ... and so have everyone else's; every wave has left chunk ci - 1
    IVL_T(tc_b);
    if (ci + 1 < nt_seg) issue(ci + 1);  // overwrites the image of chunk ci - 1
    IVL_T(tc_c);

    auto blk = [&](int off, int idx) { return mf(*(const u32x4*)(img + off + idx * 1024 + lane16)); };

    // ---- B-operand fragments: the state (bf16) ----------------------------------------------------------------
    u32x4 sb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
      sb[s] = pack8(S[2 * s][0], S[2 * s][1], S[2 * s][2], S[2 * s][3], S[2 * s + 1][0], S[2 * s + 1][1], S[2 * s + 1][2], S[2 * s + 1][3]);
    // ---- bf16(beta v): LDS transpose read of the raw slab [time][16 cols]: lane (g, j) <- times 32 s2 + {4g.., 16+4g..}
    u32x4 vb[2];
    {
      const unsigned char* vimg = img + IMG_V + wave * 2048;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int time0 = 32 * s2 + 4 * g;
        const unsigned char* p = vimg + (time0 + (j >> 2)) * 32 + 8 * (j & 3);
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * 32));
        const f32x4 b0 = *(const f32x4*)(img + REC_BETA + time0 * 4), b1 = *(const f32x4*)(img + REC_BETA + (time0 + 16) * 4);
        u32x2 w0, w1;
        __builtin_memcpy(&w0, &a0, 8);
        __builtin_memcpy(&w1, &a1, 8);
        vb[s2] = pack8(bflo(w0.x) * b0[0], bfhi(w0.x) * b0[1], bflo(w0.y) * b0[2], bfhi(w0.y) * b0[3],
                       bflo(w1.x) * b1[0], bfhi(w1.x) * b1[1], bflo(w1.y) * b1[2], bfhi(w1.y) * b1[3]);
      }
    }
    // ---- u = Tu (beta v);  v_new = u + Wn S   (time tiles m, lane (g, j) register r <-> time 16m + 4g + r, column j) ---
    f32x4 accV[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) accV[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m < 2 && s2 == 1) continue;                 // strictly upper blocks of Tu
        accV[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blk(REC_TU, 2 * m + s2), mf(vb[s2]), accV[m], 0, 0, 0);
      }
#pragma unroll
    for (int m = 0; m < 4; ++m)                         // the reference keeps u in bf16 (wy_fast.py:341-343): same rounding point
#pragma unroll
      for (int r = 0; r < 4; ++r) accV[m][r] = bf_round(accV[m][r]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        accV[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blk(REC_WN, 4 * m + s), mf(sb[s]), accV[m], 0, 0, 0);
    // ---- (q_hat S)^T, independent of v_new: fills the MFMA pipe while v_new is converted ------------------------
    f32x4 accO[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) accO[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        accO[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf(sb[s]), blk(REC_QH, 4 * m + s), accO[m], 0, 0, 0);
    u32x4 vn[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      vn[s2] = pack8(accV[2 * s2][0], accV[2 * s2][1], accV[2 * s2][2], accV[2 * s2][3],
                     accV[2 * s2 + 1][0], accV[2 * s2 + 1][1], accV[2 * s2 + 1][2], accV[2 * s2 + 1][3]);
    // ---- S = e^{gamma_L} S + Kd^T v_new ----------------------------------------------------------------------------
    {
      const float egl = *(const float*)(img + REC_EGL);
#pragma unroll
      for (int t = 0; t < 8; ++t) S[t] *= egl;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int t = 0; t < 8; ++t)
        S[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blk(REC_KDT, 2 * t + s2), mf(vn[s2]), S[t], 0, 0, 0);
    // ---- o^T = scale ((q_hat S)^T e^gamma + v_new^T Aqk^T): lane (g, j) register r <-> column v0 + 4g + r, time 16m + j --
#pragma unroll
    for (int m = 0; m < 4; ++m) accO[m] *= *(const float*)(img + REC_EG + (16 * m + j) * 4);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m < 2 && s2 == 1) continue;                 // strictly upper blocks of Aqk
        accO[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf(vn[s2]), blk(REC_AQK, 2 * m + s2), accO[m], 0, 0, 0);
      }
    bf16_t* orow = o + (((size_t)b * T + tc0 + j) * H + h) * GV + v0 + 4 * g;
    if (tc0 + GC <= T) {                                // full chunk: exactly 4 stores (counted by the vmcnt(4) above)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        *(u32x2*)(orow + (size_t)16 * m * H * GV) =
            u32x2{pack2bf(accO[m][0] * scale, accO[m][1] * scale), pack2bf(accO[m][2] * scale, accO[m][3] * scale)};
    } else {                                            // zero-padded tail chunk (always the last one)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (tc0 + 16 * m + j < T)
          *(u32x2*)(orow + (size_t)16 * m * H * GV) =
              u32x2{pack2bf(accO[m][0] * scale, accO[m][1] * scale), pack2bf(accO[m][2] * scale, accO[m][3] * scale)};
    }
    IVL_T(tc_d);
    IVL_TACC(t_wait, tc_b, tc_a); IVL_TACC(t_issue, tc_c, tc_b); IVL_TACC(t_comp, tc_d, tc_c);
  }
  IVL_T(ts1);

  if (ht != nullptr) {
    const size_t base = ((size_t)bh * GK + 4 * g) * GV + v0 + j;
    if (ht_dtype == IVL_F32) {
      float* hp = (float*)ht + base;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[(size_t)(16 * t + r) * GV] = S[t][r];
    } else {
      bf16_t* hp = (bf16_t*)ht + base;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[(size_t)(16 * t + r) * GV] = f2bf(S[t][r]);
    }
  }
  IVL_T(ts2);
  IVL_TOUT(16, ts0); IVL_TOUT(17, t_wait); IVL_TOUT(18, t_issue); IVL_TOUT(19, t_comp); IVL_TOUT(20, ts1 - ts0); IVL_TOUT(21, ts2 - ts1);
  IVL_TOUT(22, ts2);
}

#ifdef IVL_TRACE
int g_scan_nw = 4;                         // developer knob (trace build only): waves per scan workgroup, 2 or 4
#endif

}  // namespace ivl

using namespace ivl;

static inline int seg_chunks(int NT) { return NT < G_SEG_CHUNKS ? NT : G_SEG_CHUNKS; }

// dynamic-LDS opt-in, once per device (hipFuncSetAttribute acts on the current device)
static void gdn_chunk_init_device() {
  static std::once_flag once[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::call_once(once[dev & 63], [] {
    (void)hipFuncSetAttribute((const void*)gdn_chunk_prepare_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P_BYTES);
    (void)hipFuncSetAttribute((const void*)gdn_chunk_scan_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * img_bytes(4));
    (void)hipFuncSetAttribute((const void*)gdn_chunk_scan_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * img_bytes(2));
  });
}

extern "C" size_t ivl_gdn_chunk_workspace_bytes(int B, int T, int H, int K, int V) {
  if (B <= 0 || T <= 0 || H <= 0 || K != GK || V != GV) return 0;
  const int NT = (T + GC - 1) / GC;
  size_t bytes = (size_t)B * H * seg_chunks(NT) * REC_STRIDE;
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
  gdn_chunk_init_device();
  hipStream_t st = (hipStream_t)stream;
  const int NT = (T + GC - 1) / GC;
  const int segc = seg_chunks(NT);
  unsigned char* wsb = (unsigned char*)workspace;
  float* carry = NT > G_SEG_CHUNKS ? (float*)(wsb + (size_t)B * H * segc * REC_STRIDE) : nullptr;
  int nw = 4;
#ifdef IVL_TRACE
  nw = g_scan_nw;
#endif
  for (int c0 = 0; c0 < NT; c0 += segc) {
    const int nseg = (NT - c0) < segc ? (NT - c0) : segc;
    const bool first = c0 == 0, last = c0 + nseg >= NT;
    hipLaunchKernelGGL(gdn_chunk_prepare_kernel, dim3(nseg, B * H), dim3(512), P_BYTES, st,
                       (const bf16_t*)q, (const bf16_t*)k, g, (const bf16_t*)beta, wsb, T, H, c0 * GC, nseg, use_qk_l2norm);
    int rc = check_launch("ivl_gdn_chunk_fwd(prepare)");
    if (rc != IVL_OK) return rc;
    const void* hin = first ? h0 : (const void*)carry;
    const int hin_dt = first ? h0_dtype : IVL_F32;
    void* hout = last ? ht : (void*)carry;
    const int hout_dt = last ? ht_dtype : IVL_F32;
    if (nw == 2)
      hipLaunchKernelGGL((gdn_chunk_scan_kernel<2>), dim3(B * H, 8), dim3(128), 2 * img_bytes(2), st, (const unsigned char*)wsb,
                         (const bf16_t*)v, (bf16_t*)o, hin, hin_dt, hout, hout_dt, T, H, c0 * GC, nseg, scale);
    else
      hipLaunchKernelGGL((gdn_chunk_scan_kernel<4>), dim3(B * H, 4), dim3(256), 2 * img_bytes(4), st, (const unsigned char*)wsb,
                         (const bf16_t*)v, (bf16_t*)o, hin, hin_dt, hout, hout_dt, T, H, c0 * GC, nseg, scale);
    rc = check_launch("ivl_gdn_chunk_fwd(scan)");
    if (rc != IVL_OK) return rc;
  }
  return IVL_OK;
}
