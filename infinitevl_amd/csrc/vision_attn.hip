// Vision-tower window attention for gfx950 (second half).
//
// Replaces, per InfiniteVLVisionAttention.forward (strm:713-801 = std:583-668):
//   apply_rotary_pos_emb_vision (strm:657-671: fp32 products and sum, rounded back to bf16)
//   + the per-window loop over attention_interface / the flash-attention varlen call (strm:752-796):
//     NON-causal softmax attention inside each segment [cu_seqlens[s], cu_seqlens[s+1]) of one packed sequence of
//     patches; 16 heads x 80 channels in the 3B model (window layers: <= 64 patches per segment, the four "full"
//     layers: one segment per frame).
//
// Same construction as the 64-row sliding-window kernel (swa.hip: S^T = K Q^T and O^T = V^T P^T on
// v_mfma_f32_16x16x32_bf16, online softmax with lane-local rows, P^T packed in place), minus everything a ViT does not
// have (ring cache, band, GQA, split-KV), plus a head dimension that is not a multiple of 32:
//   D = 80 : the QK contraction runs over 96 channels (the 16-byte pieces 10, 11 of every K row in LDS and of every Q
//            fragment are zero: 3 MFMA steps instead of 2.5), the PV product over exactly 5 column tiles of 16.
// Workgroup = 4 waves x 16 query rows of one (head, segment, 64-row tile); the 1-D grid enumerates them with
// ceil(max_seqlen / 64) tiles per segment (tiles beyond a segment's length exit at once: the host never reads
// cu_seqlens, so the call is graph-capturable), mapped onto the XCDs so that the tiles of one (segment, head) share an L2.
#include "ivl_common.h"

namespace ivl {

typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

IVL_TRACE_DECL(vis)

constexpr float VA_LOG2E = 1.4426950408889634f;
constexpr int VA_QT = 64, VA_KT = 64;

struct VisionAttnParams {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
  long long q_st, q_sh, k_st, k_sh, v_st, v_sh, o_st, o_sh;      // element strides: token, head
  const int* cu; int n_seg, tiles_per_seg, H;
  const float* rcos; const float* rsin;                          // [S, D] fp32 or NULL (inputs already rotated)
  int k_rotated;                                                 // the keys come from the rotary pre-pass: only q is rotated in the kernel
  float scaling;
};

__device__ __forceinline__ mfma_bf16x8 va_mfma(u32x4 v) {
  mfma_bf16x8 r;
  __builtin_memcpy(&r, &v, 16);
  return r;
}
__device__ __forceinline__ float va_max2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float va_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// butterflies over the four 16-lane groups of a wave (the lanes that share a query row)
__device__ __forceinline__ float va_group_max(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = va_max2(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return va_max2(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float va_group_sum(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// apply_rotary_pos_emb_vision on one 16-byte piece (8 channels c0..c0+7 of one token and head):
//   out = x * cos + rotate_half(x) * sin,  rotate_half(x)[c] = -x[c + D/2] (c < D/2) | x[c - D/2] (c >= D/2)
// `own` = the piece, `part` = the piece D/2 channels away, cos / sin = fp32 rows of the token at c0.  Each product and the
// sum are rounded separately (no fused multiply-add): the reference does q.float() * cos, rotate_half(q) * sin and the
// addition as three fp32 tensor operations and rounds the result to bf16 once (strm:665-670) - bit-identical.
// (the contraction into v_fma is switched off per function: `__fmul_rn` / `__fadd_rn` are plain operators in the HIP headers
// and contract like any other once inlined)
__device__ __forceinline__ u32x4 vision_rope_piece(u32x4 own, u32x4 part, const float* cosp, const float* sinp, bool lower_half) {
#pragma clang fp contract(off)
  const f32x4 c0 = *(const f32x4*)cosp, c1 = *(const f32x4*)(cosp + 4);
  const f32x4 s0 = *(const f32x4*)sinp, s1 = *(const f32x4*)(sinp + 4);
  const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
  const float sn[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
  const unsigned int xo[4] = {own.x, own.y, own.z, own.w}, xp[4] = {part.x, part.y, part.z, part.w};
  unsigned int out[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a0 = bflo(xo[i]), a1 = bfhi(xo[i]);
    const float r0 = lower_half ? -bflo(xp[i]) : bflo(xp[i]), r1 = lower_half ? -bfhi(xp[i]) : bfhi(xp[i]);
    out[i] = pack2bf(a0 * cs[2 * i] + r0 * sn[2 * i], a1 * cs[2 * i + 1] + r1 * sn[2 * i + 1]);
  }
  return u32x4{out[0], out[1], out[2], out[3]};
}

// QG = 16-row query groups per wave: 1 -> 64 rows per workgroup; 2 -> 128 rows, every K / V^T fragment read from LDS feeds
// two MFMAs and a staged tile serves twice the rows (chosen by the host when the 128-row grid still fills the chip).
template <int D, int QG>
__global__ __launch_bounds__(256, 2) void vision_attn_kernel(VisionAttnParams p) {
  static_assert(D % 16 == 0 && D <= 128, "head_dim: a multiple of 16, at most 128");
  constexpr int QT = VA_QT * QG;               // query rows per workgroup
  constexpr int NCH = D / 8;                   // 16-byte pieces per row
  constexpr int DK = (D + 31) / 32 * 32;       // contraction length of the QK product (zero padded)
  constexpr int NKS = DK / 32;                 // QK MFMA steps
  constexpr int NDT = D / 16;                  // PV output column tiles
  // K rows of 256 B with the 16-byte pieces XOR-swizzled by the row (piece' = piece ^ (row & 15), as in swa.hip: the
  // ds_read_b128 lane groups {0-3,12-15,20-27}, ... see distinct banks; a padded 208-byte row measured 37 % of the LDS
  // cycles as bank conflicts).  V rows: ds_read_b64_tr_b16 serves 8 rows x 32 B per 32-lane group, conflict-free when the
  // row stride is an odd multiple of 32 B: 160 B (D = 80) as it is, 128 / 256 B rows get 32 B of padding.
  constexpr int KS = 256;
  constexpr int VS = (D * 2 / 32) % 2 == 1 ? D * 2 : D * 2 + 32;
  constexpr int LDS_K = VA_KT * KS;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_K + VA_KT * VS];

  IVL_T(tv_start);
  IVL_TVAR(tv_b1); IVL_TVAR(tv_st); IVL_TVAR(tv_qk); IVL_TVAR(tv_sm); IVL_TVAR(tv_pv);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  // 1-D grid, XCD-aware: hardware block i runs on XCD i % 8; logical ids are ordered (segment, head, query tile), and XCD x
  // takes the contiguous range [x N/8, (x+1) N/8) of them: the query tiles of one (segment, head) - which all read the same
  // K / V - and the heads of one segment - which share its rotary tables and the cache lines of the qkv rows - meet in one
  // L2 (tiles dealt round-robin made each of the 8 L2s fetch every K / V of a full layer; heads dealt round-robin made the
  // window layers 1.5 x slower)
  int lid = blockIdx.x;
  if ((gridDim.x & 7) == 0) lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = lid % p.tiles_per_seg;
  const int h = (lid / p.tiles_per_seg) % p.H;
  const int seg = lid / (p.tiles_per_seg * p.H);
  const int seg0 = p.cu[seg];
  const int len = p.cu[seg + 1] - seg0;
  if (tile * QT >= len) return;                // workgroup-uniform (before any barrier)

  bool row_ok[QG];
  long long tok_q[QG];
  // ---- Q fragments (B operand of S^T = K Q^T): lane = query row, k-slots 8g..8g+7 of each 32-chunk -----------------
  u32x4 qf[QG][NKS];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    const int row = tile * QT + (wave * QG + qg) * 16 + l15;          // query row inside the segment
    row_ok[qg] = row < len;
    tok_q[qg] = seg0 + (row_ok[qg] ? row : len - 1);
    const bf16_t* qp = p.q + tok_q[qg] * p.q_st + (long long)h * p.q_sh;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int ch = 4 * ks + g;
      if (ch < NCH) {
        qf[qg][ks] = *(const u32x4*)(qp + ch * 8);
        if (p.rcos != nullptr) {
          const int pch = ch < NCH / 2 ? ch + NCH / 2 : ch - NCH / 2;
          const u32x4 part = *(const u32x4*)(qp + pch * 8);
          qf[qg][ks] = vision_rope_piece(qf[qg][ks], part, p.rcos + tok_q[qg] * D + ch * 8, p.rsin + tok_q[qg] * D + ch * 8, ch < NCH / 2);
        }
      } else {
        qf[qg][ks] = u32x4{0u, 0u, 0u, 0u};
      }
    }
  }

  // the zero padding of the K rows (pieces NCH .. DK/8 - 1) is written once; the staging never touches it
  if constexpr (DK > D) {
    constexpr int NPAD = DK / 8 - NCH;
    for (int i = tid; i < VA_KT * NPAD; i += 256)
      *(u32x4*)(smem + (i / NPAD) * KS + (((NCH + i % NPAD) ^ ((i / NPAD) & 15)) << 4)) = u32x4{0u, 0u, 0u, 0u};
  }

  float m_run[QG], l_run[QG];
  f32x4 oacc[QG][NDT];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    m_run[qg] = -INFINITY;
    l_run[qg] = 0.f;
#pragma unroll
    for (int i = 0; i < NDT; ++i) oacc[qg][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging: thread -> rows (tid >> 4) + 16 i, 16-byte piece tid & 15 (pieces >= NCH idle) ---------------------
  const int srow = tid >> 4, schunk = tid & 15;
  const bool s_act = schunk < NCH;
  const int spart = schunk < NCH / 2 ? schunk + NCH / 2 : schunk - NCH / 2;
  const bf16_t* kb = p.k + (long long)h * p.k_sh;
  const bf16_t* vb = p.v + (long long)h * p.v_sh;
  u32x4 kreg[4], vreg[4];
  const bool rope_k = p.rcos != nullptr && !p.k_rotated;
  const bf16_t* kp0 = kb + (long long)(seg0 + srow) * p.k_st + schunk * 8;      // this thread's piece of row srow of tile 0
  const bf16_t* vp0 = vb + (long long)(seg0 + srow) * p.v_st + schunk * 8;
  const long long k16 = 16 * p.k_st, v16 = 16 * p.v_st;
  auto load_tile = [&](int kt) {
    if (!s_act) return;
    if (!rope_k && kt * VA_KT + VA_KT <= len) {
      // workgroup-uniform fast path (every tile but the last of a segment, keys already rotated): eight loads off two
      // running pointers, nothing here reads the loaded values (a per-row select or the rotation would make the tile
      // wait for its loads on the spot instead of at the LDS store one iteration later)
      const bf16_t* kp = kp0 + (long long)kt * 4 * k16;
      const bf16_t* vp = vp0 + (long long)kt * 4 * v16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kreg[i] = *(const u32x4*)(kp + i * k16);
        vreg[i] = *(const u32x4*)(vp + i * v16);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = kt * VA_KT + srow + 16 * i;
      const long long tok = seg0 + min(j, len - 1);          // clamped address; rows >= len are zeroed below
      const bf16_t* kp = kb + tok * p.k_st;
      kreg[i] = *(const u32x4*)(kp + schunk * 8);
      vreg[i] = *(const u32x4*)(vb + tok * p.v_st + schunk * 8);
      if (rope_k) {
        const u32x4 part = *(const u32x4*)(kp + spart * 8);
        kreg[i] = vision_rope_piece(kreg[i], part, p.rcos + tok * D + schunk * 8, p.rsin + tok * D + schunk * 8, schunk < NCH / 2);
      }
      if (j >= len) {
        kreg[i] = u32x4{0u, 0u, 0u, 0u};
        vreg[i] = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  auto store_tile = [&]() {
    if (!s_act) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = srow + 16 * i;
      *(u32x4*)(smem + r * KS + ((schunk ^ (r & 15)) << 4)) = kreg[i];
      *(u32x4*)(smem + LDS_K + r * VS + schunk * 16) = vreg[i];
    }
  };

  const int n_kt = (len + VA_KT - 1) / VA_KT;
  load_tile(0);
  const float sc = p.scaling * VA_LOG2E;

  IVL_T(tv_loop);
  for (int kt = 0; kt < n_kt; ++kt) {
    IVL_T(t0);
    __syncthreads();
    IVL_T(t1);
    store_tile();
    __syncthreads();
    IVL_T(t2);

    // ---- S^T = K Q^T : 4 key sub-tiles x NKS channel steps -----------------------------------------------------
    f32x4 sacc[QG][4];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) sacc[qg][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const u32x4 kf = *(const u32x4*)(smem + (16 * mt + l15) * KS + (((4 * ks + g) ^ l15) << 4));
#pragma unroll
        for (int qg = 0; qg < QG; ++qg)
          sacc[qg][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va_mfma(kf), va_mfma(qf[qg][ks]), sacc[qg][mt], 0, 0, 0);
      }
    if (kt + 1 < n_kt) load_tile(kt + 1);      // lands under the softmax and the PV product
    IVL_T(t3);

    // ---- tail mask + online softmax (lane-local rows): lane (g, l15) register r of sub-tile mt <-> key 16 mt + 4 g + r --
    u32x4 pf[QG][2];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
      if (kt * VA_KT + VA_KT > len) {
        const int jbase = kt * VA_KT + 4 * g;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[qg][mt][r] = jbase + 16 * mt + r < len ? sacc[qg][mt][r] : -INFINITY;
      }
      // the FIRST reader of the MFMA results is an instruction the compiler sees (its hazard recognizer does not look
      // inside inline asm; see swa.hip)
      float rmax = va_max2(__builtin_fmaxf(sacc[qg][0][0], sacc[qg][0][1]), sacc[qg][0][2]);
      rmax = va_max3(rmax, sacc[qg][0][3], sacc[qg][1][0]);
      rmax = va_max3(rmax, sacc[qg][1][1], sacc[qg][1][2]);
      rmax = va_max3(rmax, sacc[qg][1][3], sacc[qg][2][0]);
      rmax = va_max3(rmax, sacc[qg][2][1], sacc[qg][2][2]);
      rmax = va_max3(rmax, sacc[qg][2][3], sacc[qg][3][0]);
      rmax = va_max3(rmax, sacc[qg][3][1], sacc[qg][3][2]);
      rmax = va_max2(rmax, sacc[qg][3][3]);
      rmax = va_group_max(rmax) * sc;                          // sc > 0: max commutes with the scale
      const float m_new = va_max2(m_run[qg], rmax);            // every tile holds at least one valid key: finite
      float rsum = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qg][mt][r], sc, -m_new));   // argument <= 0
          sacc[qg][mt][r] = pv;
          rsum += pv;
        }
      rsum = va_group_sum(rsum);
      if (__any(m_new > m_run[qg])) {                          // some row's running maximum moved: rescale (exact)
        const float alpha = __builtin_amdgcn_exp2f(m_run[qg] - m_new);      // m_run = -inf -> 0
        l_run[qg] = l_run[qg] * alpha + rsum;
#pragma unroll
        for (int i = 0; i < NDT; ++i) oacc[qg][i] *= alpha;
      } else {
        l_run[qg] += rsum;
      }
      m_run[qg] = m_new;
      // P^T fragments (B operand): slots 8g+e <-> keys 32 ks2 + 4g + e | 32 ks2 + 16 + 4g + (e - 4)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        pf[qg][ks2].x = pack2bf(sacc[qg][2 * ks2][0], sacc[qg][2 * ks2][1]);
        pf[qg][ks2].y = pack2bf(sacc[qg][2 * ks2][2], sacc[qg][2 * ks2][3]);
        pf[qg][ks2].z = pack2bf(sacc[qg][2 * ks2 + 1][0], sacc[qg][2 * ks2 + 1][1]);
        pf[qg][ks2].w = pack2bf(sacc[qg][2 * ks2 + 1][2], sacc[qg][2 * ks2 + 1][3]);
      }
    }
    IVL_T(t4);
    // ---- O^T += V^T P^T : NDT column tiles x 2 key steps ------------------------------------------------------
    const unsigned char* vbase = smem + LDS_K;
#pragma unroll
    for (int mt2 = 0; mt2 < NDT; ++mt2)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        // 16-lane group g reads the 4x16 block rows (32 ks2 [+16] + 4g .. +3), columns 16 mt2 .. +15; lane i supplies
        // the address of row (i >> 2), columns 4 (i & 3) .. +3 and receives column i
        const int r0 = 32 * ks2 + 4 * g + (l15 >> 2);
        const int cb = (16 * mt2 + 4 * (l15 & 3)) * 2;
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vbase + r0 * VS + cb));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vbase + (r0 + 16) * VS + cb));
        u32x2 w0, w1;
        __builtin_memcpy(&w0, &a0, 8);
        __builtin_memcpy(&w1, &a1, 8);
        const u32x4 vf = u32x4{w0.x, w0.y, w1.x, w1.y};
#pragma unroll
        for (int qg = 0; qg < QG; ++qg)
          oacc[qg][mt2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va_mfma(vf), va_mfma(pf[qg][ks2]), oacc[qg][mt2], 0, 0, 0);
      }
    IVL_T(t5);
    IVL_TACC(tv_b1, t1, t0); IVL_TACC(tv_st, t2, t1); IVL_TACC(tv_qk, t3, t2); IVL_TACC(tv_sm, t4, t3); IVL_TACC(tv_pv, t5, t4);
  }
  IVL_T(tv_end);
  IVL_TOUT(0, tv_loop - tv_start); IVL_TOUT(1, tv_b1); IVL_TOUT(2, tv_st); IVL_TOUT(3, tv_qk); IVL_TOUT(4, tv_sm); IVL_TOUT(5, tv_pv);
  IVL_TOUT(6, tv_end - tv_start); IVL_TOUT(7, n_kt);

  // ---- epilogue: lane owns its rows, channels 16 mt2 + 4 g + r ---------------------------------------------------
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    if (!row_ok[qg]) continue;
    const float inv = 1.0f / l_run[qg];        // l_run >= 1 term: the row's own maximum contributes 2^0
    bf16_t* op = p.o + tok_q[qg] * p.o_st + (long long)h * p.o_sh + 4 * g;
#pragma unroll
    for (int mt2 = 0; mt2 < NDT; ++mt2)
      *(u32x2*)(op + 16 * mt2) = u32x2{pack2bf(oacc[qg][mt2][0] * inv, oacc[qg][mt2][1] * inv), pack2bf(oacc[qg][mt2][2] * inv, oacc[qg][mt2][3] * inv)};
  }
}

// Rotary pre-pass for calls whose segments span several 64-row query tiles (the full-attention layers): every key would otherwise be rotated once per query tile of its segment, and the rotation (partner piece + 64
// bytes of tables per piece) sits between a tile's loads and its LDS store.  Only the KEYS take the extra trip through HBM:
// a query row is used by one workgroup only and stays rotated in its prologue.  One thread per (token, group of 4 heads,
// pair of 16-byte pieces D/2 channels apart): the 4 x 32 bytes of tables are loaded once and serve 4 heads x both pieces of
// the pair (each piece is the other's rotate_half partner)  ->  [S][H][D] bf16.
template <int D>
__global__ __launch_bounds__(256) void vision_rope_prepass_kernel(VisionAttnParams p, bf16_t* __restrict__ out, int S, int H) {
#pragma clang fp contract(off)
  constexpr int NP = D / 16;                   // piece pairs per row
  const int hg = (H + 3) / 4;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n = (long long)S * hg * NP;
  if (idx >= n) return;
  const long long r = idx;
  const int pr = (int)(r % NP);
  const int h0 = (int)((r / NP) % hg) * 4;
  const long long tok = r / ((long long)NP * hg);
  const float* cl = p.rcos + tok * D + pr * 8;
  const float* sl = p.rsin + tok * D + pr * 8;
  const f32x4 cl0 = *(const f32x4*)cl, cl1 = *(const f32x4*)(cl + 4), ch0 = *(const f32x4*)(cl + D / 2), ch1 = *(const f32x4*)(cl + D / 2 + 4);
  const f32x4 sl0 = *(const f32x4*)sl, sl1 = *(const f32x4*)(sl + 4), sh0 = *(const f32x4*)(sl + D / 2), sh1 = *(const f32x4*)(sl + D / 2 + 4);
  const float c_lo[8] = {cl0[0], cl0[1], cl0[2], cl0[3], cl1[0], cl1[1], cl1[2], cl1[3]};
  const float c_hi[8] = {ch0[0], ch0[1], ch0[2], ch0[3], ch1[0], ch1[1], ch1[2], ch1[3]};
  const float s_lo[8] = {sl0[0], sl0[1], sl0[2], sl0[3], sl1[0], sl1[1], sl1[2], sl1[3]};
  const float s_hi[8] = {sh0[0], sh0[1], sh0[2], sh0[3], sh1[0], sh1[1], sh1[2], sh1[3]};
  const bf16_t* src = p.k + tok * p.k_st + pr * 8;
  const long long sh = p.k_sh;
  bf16_t* dst = out + (tok * H) * D + pr * 8;
  u32x4 lo[4], hi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int h = min(h0 + i, H - 1);
    lo[i] = *(const u32x4*)(src + h * sh);
    hi[i] = *(const u32x4*)(src + h * sh + D / 2);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (h0 + i >= H) break;
    const unsigned int xl[4] = {lo[i].x, lo[i].y, lo[i].z, lo[i].w}, xh[4] = {hi[i].x, hi[i].y, hi[i].z, hi[i].w};
    unsigned int ol[4], oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a0 = bflo(xl[e]), a1 = bfhi(xl[e]), b0 = bflo(xh[e]), b1 = bfhi(xh[e]);
      ol[e] = pack2bf(a0 * c_lo[2 * e] + (-b0) * s_lo[2 * e], a1 * c_lo[2 * e + 1] + (-b1) * s_lo[2 * e + 1]);
      oh[e] = pack2bf(b0 * c_hi[2 * e] + a0 * s_hi[2 * e], b1 * c_hi[2 * e + 1] + a1 * s_hi[2 * e + 1]);
    }
    *(u32x4*)(dst + (long long)(h0 + i) * D) = u32x4{ol[0], ol[1], ol[2], ol[3]};
    *(u32x4*)(dst + (long long)(h0 + i) * D + D / 2) = u32x4{oh[0], oh[1], oh[2], oh[3]};
  }
}

}  // namespace ivl

using namespace ivl;

extern "C" size_t ivl_vision_attn_workspace_bytes(int S, int H, int d, int max_seqlen) {
  // one query tile per segment (window layers): nothing is rotated twice, the rotation stays in the tile loads (a pre-pass
  // is a second trip of q and k through HBM: 36 vs 31 us on 8 frames of a window layer)
  if (S <= 0 || H <= 0 || d <= 0 || max_seqlen <= VA_QT) return 0;
  return (size_t)S * H * d * sizeof(bf16_t);            // the rotated keys
}

extern "C" int ivl_vision_attn_fwd(const void* q, const void* k, const void* v, void* o,
                                   int64_t q_st, int64_t q_sh, int64_t k_st, int64_t k_sh, int64_t v_st, int64_t v_sh,
                                   int64_t o_st, int64_t o_sh, const int32_t* cu_seqlens, int n_seg, int max_seqlen,
                                   int S, int H, int d, float scaling, const float* rope_cos, const float* rope_sin,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  IVL_REQUIRE(q && k && v && o && cu_seqlens, IVL_ERR_INVALID_ARG, "ivl_vision_attn_fwd: null pointer");
  IVL_REQUIRE(n_seg > 0 && max_seqlen > 0 && H > 0 && S > 0, IVL_ERR_INVALID_ARG, "ivl_vision_attn_fwd: bad shape S=%d n_seg=%d max_seqlen=%d H=%d", S, n_seg, max_seqlen, H);
  IVL_REQUIRE(d == 64 || d == 80 || d == 128, IVL_ERR_UNSUPPORTED, "ivl_vision_attn_fwd: head_dim %d not built (64, 80, 128)", d);
  IVL_REQUIRE((rope_cos == nullptr) == (rope_sin == nullptr), IVL_ERR_INVALID_ARG, "ivl_vision_attn_fwd: rope_cos and rope_sin go together");
  IVL_REQUIRE(q_st % 8 == 0 && q_sh % 8 == 0 && k_st % 8 == 0 && k_sh % 8 == 0 && v_st % 8 == 0 && v_sh % 8 == 0 && o_st % 4 == 0 && o_sh % 4 == 0,
              IVL_ERR_UNSUPPORTED, "ivl_vision_attn_fwd: strides must keep 16-byte (q, k, v) / 8-byte (o) alignment");
  IVL_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0 && (uintptr_t)o % 8 == 0, IVL_ERR_UNSUPPORTED, "ivl_vision_attn_fwd: misaligned pointer");
  // 128-row workgroups when that grid still gives every CU two of them; otherwise 64 rows (window layers, single frames)
  const int qg = (max_seqlen > VA_QT && (long long)n_seg * ((max_seqlen + 2 * VA_QT - 1) / (2 * VA_QT)) * H >= 512) ? 2 : 1;
  const int tiles = (max_seqlen + VA_QT * qg - 1) / (VA_QT * qg);
  IVL_REQUIRE((long long)n_seg * tiles * H < (1ll << 31), IVL_ERR_UNSUPPORTED, "ivl_vision_attn_fwd: grid too large");
  VisionAttnParams p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
  p.q_st = q_st; p.q_sh = q_sh; p.k_st = k_st; p.k_sh = k_sh; p.v_st = v_st; p.v_sh = v_sh; p.o_st = o_st; p.o_sh = o_sh;
  p.cu = cu_seqlens; p.n_seg = n_seg; p.tiles_per_seg = tiles; p.H = H;
  p.rcos = rope_cos; p.rsin = rope_sin; p.scaling = scaling; p.k_rotated = 0;
  const dim3 grid(n_seg * tiles * H);
  hipStream_t st = (hipStream_t)stream;
  const size_t ws_need = ivl_vision_attn_workspace_bytes(S, H, d, max_seqlen);
  if (rope_cos != nullptr && ws_need > 0 && workspace != nullptr && workspace_bytes >= ws_need) {
    // several query tiles per segment: rotate the keys once (without a workspace the rotation stays in the tile loads:
    // correct, but redone per query tile)
    const long long work = (long long)S * ((H + 3) / 4) * (d / 16);
    const dim3 pg((unsigned int)((work + 255) / 256));
    bf16_t* ws = (bf16_t*)workspace;
    if (d == 80) hipLaunchKernelGGL((vision_rope_prepass_kernel<80>), pg, dim3(256), 0, st, p, ws, S, H);
    else if (d == 64) hipLaunchKernelGGL((vision_rope_prepass_kernel<64>), pg, dim3(256), 0, st, p, ws, S, H);
    else hipLaunchKernelGGL((vision_rope_prepass_kernel<128>), pg, dim3(256), 0, st, p, ws, S, H);
    p.k = ws;
    p.k_st = (long long)H * d; p.k_sh = d;
    p.k_rotated = 1;
  }
  if (qg == 1) {
    if (d == 80) hipLaunchKernelGGL((vision_attn_kernel<80, 1>), grid, dim3(256), 0, st, p);
    else if (d == 64) hipLaunchKernelGGL((vision_attn_kernel<64, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((vision_attn_kernel<128, 1>), grid, dim3(256), 0, st, p);
  } else {
    if (d == 80) hipLaunchKernelGGL((vision_attn_kernel<80, 2>), grid, dim3(256), 0, st, p);
    else if (d == 64) hipLaunchKernelGGL((vision_attn_kernel<64, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((vision_attn_kernel<128, 2>), grid, dim3(256), 0, st, p);
  }
  return check_launch("ivl_vision_attn_fwd");
}
