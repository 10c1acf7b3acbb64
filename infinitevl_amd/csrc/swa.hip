// Sliding-window attention for gfx950: flash-style online softmax on MFMA 16x16x32 bf16, GQA,
// head_dim 128, reading the ring-buffer KV cache in place (no torch.cat) plus the call's new K/V.
//
// Formulation (everything per query row stays inside one lane group; no P round trip through LDS):
//   S^T = K Q^T      A = K tile rows (keys)      B = Q^T            -> lane owns ONE query row
//   O^T = V^T P^T    A = V^T (LDS transpose read) B = P^T (in regs) -> same lane owns that row's O
// A 64-key tile is staged once per workgroup (4 waves x 16 query rows) into LDS: K rows padded to 272 B
// (ds_read_b128 of MFMA A fragments from ONE base register + immediates), V rows padded to 288 B
// (conflict-free ds_read_b64_tr_b16).  The key -> MFMA k-slot map of the PV product is permuted so the
// probabilities produced by the first MFMA feed the second one without any cross-lane movement:
//   slot 8g+e  <->  key 32*ks + 4g + e (e<4)  |  key 32*ks + 16 + 4g + (e-4) (e>=4).
// Band (bit-exact contract): row with token index t sees call-local keys
//   lo = max(0, n_prev + t - W + 1) .. hi = n_prev + t.
// Long key ranges are optionally split across workgroups (flash-decoding); a combine kernel merges.
#include "ivl_common.h"
#include "swa_shared.h"
#include <type_traits>

namespace ivl {

constexpr int SWA_QT = 64;           // query rows per workgroup
constexpr int SWA_KT = 64;           // keys per tile
constexpr int SWA_KSTRIDE = 256;     // bytes per K row in LDS; 16-byte pieces XOR-swizzled by the row (see below)
constexpr int SWA_VSTRIDE = 288;     // bytes per V row in LDS (padded)
constexpr int SWA_LDS_K = SWA_KT * SWA_KSTRIDE;
constexpr int SWA_LDS_BYTES = SWA_LDS_K + SWA_KT * SWA_VSTRIDE;
constexpr int SWA_MAX_SPLIT = 16;        // prefill / split-KV with register-resident combine
constexpr int SWA_MAX_SPLIT_PACK = 64;   // packed decode rows: one 64-key tile per workgroup
IVL_TRACE_DECL(swa)

struct SwaParams {
  const bf16_t* q; const bf16_t* k_new; const bf16_t* v_new; const bf16_t* k_cache; const bf16_t* v_cache;
  bf16_t* o;
  long long q_sb, q_st, q_sh, kn_sb, kn_st, kn_sh;
  long long vn_sb, vn_st, vn_sh;           // strides of v_new (= kn_* unless the keys come from the rope pre-pass copy)
  int B, T, T_new, Hq, Hkv, C, W, nsplit, n_qtiles;
  long long pos; const long long* pos_dev;
  float scaling;
  float* part_o; float* part_ml;
  // M-RoPE fused into the Q load and the staging of the call's new keys (NULL: inputs are already rotated)
  const bf16_t* rcos; const bf16_t* rsin; int rs0, rs1;
};

// QG = 16-row query groups per wave (1: 64 rows per workgroup, decode / short calls; 2: 128 rows per workgroup).
// With QG = 2 every K / V^T fragment read from LDS feeds two MFMAs: at QG = 1 the 4 waves of a workgroup pull
// 128 KB through the 128 B/clk LDS port per 64-key tile (1024 clk) for 512 clk of MFMA work per SIMD.
// butterfly over the four 16-lane groups of a wave (the lanes that share a query row) on the gfx950 lane-swap
// instructions: v_permlane16_swap exchanges odd rows of one operand with even rows of the other, v_permlane32_swap
// the upper half with the lower half; with both operands = x the two results hold x and its partner.
__device__ __forceinline__ float group_max(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = vmax2(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return vmax2(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float group_sum(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <bool PACK, int QG>
__global__ __launch_bounds__(256, 2) void swa_fwd_kernel(const long long* pos_dev, SwaParams p) {   // (pos_dev: see swa_prefill_kernel)
  __shared__ __attribute__((aligned(16))) unsigned char smem[SWA_LDS_BYTES];
  constexpr int QT = SWA_QT * QG;      // query rows per workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int G = p.Hq / p.Hkv;
  // 1-D grid with an XCD-aware (bijective) remap: hardware block id i runs on XCD i % 8; logical ids are
  // ordered (b, split, kv-head, head-in-group, q-tile) so the workgroups that read the SAME K/V range are
  // consecutive and therefore land on the same XCD / L2 (they re-read each K/V tile up to 8 x n_qtiles times).
  const int heads_y = PACK ? p.Hkv : p.Hq;
  int bx, rest;                      // q-tile, and (b * nsplit + split) * heads_y + head
  {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qn = nwg >> 3, rn = nwg & 7;
    if (rn == 0 && qn % p.n_qtiles == 0) {
      // every XCD owns whole (batch, split, head) rows.  Inside an XCD the workgroups are issued heaviest first
      // (the work of a q-tile grows with its index: causal / window not yet full), so the long workgroups no
      // longer form the tail.  When all of them are co-resident (<= 2 per CU) the upper half goes first in
      // descending order and the lower half follows in ascending order: a heavy one shares its CU with a light one.
      const int hx = qn / p.n_qtiles;                       // rows per XCD
      const int r = slot / hx, half = (p.n_qtiles + 1) >> 1;
      bx = qn > 64 ? p.n_qtiles - 1 - r : (r < half ? p.n_qtiles - 1 - r : r - half);
      rest = xcd * hx + slot % hx;
    } else {
      const int lid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
      bx = lid % p.n_qtiles;
      rest = lid / p.n_qtiles;
    }
  }
  const int by = rest % heads_y;                         // q head (or kv head when PACK); heads of one group adjacent
  const int bz = rest / heads_y;                         // b * nsplit + split
  const int b = bz / p.nsplit, split = bz % p.nsplit;
  const int hk = PACK ? by : by / G;
  IVL_T(tr_start);
#ifdef IVL_TRACE
  const long long rt_start = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  IVL_TVAR(tr_b1); IVL_TVAR(tr_st); IVL_TVAR(tr_qk); IVL_TVAR(tr_sm); IVL_TVAR(tr_pv);

  const long long pos = pos_dev ? *pos_dev : p.pos;
  const int n_ring = p.C > 0 ? (int)(pos < (long long)p.C ? pos : (long long)p.C) : 0;
  const int n_extra = p.T_new - p.T;
  const int n_prev = n_ring + n_extra;
  const int S = n_prev + p.T;
  const int s0 = p.C > 0 ? mod_pos(pos - n_ring, p.C) : 0;      // ring slot of call-local key 0

  // ---- rows of this workgroup / wave / lane ----------------------------------------------------
  const int total_rows = PACK ? p.T * G : p.T;
  const int tile_row0 = bx * QT;
  int t_row[QG], hq[QG], hi[QG], lo[QG];
  bool row_ok[QG];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    const int row = tile_row0 + (wave * QG + qg) * 16 + l15;
    row_ok[qg] = row < total_rows;
    t_row[qg] = PACK ? row / G : row;
    hq[qg] = PACK ? hk * G + row % G : by;
    hi[qg] = n_prev + t_row[qg];
    lo[qg] = p.W > 0 ? max(0, n_prev + t_row[qg] - p.W + 1) : 0;
  }

  // band extremes over the rows of THIS wave (rows are consecutive; lo/hi are monotone in the row index)
  int w_lo_max, w_hi_min;
  {
    const int wr0 = tile_row0 + wave * 16 * QG;
    const int wr1 = min(wr0 + 16 * QG - 1, total_rows - 1);
    const int t_first = PACK ? wr0 / G : wr0;
    const int t_last = PACK ? max(wr1, wr0) / G : max(wr1, wr0);
    w_hi_min = n_prev + t_first;
    w_lo_max = p.W > 0 ? max(0, n_prev + t_last - p.W + 1) : 0;
  }

  // workgroup key-tile range
  const int last_row = min(tile_row0 + QT, total_rows) - 1;
  const int t_min = PACK ? tile_row0 / G : tile_row0;
  const int t_max = PACK ? last_row / G : last_row;
  const int lo_min = p.W > 0 ? max(0, n_prev + t_min - p.W + 1) : 0;
  const int kt0 = lo_min / SWA_KT, kt1 = (n_prev + t_max) / SWA_KT + 1;
  const int per = (kt1 - kt0 + p.nsplit - 1) / p.nsplit;
  const int kt_begin = kt0 + split * per;
  const int kt_end = min(kt1, kt_begin + per);

  // ---- Q fragments (B operand of S^T = K Q^T): lane = query row, k-slots 8g..8g+7 of each 32-chunk ----
  u32x4 qf[QG][4];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    const bf16_t* qp = p.q + (long long)b * p.q_sb + (long long)t_row[qg] * p.q_st + (long long)hq[qg] * p.q_sh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (row_ok[qg]) qf[qg][ks] = *(const u32x4*)(qp + 32 * ks + 8 * g);
      else qf[qg][ks] = u32x4{0u, 0u, 0u, 0u};
    }
  }

  const long long rplane = (long long)p.B * p.T * SWA_D;
  if (p.rcos != nullptr) {
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
      const long long row_off = ((long long)b * p.T + min(t_row[qg], p.T - 1)) * SWA_D;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) rope_pair(qf[qg][ks], qf[qg][ks + 2], p.rcos, p.rsin, rplane, row_off, 32 * ks + 8 * g, p.rs0, p.rs1);
    }
  }
  float m_run[QG], l_run[QG];
  f32x4 oacc[QG][8];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    m_run[qg] = -INFINITY;
    l_run[qg] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) oacc[qg][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging: thread -> rows (tid>>4) + 16 i, 16-byte chunk tid&15 ------------------------------
  const int srow = tid >> 4, schunk = tid & 15;
  u32x4 kreg[4], vreg[4];
  // per-(batch, kv-head) base pointers; rows are addressed with 32-bit element offsets from them
  const bf16_t* kb_ring = p.C > 0 ? p.k_cache + (((long long)b * p.Hkv + hk) * p.C) * SWA_D + schunk * 8 : p.k_new;
  const bf16_t* vb_ring = p.C > 0 ? p.v_cache + (((long long)b * p.Hkv + hk) * p.C) * SWA_D + schunk * 8 : p.v_new;
  const bf16_t* kb_new = p.k_new + (long long)b * p.kn_sb + (long long)hk * p.kn_sh + schunk * 8;
  const bf16_t* vb_new = p.v_new + (long long)b * p.kn_sb + (long long)hk * p.kn_sh + schunk * 8;
  const unsigned int kn_st32 = (unsigned int)p.kn_st;
  // the call's new keys arrive un-rotated when the rope is fused: thread (row, 16-byte chunk) fetches the partner chunk
  // (channels +-64) and the row's cos / sin and rotates its chunk in place (only the few tiles that hold new keys pay this)
  auto rope_new_key = [&](u32x4& kv, int jn /* index among the new keys */) {
    const int lo_ch = (schunk & 7) * 8;                    // channel block of the "lo" half of the pair
    const u32x4 part = *(const u32x4*)(kb_new - schunk * 8 + (unsigned int)jn * kn_st32 + (schunk ^ 8) * 8);
    u32x4 lo = schunk < 8 ? kv : part, hi = schunk < 8 ? part : kv;
    rope_pair(lo, hi, p.rcos, p.rsin, rplane, ((long long)b * p.T + jn) * SWA_D, lo_ch, p.rs0, p.rs1);
    kv = schunk < 8 ? lo : hi;
  };
  auto load_tile = [&](int kt) {
    const int j0 = kt * SWA_KT;
    const int slot0 = s0 + j0;
    if (j0 + SWA_KT <= n_ring && (slot0 + SWA_KT <= p.C || slot0 >= p.C)) {
      // wave-uniform fast path (almost every tile of a full window): 64 consecutive ring slots, no wrap inside
      const unsigned int off = (unsigned int)((slot0 >= p.C ? slot0 - p.C : slot0) + srow) * SWA_D;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kreg[i] = *(const u32x4*)(kb_ring + off + 16 * i * SWA_D);
        vreg[i] = *(const u32x4*)(vb_ring + off + 16 * i * SWA_D);
      }
      return;
    }
    if (j0 >= n_ring && j0 + SWA_KT <= S) {
      // wave-uniform fast path: 64 keys of this call
      const unsigned int off = (unsigned int)(j0 - n_ring + srow) * kn_st32;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kreg[i] = *(const u32x4*)(kb_new + off + 16 * i * kn_st32);
        vreg[i] = *(const u32x4*)(vb_new + off + 16 * i * kn_st32);
      }
      if (p.rcos != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rope_new_key(kreg[i], j0 - n_ring + srow + 16 * i);
      }
      return;
    }
    // generic tile (ring wrap, ring/new seam or tail).  Branch-free per row: every row issues its two 16-byte
    // loads (clamped address); a conditional load per row would serialise one memory round trip per row.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = j0 + srow + 16 * i;
      const int jc = min(j, S - 1);
      const bool in_ring = jc < n_ring;
      int slot = s0 + jc;
      slot = slot >= p.C ? slot - p.C : slot;
      const unsigned int off = in_ring ? (unsigned int)slot * SWA_D : (unsigned int)(jc - n_ring) * kn_st32;
      const bf16_t* kp = (in_ring ? kb_ring : kb_new) + off;
      const bf16_t* vp = (in_ring ? vb_ring : vb_new) + off;
      kreg[i] = *(const u32x4*)kp;
      vreg[i] = *(const u32x4*)vp;
      if (p.rcos != nullptr && !in_ring) rope_new_key(kreg[i], jc - n_ring);
    }
    if (j0 + SWA_KT > S) {          // wave-uniform: tail tile, rows >= S are zeroed
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (j0 + srow + 16 * i >= S) {
          kreg[i] = u32x4{0u, 0u, 0u, 0u};
          vreg[i] = u32x4{0u, 0u, 0u, 0u};
        }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = srow + 16 * i;
      *(u32x4*)(smem + r * SWA_KSTRIDE + ((schunk ^ (r & 15)) << 4)) = kreg[i];
      *(u32x4*)(smem + SWA_LDS_K + r * SWA_VSTRIDE + schunk * 16) = vreg[i];
    }
  };

  if (kt_begin < kt_end) load_tile(kt_begin);
  const float sc = p.scaling * LOG2E;
  IVL_T(tr_loop);

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    IVL_T(tr0);
    __syncthreads();
    IVL_T(tr1);
    store_tile();
    __syncthreads();
    IVL_T(tr2);

    // ---- S^T = K Q^T : 4 key sub-tiles x 4 d-steps; d-step outermost so that consecutive MFMAs go to
    //      independent accumulators (no back-to-back dependent issue) -------------------------------------
    f32x4 sacc[QG][4];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) sacc[qg][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        // piece' = piece ^ (row & 15): ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31},
        // ... (MI355X_MICROARCH.md, LDS); a padded 272-byte stride puts two lanes of every group on the same banks
        // (SQ_LDS_BANK_CONFLICT = 25 % of the LDS cycles), the XOR image is conflict-free for this fragment shape
        const u32x4 kf = *(const u32x4*)(smem + (16 * mt + l15) * SWA_KSTRIDE + (((4 * ks + g) ^ l15) << 4));
#pragma unroll
        for (int qg = 0; qg < QG; ++qg)
          sacc[qg][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_mfma(kf), as_mfma(qf[qg][ks]), sacc[qg][mt], 0, 0, 0);
      }
    }
    // next tile's global loads are issued behind the first MFMA batch (their address arithmetic no longer
    // delays it); they have the softmax + PV phases to land before store_tile of the next iteration
    if (kt + 1 < kt_end) load_tile(kt + 1);
    IVL_T(tr3);
    // ---- band mask + online softmax (lane-local rows) -----------------------------------------
    // Interior tiles (every key visible to every row of this wave) skip the per-element band test.
    const int jbase = kt * SWA_KT + 4 * g;
    const bool interior = kt * SWA_KT >= w_lo_max && kt * SWA_KT + SWA_KT - 1 <= w_hi_min;
    u32x4 pf[QG][2];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
      // scores stay raw; the softmax scale is folded into the exponent: p = 2^(s*sc - m), m tracked in scaled units
      if (!interior) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = jbase + 16 * mt + r;
            const bool vis = row_ok[qg] && j >= lo[qg] && j <= hi[qg];
            sacc[qg][mt][r] = vis ? sacc[qg][mt][r] : -INFINITY;
          }
      }
      // the FIRST reader of the MFMA results must be an instruction the compiler sees: its hazard recognizer inserts the
      // wait states between an MFMA and a dependent VALU read, but does not look inside inline asm (on an interior tile
      // the asm v_max3 would otherwise follow the last MFMA directly and read the accumulators too early)
      float rmax = vmax2(__builtin_fmaxf(sacc[qg][0][0], sacc[qg][0][1]), sacc[qg][0][2]);
      rmax = vmax3(rmax, sacc[qg][0][3], sacc[qg][1][0]);
      rmax = vmax3(rmax, sacc[qg][1][1], sacc[qg][1][2]);
      rmax = vmax3(rmax, sacc[qg][1][3], sacc[qg][2][0]);
      rmax = vmax3(rmax, sacc[qg][2][1], sacc[qg][2][2]);
      rmax = vmax3(rmax, sacc[qg][2][3], sacc[qg][3][0]);
      rmax = vmax3(rmax, sacc[qg][3][1], sacc[qg][3][2]);
      rmax = vmax2(rmax, sacc[qg][3][3]);
      rmax = group_max(rmax) * sc;                             // sc > 0: max commutes with the scale
      const float m_new = vmax2(m_run[qg], rmax);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;
      float rsum = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qg][mt][r], sc, -m_use));   // argument <= 0
          sacc[qg][mt][r] = pv;
          rsum += pv;
        }
      rsum = group_sum(rsum);
      if (__any(m_new > m_run[qg])) {                        // some row's running max moved: rescale (exact)
        const float alpha = __builtin_amdgcn_exp2f(m_run[qg] - m_use);      // m_run = -inf -> 0
        l_run[qg] = l_run[qg] * alpha + rsum;
#pragma unroll
        for (int i = 0; i < 8; ++i) oacc[qg][i] *= alpha;
      } else {
        l_run[qg] += rsum;
      }
      m_run[qg] = m_new;
      // P^T fragments (B operand): slots 8g+e <-> keys 32ks2+4g+e | 32ks2+16+4g+(e-4)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        pf[qg][ks2].x = pack2bf(sacc[qg][2 * ks2][0], sacc[qg][2 * ks2][1]);
        pf[qg][ks2].y = pack2bf(sacc[qg][2 * ks2][2], sacc[qg][2 * ks2][3]);
        pf[qg][ks2].z = pack2bf(sacc[qg][2 * ks2 + 1][0], sacc[qg][2 * ks2 + 1][1]);
        pf[qg][ks2].w = pack2bf(sacc[qg][2 * ks2 + 1][2], sacc[qg][2 * ks2 + 1][3]);
      }
    }
    IVL_T(tr4);
    // ---- O^T += V^T P^T : 8 d sub-tiles x 2 key-steps ------------------------------------------
    const unsigned char* vbase = smem + SWA_LDS_K;
#pragma unroll
    for (int mt2 = 0; mt2 < 8; ++mt2) {
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        // 16-lane group g reads the 4x16 block rows (32ks2 [+16] + 4g .. +3), cols 16mt2..+15;
        // lane i supplies the address of row (i>>2), cols 4(i&3)..+3 and receives column i.
        const int r0 = 32 * ks2 + 4 * g + (l15 >> 2);
        const int cb = (16 * mt2 + 4 * (l15 & 3)) * 2;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vbase + r0 * SWA_VSTRIDE + cb));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vbase + (r0 + 16) * SWA_VSTRIDE + cb));
        u32x2 w0, w1;
        __builtin_memcpy(&w0, &a0, 8);
        __builtin_memcpy(&w1, &a1, 8);
        const u32x4 vf = u32x4{w0.x, w0.y, w1.x, w1.y};
#pragma unroll
        for (int qg = 0; qg < QG; ++qg)
          oacc[qg][mt2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_mfma(vf), as_mfma(pf[qg][ks2]), oacc[qg][mt2], 0, 0, 0);
      }
    }
    IVL_T(tr5);
    IVL_TACC(tr_b1, tr1, tr0); IVL_TACC(tr_st, tr2, tr1); IVL_TACC(tr_qk, tr3, tr2); IVL_TACC(tr_sm, tr4, tr3); IVL_TACC(tr_pv, tr5, tr4);
  }
  IVL_T(tr_end);

  // ---- epilogue: lane owns its rows, d = 16 mt2 + 4g + r ----------------------------------------
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    if (!row_ok[qg]) continue;
    if (p.nsplit == 1) {
      const float inv = l_run[qg] > 0.f ? 1.0f / l_run[qg] : 0.f;
      bf16_t* op = p.o + (((long long)b * p.T + t_row[qg]) * p.Hq + hq[qg]) * SWA_D + 4 * g;
#pragma unroll
      for (int mt2 = 0; mt2 < 8; ++mt2) {
        u32x2 w;
        w.x = pack2bf(oacc[qg][mt2][0] * inv, oacc[qg][mt2][1] * inv);
        w.y = pack2bf(oacc[qg][mt2][2] * inv, oacc[qg][mt2][3] * inv);
        *(u32x2*)(op + 16 * mt2) = w;
      }
    } else {
      const long long prow = (((long long)b * p.nsplit + split) * p.T + t_row[qg]) * p.Hq + hq[qg];
      float* po = p.part_o + prow * SWA_D + 4 * g;
#pragma unroll
      for (int mt2 = 0; mt2 < 8; ++mt2) *(f32x4*)(po + 16 * mt2) = oacc[qg][mt2];
      if (g == 0) {
        p.part_ml[prow * 2] = m_run[qg];
        p.part_ml[prow * 2 + 1] = l_run[qg];
      }
    }
  }
  IVL_T(tr_fin);
  IVL_TOUT(32, tr_loop - tr_start); IVL_TOUT(33, tr_b1); IVL_TOUT(34, tr_st); IVL_TOUT(35, tr_qk); IVL_TOUT(36, tr_sm);
#ifdef IVL_TRACE
  IVL_TOUT(41, (long long)__builtin_amdgcn_s_memrealtime() - rt_start);
#endif
  IVL_TOUT(37, tr_pv); IVL_TOUT(38, tr_fin - tr_end); IVL_TOUT(39, tr_fin - tr_start); IVL_TOUT(40, kt_end - kt_begin);
}

// ------------------------------------------------------------------------------------------------------------------
// Prefill kernel (T > 64 query rows per head): 8 compute wavefronts = 4 row groups x 2 key halves on v_mfma_f32_32x32x16_bf16,
// plus 4 loader wavefronts.
//   workgroup : 128 query rows of one head x a range of 64-key tiles; wave (rg, kh) owns rows 32 rg .. 32 rg + 31 and
//               keys 32 kh .. 32 kh + 31 of every tile, with its OWN online-softmax state (m, l, O^T); the two key halves
//               of a row group are merged once, through LDS, when the tile loop ends (an in-workgroup split-KV).
//   why       : a 32x32x16 A fragment (1 KB of LDS) feeds 16 K MAC, twice the 16x16x32 one - the 64-row kernel above
//               keeps the LDS array busy 768 clk per tile for 544 clk of MFMA per SIMD; 128 rows at 32 per wave give the
//               same parallelism (workgroups per call) as 16-row waves with half the LDS bytes per MFMA.
//   S^T = K Q^T : A = K rows (ds_read_b128), B = Q^T in registers; lane (row = l & 31, hi = l >> 5) receives the scores of
//               keys (r & 3) + 8 (r >> 2) + 4 hi, r = 0..15.
//   O^T = V^T P^T : the MFMA k-slot 8 hi + j of key step ks carries key 16 ks + 8 (j >> 2) + 4 hi + (j & 3): exactly
//               the order of the lane's own score registers r = 8 ks + j, so P^T is packed in place (no lane exchange);
//               V^T fragments come from ds_read_b64_tr_b16.
//   epilogue  : key-half 1 parks (m, l, O) in LDS, key-half 0 merges and writes the finished rows back, then all 512
//               compute threads store whole rows (256 B of bf16 per row, coalesced), into o or the split's partial rows.
// Round 5 rebuilt the tile loop (the round-2 form -- two barrier segments per tile, key half 1 one segment behind,
// padded images, 37 DMA pieces per tile -- measured 2,400-2,550 cycles per tile; this one 2,100, same-box A/B in DESIGN.md 4.3):
//   * ONE barrier per tile (16 MFMAs per compute wave between barriers instead of 8): a 4-stage ring; the loaders request tile
//     t + 3 during tile t into the stage tile t - 1 left at the previous barrier, and arrive at the barrier that ends tile t with
//     tile t + 2 landed (counted vmcnt) -- the rotated half reads K(t + 1) one barrier early.
//   * the key-half-1 waves (the SIMD partners of the key-half-0 waves) run their phases ROTATED by one: softmax(t), PV(t), then
//     QK^T(t + 1) -- the partner's QK^T MFMAs face this wave's softmax VALU and the partner's softmax faces this wave's PV MFMAs
//     -- at static priority 1 (the second-dispatched half loses the VALU arbitration otherwise: MI355X_MICROARCH.md).
//   * unpadded 64 x 256 B images, XOR-swizzled on the SOURCE side of the DMA (the LDS side of a DMA piece is lane-linear): K
//     16-byte piece p of row r sits at p ^ (r & 15) (conflict-free for the 32-row ds_read_b128 pattern), V 64-byte granule g of
//     row r at g ^ (r & 3) (conflict-free for ds_read_b64_tr_b16); a fragment address is base ^ (kd << 5) | base ^ (mt << 6);
//     32 one-KB DMA pieces per tile, 8 per loader wave, none wasted on padding.
//   * wrap / seam / tail tiles and un-rotated keys of a short fused-rope call go through registers in the loader waves (the
//     same lane -> (row, piece) map as a DMA piece, so the swizzle is shared): at most three such tiles per 68-tile row.
//   * everything derived from the device-resident position is moved to SGPRs (v_readfirstlane): tile kinds and DMA bases are SALU.
// Measured on the way (tools/proto/attn8_proto.hip): the same loop WITHOUT loader waves (512 threads, 256 VGPRs, the
// compute waves issuing 4 DMA pieces each between row maximum and exponentials) ran 894-914 TFLOP/s on an L1-resident tile but
// LOST in the product (3,170 cycles per tile): an LDS-DMA piece blocks its issuing wave for 140-240 cycles when all eight waves
// of a CU stream real K / V (the CU accepts ~1 KB per 30 cycles) -- the loader waves exist to absorb exactly that.
constexpr int PF_QT = 128;
constexpr int PF_OSTRIDE = 528;                                  // bytes per merged fp32 row (512 + 16: bank shift per row)
constexpr int PF_ML_OFF = 4 * 32 * PF_OSTRIDE;                   // (m, l) pairs behind the four 32-row images
constexpr int P8_THREADS = 768;                                  // 8 compute wavefronts + 4 loader wavefronts (3 per SIMD: <= 168 VGPRs)
constexpr int P8_IMG = SWA_KT * 256;                              // 16 KB per image
constexpr int P8_STAGE = 2 * P8_IMG;                              // K | V
constexpr int P8_STAGES = 4;                                      // tile t in stage t % 4 (why four: see the loader loop)
constexpr int P8_LDS_BYTES = P8_STAGES * P8_STAGE;                // 128 KB
static_assert(PF_ML_OFF + 128 * 8 <= P8_LDS_BYTES, "merge image must fit the K/V stages");

// `pos_dev` (= p.pos_dev) is a parameter of its own IN FRONT of the struct: leading scalar / pointer parameters are preloaded into
// SGPRs at wave launch (Makefile: -amdgpu-kernarg-preload-count; a by-value struct is not), and the position load is the head of
// the kernel's start-up chain (kernarg -> position -> tile addresses -> first DMA): it no longer waits for the kernarg s_load.
__global__ __launch_bounds__(P8_THREADS, 1) void swa_prefill_kernel(const long long* pos_dev, SwaParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[P8_LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi5 = lane >> 5, l15 = lane & 15;
  const bool loader = wave >= 8;
  const int rg = wave & 3, kh = (wave >> 2) & 1;
  int bx, rest;
  {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qn = nwg >> 3, rn = nwg & 7;
    if (rn == 0 && qn % p.n_qtiles == 0) {
      const int hx = qn / p.n_qtiles;
      const int r = slot / hx, half = (p.n_qtiles + 1) >> 1;
      bx = qn > 32 ? p.n_qtiles - 1 - r : (r < half ? p.n_qtiles - 1 - r : r - half);
      rest = xcd * hx + slot % hx;
    } else {
      const int lid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
      bx = lid % p.n_qtiles;
      rest = lid / p.n_qtiles;
    }
  }
  const int G = p.Hq / p.Hkv;
  const int hq = rest % p.Hq, bz = rest / p.Hq;
  const int b = bz / p.nsplit, split = bz % p.nsplit;
  const int hk = hq / G;
  IVL_T(tr_start);

  // the position comes through a vector load (the pointer is not provably read-only): everything derived from it -- ring geometry,
  // tile kinds, DMA addresses, the dead-tile tests -- must live in SGPRs, or every test on it becomes a lane-mask operation and
  // every address a 64-bit VALU chain (measured: 480-1,060 cycles per tile for the four DMA pieces of a wave)
  const long long pos_v = pos_dev ? *pos_dev : p.pos;
  const long long pos = (long long)(((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)((unsigned long long)pos_v >> 32)) << 32) |
                                    (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)pos_v));
  const int n_ring = p.C > 0 ? (int)(pos < (long long)p.C ? pos : (long long)p.C) : 0;
  const int n_prev = n_ring + (p.T_new - p.T);
  const int S = n_prev + p.T;
  const int s0 = __builtin_amdgcn_readfirstlane(p.C > 0 ? mod_pos(pos - n_ring, p.C) : 0);

  const int tile_row0 = bx * PF_QT;
  const int last_row = min(tile_row0 + PF_QT, p.T) - 1;
  const int lo_min = p.W > 0 ? max(0, n_prev + tile_row0 - p.W + 1) : 0;
  const int kt0 = lo_min / SWA_KT, kt1 = (n_prev + last_row) / SWA_KT + 1;
  const int per = __builtin_amdgcn_readfirstlane((kt1 - kt0 + p.nsplit - 1) / p.nsplit);
  const int kt_begin = kt0 + split * per;
  const int kt_end = min(kt1, kt_begin + per);
  const int n = max(kt_end - kt_begin, 0);            // workgroup-uniform

  // ---- staging (loader waves 8-11): loader L takes the 4-row chunks c = L + 4 j (j < 4) of both images: 8 DMA pieces per tile ----
  const int r_in = lane >> 4, pp = lane & 15;
  const unsigned int k_piece = (unsigned int)((pp ^ (4 * (wave & 3) + r_in)) << 4);                 // source byte offset inside the row
  const unsigned int v_piece = (unsigned int)(((((pp >> 2) ^ r_in) << 2) | (pp & 3)) << 4);
  const int chunk0 = wave & 3;                                                                      // + 4 j
  const unsigned int lds_base = (unsigned int)(size_t)smem;
  const long long ring_off = p.C > 0 ? (((long long)b * p.Hkv + hk) * p.C) * SWA_D : 0;
  const bf16_t* const k_ring = p.C > 0 ? p.k_cache + ring_off : p.k_new;
  const bf16_t* const v_ring = p.C > 0 ? p.v_cache + ring_off : p.v_new;
  const bf16_t* const k_newb = p.k_new + (long long)b * p.kn_sb + (long long)hk * p.kn_sh;
  const bf16_t* const v_newb = p.v_new + (long long)b * p.vn_sb + (long long)hk * p.vn_sh;
  const unsigned int kn_st32 = (unsigned int)p.kn_st, vn_st32 = (unsigned int)p.vn_st;
  const bool rope_keys = p.rcos != nullptr;             // the call's keys arrive un-rotated (short single-split calls only)
  auto dma_piece = [&](const unsigned char* base_u, unsigned int voff, unsigned int dst) __attribute__((always_inline)) {
    const unsigned long long bu = (unsigned long long)base_u;
    const unsigned char* const base = (const unsigned char*)(((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)(bu >> 32)) << 32) |
                                                             (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)bu));
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
  };
  // tile kinds, monotone in kt: [0, kt_ring_end) lie wholly in the ring (all but the one that holds the wrap point are 64 consecutive
  // slots), [kt_new_begin, kt_new_end) wholly in the call's keys; the rest (wrap, ring / new seam, tail) take the register path
  const int kt_ring_end = n_ring / SWA_KT, kt_new_begin = (n_ring + SWA_KT - 1) / SWA_KT, kt_new_end = S / SWA_KT;
  const int kt_wrap = (p.C > 0 && s0 + n_ring > p.C) ? (p.C - s0) / SWA_KT : -1;       // the tile with slots on both sides of the wrap
  const bool wrap_aligned = p.C > 0 && (p.C - s0) % SWA_KT == 0;                         // ... unless the wrap falls on a tile boundary
  // tile kt of the call into stage `st`; returns the number of DMA instructions this wave issued (0, 4 or 8)
  auto stage_tile = [&](int kt, int st) __attribute__((always_inline)) -> int {
    const int j0 = kt * SWA_KT, slot0 = s0 + j0;
    const bool ring_fast = kt < kt_ring_end && (kt != kt_wrap || wrap_aligned);
    const bool new_fast = kt >= kt_new_begin && kt < kt_new_end;
    int issued = 0;
#pragma unroll
    for (int isv = 0; isv < 2; ++isv) {
      const unsigned int img = lds_base + (unsigned int)st * P8_STAGE + (isv ? (unsigned int)P8_IMG : 0u);
      const unsigned int piece = isv ? v_piece : k_piece;
      const unsigned int st32 = isv ? vn_st32 : kn_st32;
      const bf16_t* const rb = isv ? v_ring : k_ring;
      const bf16_t* const nb = isv ? v_newb : k_newb;
      if (ring_fast) {
        const unsigned char* src = (const unsigned char*)(rb + (long long)(slot0 >= p.C ? slot0 - p.C : slot0) * SWA_D);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_piece(src + (chunk0 + 4 * j) * 1024, (unsigned int)(r_in * 256) + piece, img + 1024u * (chunk0 + 4 * j));
        issued += 4;
      } else if (new_fast && !(isv == 0 && rope_keys)) {
        const unsigned char* src = (const unsigned char*)(nb + (long long)(j0 - n_ring) * st32);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          dma_piece(src + (long long)(chunk0 + 4 * j) * 4 * st32 * 2, (unsigned int)r_in * st32 * 2 + piece, img + 1024u * (chunk0 + 4 * j));
        issued += 4;
      } else {
        // register path (ring wrap, ring / new seam, tail, un-rotated keys): clamped loads from both sources, rows past the end
        // are zero, the call's keys are rotated; written with the lane -> (row, piece) map of a DMA piece
        u32x4 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int jk = j0 + 4 * (chunk0 + 4 * j) + r_in;
          const int jc = min(jk, S - 1);
          int slot = s0 + min(jc, max(n_ring - 1, 0));
          slot = slot >= p.C ? slot - p.C : slot;
          const u32x4 a0 = *(const u32x4*)((const unsigned char*)(rb + (unsigned int)slot * SWA_D) + piece);
          const u32x4 c0 = *(const u32x4*)((const unsigned char*)(nb + (unsigned int)max(jc - n_ring, 0) * st32) + piece);
          u32x4 val = jc < n_ring ? a0 : c0;
          if (isv == 0 && rope_keys && jc >= n_ring) {
            const int jn = jc - n_ring;
            const u32x4 part = *(const u32x4*)((const unsigned char*)(nb + (unsigned int)jn * st32) + (piece ^ 128u));
            const bool is_lo = piece < 128u;
            u32x4 lo = is_lo ? val : part, hi = is_lo ? part : val;
            rope_pair(lo, hi, p.rcos, p.rsin, (long long)p.B * p.T * SWA_D, ((long long)b * p.T + jn) * SWA_D, (int)((piece & 127u) >> 1), p.rs0, p.rs1);
            val = is_lo ? lo : hi;
          }
          r[j] = jk < S ? val : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) *(u32x4*)(smem + st * P8_STAGE + (isv ? P8_IMG : 0) + 1024 * (chunk0 + 4 * j) + 16 * lane) = r[j];
      }
    }
    return issued;
  };

  if (loader) {
    // ---- loader wavefronts: per tile  [request tile t + 2 over tile t - 1] | vmcnt: tile t + 1 has landed | barrier ---------------
    auto land = [&](int fly) __attribute__((always_inline)) {            // `fly` younger DMA instructions may stay in flight
      __builtin_amdgcn_sched_barrier(0);
      if (fly == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else if (fly == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    // The rotated key half reads K(t + 1) during tile t, i.e. BEHIND the barrier that ends tile t - 1: the loaders arrive at the
    // barrier that ends tile t with tile t + 2 landed (not t + 1), which takes a request distance of three tiles and FOUR stages
    // (tile t + 3 goes into the stage tile t - 1 left at the previous barrier).  [A three-stage version that awaited tile t + 1
    // only passed every single-process test and failed 4 of 12 two-process runs: the DMA of tile t + 1 was usually, not always,
    // complete when the rotated half read it.]
    if (n > 0) stage_tile(kt_begin, 0);
    if (n > 1) stage_tile(kt_begin + 1, 1);
    const int fly2 = n > 2 ? stage_tile(kt_begin + 2, 2) : 0;
    land(fly2);                                                           // tiles 0 and 1 are in LDS: the compute waves start
#pragma nounroll
    for (int t = 0; t < n; ++t) {
      const int fly = t + 3 < n ? stage_tile(kt_begin + t + 3, (t + 3) & 3) : 0;
      land(fly);                                                          // tile t + 2 has landed
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                      // the three barriers of the merge epilogue
    __syncthreads();
    __syncthreads();
    return;
  }

  // ---- compute-side set-up ----------------------------------------------------------------------------------------------------
  const int row = tile_row0 + 32 * rg + l31;
  const bool row_ok = row < p.T;
  const int band_hi = n_prev + row;
  const int band_lo = p.W > 0 ? max(0, n_prev + row - p.W + 1) : 0;
  const int wr0 = tile_row0 + 32 * rg, wr1 = max(min(wr0 + 31, p.T - 1), wr0);
  const int w_hi_min = n_prev + wr0, w_hi_max = n_prev + wr1;
  const int w_lo_min = p.W > 0 ? max(0, n_prev + wr0 - p.W + 1) : 0;
  const int w_lo_max = p.W > 0 ? max(0, n_prev + wr1 - p.W + 1) : 0;
  u32x4 qf[8];
  {
    const bf16_t* qp = p.q + (long long)b * p.q_sb + (long long)min(row, p.T - 1) * p.q_st + (long long)hq * p.q_sh;
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) qf[kd] = *(const u32x4*)(qp + 16 * kd + 8 * hi5);
  }
  if (p.rcos != nullptr) {
    const long long row_off = ((long long)b * p.T + min(row, p.T - 1)) * SWA_D;
    const long long plane = (long long)p.B * p.T * SWA_D;
    u32x4 c1[4], n1[4], c2[4], n2[4];
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
      const int c0 = 16 * kd + 8 * hi5;
      const int sec = c0 < p.rs0 ? 0 : (c0 < p.rs0 + p.rs1 ? 1 : 2);
      const long long off = sec * plane + row_off + c0;
      c1[kd] = *(const u32x4*)(p.rcos + off); n1[kd] = *(const u32x4*)(p.rsin + off);
      c2[kd] = *(const u32x4*)(p.rcos + off + 64); n2[kd] = *(const u32x4*)(p.rsin + off + 64);
    }
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) rope_apply(qf[kd], qf[kd + 4], c1[kd], n1[kd], c2[kd], n2[kd]);
  }
  if (!row_ok) {
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) qf[kd] = u32x4{0u, 0u, 0u, 0u};
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  const float sc = p.scaling * LOG2E;
  const unsigned int k_base = (unsigned int)((32 * kh + l31) * 256 + ((l15 >> 1) << 5) + ((hi5 ^ (l15 & 1)) << 4));
  const unsigned int v_base = (unsigned int)(P8_IMG + (32 * kh + 4 * hi5 + (l15 >> 2)) * 256 + ((l15 >> 2) << 6) + 32 * ((lane >> 4) & 1) + 8 * (l15 & 3));
  unsigned int ak[8], av[4];
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) ak[kd] = k_base ^ (unsigned int)(kd << 5);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) av[mt] = v_base ^ (unsigned int)(mt << 6);

  auto is_dead = [&](int kt) __attribute__((always_inline)) {                // no row of the wave sees any of its 32 keys of tile kt
    const int kbeg = kt * SWA_KT + 32 * kh;
    return kbeg > w_hi_max || kbeg + 31 < w_lo_min;
  };
  auto qk = [&](int st, f32x16& s) __attribute__((always_inline)) {
    const unsigned int so = (unsigned int)st * P8_STAGE;
    u32x4 fr[8];
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) fr[kd] = *(const u32x4*)(smem + (ak[kd] + so));
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_mfma(fr[kd]), as_mfma(qf[kd]), s, 0, 0, 0);
  };
  // band mask, running exponent reference, rescale.  LAZY reference: m_run is the exponent reference of the row, not its exact
  // running maximum.  It follows the maximum only when some row of the wave has outgrown it by more than 2^8 (always on a row's
  // first visible tile: -inf + 8 = -inf); until then the probabilities are 2^(s - m_run) <= 2^8 -- exact in fp32 and of the same
  // relative precision in bf16 -- and the accumulators need no rescale: 33 v_pk_mul_f32 per wave and tile behind a wave-uniform
  // branch that is taken a handful of times per call.  (m, l, O) stay consistent: the merge of the key halves and the split-KV
  // combine use m_run as the partial's reference.  The first reader of the MFMA results is an instruction the compiler sees
  // (hazard wait states: see swa_fwd_kernel).
  auto row_max = [&](int kt, f32x16& s) __attribute__((always_inline)) {
    const int kbeg = kt * SWA_KT + 32 * kh;
    const bool interior = kbeg >= w_lo_max && kbeg + 31 <= w_hi_min;
    if (!interior) {
      const int jb = kbeg + 4 * hi5;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = jb + (r & 3) + 8 * (r >> 2);
        const bool vis = row_ok && j >= band_lo && j <= band_hi;
        s[r] = vis ? s[r] : -INFINITY;
      }
    }
    float rmax = vmax2(__builtin_fmaxf(s[0], s[1]), s[2]);       // first reader of the MFMA results: compiler-visible (hazard wait states)
    rmax = vmax3(rmax, s[3], s[4]);
    rmax = vmax3(rmax, s[5], s[6]);
    rmax = vmax3(rmax, s[7], s[8]);
    rmax = vmax3(rmax, s[9], s[10]);
    rmax = vmax3(rmax, s[11], s[12]);
    rmax = vmax3(rmax, s[13], s[14]);
    rmax = vmax2(rmax, s[15]);
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
    rmax = vmax2(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * sc;
    const float m_new = vmax2(m_run, rmax);
    if (__any(m_new > m_run + 8.0f)) {
      const float m_u = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_u);
#pragma unroll
      for (int i = 0; i < 4; ++i) oacc[i] *= alpha;
      l_run *= alpha;
      m_run = m_new;
    }
  };
  auto exps = [&](f32x16& s, u32x4 (&pf)[2]) __attribute__((always_inline)) {
    const float m_use = m_run == -INFINITY ? 0.f : m_run;
    float rsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sc, -m_use));
      s[r] = pv;
      rsum += pv;
    }
    l_run += rsum;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      pf[ks] = u32x4{pack2bf(s[8 * ks + 0], s[8 * ks + 1]), pack2bf(s[8 * ks + 2], s[8 * ks + 3]),
                     pack2bf(s[8 * ks + 4], s[8 * ks + 5]), pack2bf(s[8 * ks + 6], s[8 * ks + 7])};
  };
  auto pv = [&](int st, const u32x4 (&pf)[2]) __attribute__((always_inline)) {
    const unsigned int so = (unsigned int)st * P8_STAGE;
    u32x4 fv[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const unsigned char* vp = smem + (av[mt] + so) + 16 * ks * 256;
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)vp);
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp + 8 * 256));
        u32x2 w0, w1;
        __builtin_memcpy(&w0, &a0, 8);
        __builtin_memcpy(&w1, &a1, 8);
        fv[ks][mt] = u32x4{w0.x, w0.y, w1.x, w1.y};
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_mfma(fv[ks][mt]), as_mfma(pf[ks]), oacc[mt], 0, 0, 0);
  };
  // the barrier that ends a tile: nobody reads the current tile any more; the loaders arrive with the next tile in LDS
  auto tile_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  tile_barrier();                              // tiles 0 and 1 are in LDS

  IVL_T(tr_loop);
  IVL_TVAR(tr_c1); IVL_TVAR(tr_st); IVL_TVAR(tr_c2); IVL_TVAR(tr_w);
  f32x16 sA;
  u32x4 pA[2];
  int st = 0;                                  // stage of tile t = t % 3
  if (kh == 0) {
    // key half 0:  QK^T(t) | row max | exponentials | PV(t) | barrier
#pragma nounroll
    for (int t = 0; t < n; ++t) {
      const int kt = kt_begin + t;
      const bool dead = is_dead(kt);
      IVL_T(t0);
      if (!dead) {
        qk(st, sA);
        row_max(kt, sA);
      }
      IVL_T(t1);
      IVL_T(t2);
      if (!dead) {
        exps(sA, pA);
        pv(st, pA);
      }
      IVL_T(t3);
      tile_barrier();
      IVL_T(t4);
      IVL_TACC(tr_c1, t1, t0); IVL_TACC(tr_st, t2, t1); IVL_TACC(tr_c2, t3, t2); IVL_TACC(tr_w, t4, t3);
      st = (st + 1) & 3;
    }
  } else {
    // key half 1, rotated:  row max(t) | exponentials | PV(t) | QK^T(t + 1) | barrier
    // (tile t + 1 is complete since the barrier that ended tile t - 1: the loaders run three tiles ahead)
    __builtin_amdgcn_s_setprio(1);               // the second-dispatched half loses the VALU arbitration against its SIMD partner
    if (n > 0 && !is_dead(kt_begin)) qk(0, sA);
#pragma nounroll
    for (int t = 0; t < n; ++t) {
      const int kt = kt_begin + t;
      const bool dead = is_dead(kt);
      IVL_T(t0);
      if (!dead) row_max(kt, sA);
      IVL_T(t1);
      const int st1 = (st + 1) & 3;
      IVL_T(t2);
      if (!dead) {
        exps(sA, pA);
        pv(st, pA);
      }
      if (t + 1 < n && !is_dead(kt + 1)) qk(st1, sA);       // (the scores of tile t are packed in pA by now)
      IVL_T(t3);
      tile_barrier();
      IVL_T(t4);
      IVL_TACC(tr_c1, t1, t0); IVL_TACC(tr_st, t2, t1); IVL_TACC(tr_c2, t3, t2); IVL_TACC(tr_w, t4, t3);
      st = st1;
    }
  }
  IVL_T(tr_end);

  // ---- merge the two key halves of every row group through LDS (the K/V stages are free behind the next barrier), then whole-row
  //      stores by the 512 compute threads: 256 B of bf16 per row, into o or into the split's partial rows ----------------------------
  {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  __syncthreads();
  unsigned char* oimg = smem + rg * 32 * PF_OSTRIDE + l31 * PF_OSTRIDE + 16 * hi5;      // + 128 mt + 32 q : d = 32 mt + 8 q + 4 hi
  float* ml = (float*)(smem + PF_ML_OFF) + (rg * 32 + l31) * 2;
  if (kh == 1) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4*)(oimg + 128 * mt + 32 * q) = f32x4{oacc[mt][4 * q], oacc[mt][4 * q + 1], oacc[mt][4 * q + 2], oacc[mt][4 * q + 3]};
    if (hi5 == 0) {
      ml[0] = m_run;
      ml[1] = l_run;
    }
  }
  __syncthreads();
  if (kh == 0) {
    const float m1 = ml[0], l1 = ml[1];
    const float m = vmax2(m_run, m1);
    const float mu = m == -INFINITY ? 0.f : m;
    const float a0 = __builtin_amdgcn_exp2f(m_run - mu), a1 = __builtin_amdgcn_exp2f(m1 - mu);
    const float l = l_run * a0 + l1 * a1;
    const float inv = l > 0.f ? 1.0f / l : 0.f;      // normalised rows also for the split-KV partials (stored in bf16)
    const float w0 = a0 * inv, w1 = a1 * inv;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 o1 = *(const f32x4*)(oimg + 128 * mt + 32 * q);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = oacc[mt][4 * q + e] * w0 + o1[e] * w1;
        *(f32x4*)(oimg + 128 * mt + 32 * q) = o;
      }
    if (hi5 == 0) {
      ml[0] = m;
      ml[1] = l;
    }
  }
  __syncthreads();
  {
    bf16_t* const dst = p.nsplit == 1 ? p.o : (bf16_t*)p.part_o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 512 * i, r = idx >> 4, c = idx & 15;       // row, 8-channel piece
      const int t = tile_row0 + r;
      if (t < p.T) {
        const unsigned char* src = smem + r * PF_OSTRIDE + c * 32;
        const f32x4 x = *(const f32x4*)src, y = *(const f32x4*)(src + 16);
        const long long orow = p.nsplit == 1 ? ((long long)b * p.T + t) * p.Hq + hq
                                             : (((long long)b * p.nsplit + split) * p.T + t) * p.Hq + hq;
        store_out16(dst + orow * SWA_D + 8 * c,
                    u32x4{pack2bf(x[0], x[1]), pack2bf(x[2], x[3]), pack2bf(y[0], y[1]), pack2bf(y[2], y[3])});
      }
    }
    if (p.nsplit > 1 && tid < PF_QT && tile_row0 + tid < p.T) {
      const long long prow = (((long long)b * p.nsplit + split) * p.T + tile_row0 + tid) * p.Hq + hq;
      *(float2*)(p.part_ml + prow * 2) = *(const float2*)(smem + PF_ML_OFF + tid * 8);
    }
  }
  IVL_T(tr_fin);
  IVL_TOUT(32, tr_loop - tr_start); IVL_TOUT(38, tr_fin - tr_end); IVL_TOUT(39, tr_fin - tr_start); IVL_TOUT(40, kt_end - kt_begin);
  IVL_TOUT(47, tr_end - tr_loop); IVL_TOUT(42, 2);
  IVL_TOUT(33, tr_c1); IVL_TOUT(34, tr_st); IVL_TOUT(35, tr_c2); IVL_TOUT(36, tr_w);
  IVL_TOUT_AT(256, 50, tr_c1); IVL_TOUT_AT(256, 51, tr_st); IVL_TOUT_AT(256, 52, tr_c2); IVL_TOUT_AT(256, 53, tr_w);
}

// ------------------------------------------------------------------------------------------------------------------
// fp8 (e4m3) variant of the packed DECODE step (BASELINE.json configs[4]; the reference has no such path): q, the K / V
// tile and the probabilities are rounded to OCP e4m3 and both products run on v_mfma_f32_16x16x32_fp8_fp8; scores, the
// online softmax, the row sums and the output accumulators stay fp32.  Same band, same split-KV partials and combine
// kernels as the bf16 path.  The ring cache stays bf16 in HBM (the tile is converted while it is staged): the variant
// halves the LDS traffic per MFMA, not the HBM read.
//   K image : [64 keys][128 B], row stride 136 B           A fragment of S^T = K Q^T: 8 bytes at d = 32 ks + 8 g
//   V^T image: [128 d][64 B], row stride 72 B, key 32 ks2 + r stored at byte 32 ks2 + (r < 16 ? 8 (r >> 2) + (r & 3)
//             : 8 ((r - 16) >> 2) + 4 + (r & 3))  = the slot order of the probabilities in the accumulators
constexpr int F8_KSTRIDE = 136, F8_VSTRIDE = 72;
constexpr int F8_LDS_K = SWA_KT * F8_KSTRIDE;
constexpr int F8_LDS_BYTES = F8_LDS_K + SWA_D * F8_VSTRIDE;

__device__ __forceinline__ unsigned int pack4_fp8(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f);
  b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f);
  d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (unsigned int)r;
}
__device__ __forceinline__ u32x2 bf16x8_to_fp8(u32x4 v) {
  return u32x2{pack4_fp8(bflo(v.x), bfhi(v.x), bflo(v.y), bfhi(v.y)), pack4_fp8(bflo(v.z), bfhi(v.z), bflo(v.w), bfhi(v.w))};
}
__device__ __forceinline__ f32x4 mma_fp8(u32x2 a, u32x2 b, f32x4 c) {
  long la, lb;
  __builtin_memcpy(&la, &a, 8);
  __builtin_memcpy(&lb, &b, 8);
  return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void swa_decode_fp8_kernel(const long long* pos_dev, SwaParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[F8_LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int G = p.Hq / p.Hkv;
  // grid = B * nsplit * Hkv (one 64-row packed query tile per kv head); ordered (b, split, kv head)
  const int hk = blockIdx.x % p.Hkv;
  const int bz = blockIdx.x / p.Hkv;
  const int b = bz / p.nsplit, split = bz % p.nsplit;

  const long long pos = pos_dev ? *pos_dev : p.pos;
  const int n_ring = p.C > 0 ? (int)(pos < (long long)p.C ? pos : (long long)p.C) : 0;
  const int n_extra = p.T_new - p.T;
  const int n_prev = n_ring + n_extra;
  const int S = n_prev + p.T;
  const int s0 = p.C > 0 ? mod_pos(pos - n_ring, p.C) : 0;

  const int total_rows = p.T * G;
  const int row = wave * 16 + l15;
  const bool row_ok = row < total_rows;
  const int t_row = row / G, hq = hk * G + row % G;
  const int hi = n_prev + t_row;
  const int lo = p.W > 0 ? max(0, n_prev + t_row - p.W + 1) : 0;

  const int t_max = (total_rows - 1) / G;
  const int lo_min = p.W > 0 ? max(0, n_prev - p.W + 1) : 0;
  const int kt0 = lo_min / SWA_KT, kt1 = (n_prev + t_max) / SWA_KT + 1;
  const int per = (kt1 - kt0 + p.nsplit - 1) / p.nsplit;
  const int kt_begin = kt0 + split * per;
  const int kt_end = min(kt1, kt_begin + per);

  // Q fragments in e4m3: lane = query row, k-slots 8g..8g+7 of each 32-chunk
  u32x2 qf[4];
  {
    const bf16_t* qp = p.q + (long long)b * p.q_sb + (long long)t_row * p.q_st + (long long)hq * p.q_sh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = row_ok ? bf16x8_to_fp8(*(const u32x4*)(qp + 32 * ks + 8 * g)) : u32x2{0u, 0u};
  }
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 oacc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int srow = tid >> 4, schunk = tid & 15;
  const bf16_t* kb_ring = p.C > 0 ? p.k_cache + (((long long)b * p.Hkv + hk) * p.C) * SWA_D + schunk * 8 : p.k_new;
  const bf16_t* vb_ring = p.C > 0 ? p.v_cache + (((long long)b * p.Hkv + hk) * p.C) * SWA_D + schunk * 8 : p.v_new;
  const bf16_t* kb_new = p.k_new + (long long)b * p.kn_sb + (long long)hk * p.kn_sh + schunk * 8;
  const bf16_t* vb_new = p.v_new + (long long)b * p.kn_sb + (long long)hk * p.kn_sh + schunk * 8;
  const unsigned int kn_st32 = (unsigned int)p.kn_st;
  const float sc = p.scaling * LOG2E;

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int j0 = kt * SWA_KT;
    // ---- stage the tile: branch-free clamped loads (ring wrap / seam / tail), converted to e4m3 on the way to LDS ----
    u32x4 kreg[4], vreg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = j0 + srow + 16 * i;
      const int jc = min(j, S - 1);
      const bool in_ring = jc < n_ring;
      int slot = s0 + jc;
      slot = slot >= p.C ? slot - p.C : slot;
      const unsigned int off = in_ring ? (unsigned int)slot * SWA_D : (unsigned int)(jc - n_ring) * kn_st32;
      kreg[i] = *(const u32x4*)((in_ring ? kb_ring : kb_new) + off);
      vreg[i] = *(const u32x4*)((in_ring ? vb_ring : vb_new) + off);
      if (j >= S) { kreg[i] = u32x4{0u, 0u, 0u, 0u}; vreg[i] = u32x4{0u, 0u, 0u, 0u}; }
    }
    __syncthreads();                     // the previous tile's readers are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = srow + 16 * i;       // key within the tile
      *(u32x2*)(smem + r * F8_KSTRIDE + schunk * 8) = bf16x8_to_fp8(kreg[i]);
      const u32x2 v8 = bf16x8_to_fp8(vreg[i]);
      const int r32 = r & 31;
      const int kpos = (r & 32) + (r32 < 16 ? 8 * (r32 >> 2) + (r32 & 3) : 8 * ((r32 - 16) >> 2) + 4 + (r32 & 3));
      unsigned char* vt = smem + F8_LDS_K + (schunk * 8) * F8_VSTRIDE + kpos;
#pragma unroll
      for (int c = 0; c < 8; ++c) vt[c * F8_VSTRIDE] = (unsigned char)((c < 4 ? v8.x >> (8 * c) : v8.y >> (8 * (c - 4))) & 0xffu);
    }
    __syncthreads();
    // ---- S^T = K Q^T ----------------------------------------------------------------------------------------------
    f32x4 sacc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) sacc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        sacc[mt] = mma_fp8(*(const u32x2*)(smem + (16 * mt + l15) * F8_KSTRIDE + 32 * ks + 8 * g), qf[ks], sacc[mt]);
    // ---- band + online softmax (lane-local row) -------------------------------------------------------------------
    const int jbase = j0 + 4 * g;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = jbase + 16 * mt + r;
        sacc[mt][r] = (row_ok && j >= lo && j <= hi) ? sacc[mt][r] : -INFINITY;
      }
    float rmax = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) rmax = vmax2(rmax, sacc[mt][r]);
    rmax = group_max(rmax) * sc;
    const float m_new = vmax2(m_run, rmax);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    float rsum = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[mt][r], sc, -m_use));
        sacc[mt][r] = pv;
        rsum += pv;
      }
    rsum = group_sum(rsum);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    l_run = l_run * alpha + rsum;
#pragma unroll
    for (int i = 0; i < 8; ++i) oacc[i] *= alpha;
    m_run = m_new;
    u32x2 pf[2];
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
      pf[ks2] = u32x2{pack4_fp8(sacc[2 * ks2][0], sacc[2 * ks2][1], sacc[2 * ks2][2], sacc[2 * ks2][3]),
                      pack4_fp8(sacc[2 * ks2 + 1][0], sacc[2 * ks2 + 1][1], sacc[2 * ks2 + 1][2], sacc[2 * ks2 + 1][3])};
    // ---- O^T += V^T P^T ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int mt2 = 0; mt2 < 8; ++mt2)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2)
        oacc[mt2] = mma_fp8(*(const u32x2*)(smem + F8_LDS_K + (16 * mt2 + l15) * F8_VSTRIDE + 32 * ks2 + 8 * g), pf[ks2], oacc[mt2]);
  }

  if (!row_ok) return;
  if (p.nsplit == 1) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    bf16_t* op = p.o + (((long long)b * p.T + t_row) * p.Hq + hq) * SWA_D + 4 * g;
#pragma unroll
    for (int mt2 = 0; mt2 < 8; ++mt2)
      *(u32x2*)(op + 16 * mt2) = u32x2{pack2bf(oacc[mt2][0] * inv, oacc[mt2][1] * inv), pack2bf(oacc[mt2][2] * inv, oacc[mt2][3] * inv)};
  } else {
    const long long prow = (((long long)b * p.nsplit + split) * p.T + t_row) * p.Hq + hq;
    float* po = p.part_o + prow * SWA_D + 4 * g;
#pragma unroll
    for (int mt2 = 0; mt2 < 8; ++mt2) *(f32x4*)(po + 16 * mt2) = oacc[mt2];
    if (g == 0) {
      p.part_ml[prow * 2] = m_run;
      p.part_ml[prow * 2 + 1] = l_run;
    }
  }
}

// M-RoPE pre-pass of a split-KV prefill call: q and the call's keys are rotated ONCE into the workspace (contiguous
// [B,T,H,128]) and the attention kernel runs on rotated data.  Rotating inside the attention kernel repeats the work in every
// split and every q head of a kv group: at the step shape (8 splits) the q rope alone cost every workgroup 128 KB of
// cos / sin traffic and ~5 us per launch; the pre-pass costs one ~2 us launch.  Same arithmetic (rope_pair): bit-identical.
__global__ __launch_bounds__(256) void swa_rope_prepass_kernel(const bf16_t* __restrict__ q, long long q_sb, long long q_st, long long q_sh,
                                                              const bf16_t* __restrict__ k, long long k_sb, long long k_st, long long k_sh,
                                                              bf16_t* __restrict__ q_out, bf16_t* __restrict__ k_out, int B, int T, int Hq,
                                                              int Hkv, const bf16_t* __restrict__ rcos, const bf16_t* __restrict__ rsin,
                                                              int rs0, int rs1) {
  const int HT = Hq + Hkv;
  const long long total = (long long)B * T * HT * 8;               // (row, head, channel-block pair c, c + 8)
  const long long plane = (long long)B * T * SWA_D;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx & 7);
    int h, t;
    const long long bt = divmod_idx(idx >> 3, HT, h);
    const int b = (int)divmod_idx(bt, T, t);
    const bool is_q = h < Hq;
    const bf16_t* src = is_q ? q + (long long)b * q_sb + (long long)t * q_st + (long long)h * q_sh
                             : k + (long long)b * k_sb + (long long)t * k_st + (long long)(h - Hq) * k_sh;
    u32x4 lo = *(const u32x4*)(src + 8 * c), hi = *(const u32x4*)(src + 8 * c + 64);
    rope_pair(lo, hi, rcos, rsin, plane, bt * SWA_D, 8 * c, rs0, rs1);
    bf16_t* dst = is_q ? q_out + (bt * Hq + h) * SWA_D : k_out + (bt * Hkv + (h - Hq)) * SWA_D;
    store_out16(dst + 8 * c, lo);
    store_out16(dst + 8 * c + 64, hi);
  }
}

// Ring append folded into the combine launch (ivl_swa_args.append_new): the blocks behind the combine blocks copy the
// call's tokens into the ring.  The attention kernel has finished by then (stream order), so no reader of the old
// slots is left; one launch per layer and step instead of two.
struct AppendArgs {
  const bf16_t* k_new; const bf16_t* v_new; long long kn_sb, kn_st, kn_sh, vn_sb, vn_st, vn_sh;
  bf16_t* k_cache; bf16_t* v_cache;
  int B, T, Hkv, C; long long pos; const long long* pos_dev;
  const bf16_t* rcos; const bf16_t* rsin; int rs0, rs1;
  int first_block;          // index of the first append block in the grid; < 0: no append
};
// token t of the call -> slot (pos + t) % C ; only the last min(T, C) tokens are written
__device__ __forceinline__ void ring_append(const AppendArgs& a, long long block, long long nblocks) {
  const long long pos = a.pos_dev ? *a.pos_dev : a.pos;
  const int t_first = a.T > a.C ? a.T - a.C : 0;
  const int nt = a.T - t_first;
  const long long total = (long long)a.B * nt * a.Hkv * (SWA_D / 8);
  const int pos_slot = mod_pos(pos, a.C);
  for (long long idx = block * blockDim.x + threadIdx.x; idx < total; idx += nblocks * blockDim.x) {
    const int ch = (int)(idx & (SWA_D / 8 - 1));
    int hk, tt;
    const long long bt = divmod_idx(idx >> 4, a.Hkv, hk);          // SWA_D / 8 = 16 pieces per row
    const int b = (int)divmod_idx(bt, nt, tt);
    tt += t_first;
    const int slot = (int)(((unsigned int)pos_slot + (unsigned int)tt) % (unsigned int)a.C);      // (pos + tt) % C; pos_slot, tt < 2^31
    const long long src = (long long)b * a.kn_sb + (long long)tt * a.kn_st + (long long)hk * a.kn_sh + ch * 8;
    const long long dst = (((long long)b * a.Hkv + hk) * a.C + slot) * SWA_D + ch * 8;
    u32x4 kv = *(const u32x4*)(a.k_new + src);
    if (a.rcos != nullptr) {                     // rotate on the way into the ring (same arithmetic as the attention kernel)
      const u32x4 part = *(const u32x4*)(a.k_new + src + ((ch ^ 8) - ch) * 8);
      u32x4 lo = ch < 8 ? kv : part, hi = ch < 8 ? part : kv;
      rope_pair(lo, hi, a.rcos, a.rsin, (long long)a.B * a.T * SWA_D, ((long long)b * a.T + tt) * SWA_D, (ch & 7) * 8, a.rs0, a.rs1);
      kv = ch < 8 ? lo : hi;
    }
    *(u32x4*)(a.k_cache + dst) = kv;
    *(u32x4*)(a.v_cache + dst) =
        *(const u32x4*)(a.v_new + ((long long)b * a.vn_sb + (long long)tt * a.vn_st + (long long)hk * a.vn_sh + ch * 8));
  }
}

// merge split-KV partials: one wavefront per (b, t, head) row, 2 d-values per lane; every split's loads are
// issued before the first use (template on the split count so the loop is fully unrolled)
// BF16P (partials of swa_prefill_kernel): part_o holds NORMALISED rows O_s / l_s in bf16 (half the round trip of the
// split-KV partials, which is 17 MB per launch in fp32 at the bench shape); the merge weights are then w_s l_s.
template <int NS, bool BF16P>
__global__ __launch_bounds__(256) void swa_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                         bf16_t* __restrict__ o, int B, int rows_per_b, int nsplit, AppendArgs ap) {
  if (ap.first_block >= 0 && (int)blockIdx.x >= ap.first_block) {
    ring_append(ap, (long long)blockIdx.x - ap.first_block, (long long)gridDim.x - ap.first_block);
    return;
  }
  const int ncb = ap.first_block >= 0 ? ap.first_block : (int)gridDim.x;       // combine blocks
  const int lane = threadIdx.x & 63;
  const long long wid = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nw = ((long long)ncb * blockDim.x) >> 6;
  for (long long r = wid; r < (long long)B * rows_per_b; r += nw) {
    int rr_;
    const long long b = divmod_idx(r, rows_per_b, rr_), rr = rr_;
    float ms[NS], ls[NS];
    float2 ov[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const bool on = s < nsplit;
      const long long pr = (b * nsplit + (on ? s : 0)) * rows_per_b + rr;
      const float2 ml = *(const float2*)(part_ml + pr * 2);
      ms[s] = on ? ml.x : -INFINITY;
      ls[s] = on ? ml.y : 0.f;
      if (BF16P) {
        const unsigned int w2 = *(const unsigned int*)((const bf16_t*)part_o + pr * SWA_D + 2 * lane);
        ov[s] = float2{bflo(w2), bfhi(w2)};
      } else {
        ov[s] = *(const float2*)(part_o + pr * SWA_D + 2 * lane);
      }
    }
    float m = -INFINITY;
#pragma unroll
    for (int s = 0; s < NS; ++s) m = fmaxf(m, ms[s]);
    float l = 0.f, a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float w = ms[s] == -INFINITY ? 0.f : exp2f(ms[s] - m);
      if (BF16P) w *= ls[s];
      l = BF16P ? l + w : fmaf(w, ls[s], l);
      a0 = fmaf(w, ov[s].x, a0);
      a1 = fmaf(w, ov[s].y, a1);
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    *(unsigned int*)(o + r * SWA_D + 2 * lane) = pack2bf(a0 * inv, a1 * inv);
  }
}

// merge up to 64 split-KV partials (packed decode): one wavefront per (b, t, head) row.  Lane s first owns split s
// (its running max / sum -> weight 2^(m_s - m)), then the wave walks the splits with the weights broadcast by
// v_readlane while every lane accumulates its 2 d-values; loads are issued 8 splits at a time.
__global__ __launch_bounds__(256) void swa_combine_wide_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                              bf16_t* __restrict__ o, int B, int rows_per_b, int nsplit, AppendArgs ap) {
  if (ap.first_block >= 0 && (int)blockIdx.x >= ap.first_block) {
    ring_append(ap, (long long)blockIdx.x - ap.first_block, (long long)gridDim.x - ap.first_block);
    return;
  }
  // one WORKGROUP per row: wave w merges the splits 16w .. 16w + 15 (all 16 partial rows requested at once: one memory
  // round trip instead of nsplit / 8), the four partial sums meet in LDS
  __shared__ float2 red[4][64];
  const int ncb = ap.first_block >= 0 ? ap.first_block : (int)gridDim.x;       // combine blocks
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long long r = blockIdx.x; r < (long long)B * rows_per_b; r += ncb) {
    int rr_;
    const long long b = divmod_idx(r, rows_per_b, rr_), rr = rr_;
    const bool on = lane < nsplit;
    const float2 ml = *(const float2*)(part_ml + ((b * nsplit + (on ? lane : 0)) * rows_per_b + rr) * 2);
    float2 ov[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int s2 = min(16 * wave + j, nsplit - 1);
      ov[j] = *(const float2*)(part_o + ((b * nsplit + s2) * rows_per_b + rr) * SWA_D + 2 * lane);
    }
    const float ms = on ? ml.x : -INFINITY;
    float m = ms;
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) m = fmaxf(m, __shfl_xor(m, ofs, 64));
    const float w = ms == -INFINITY ? 0.f : exp2f(ms - m);
    const float l = wave_sum(w * (on ? ml.y : 0.f));
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int s2 = 16 * wave + j;
      const float ws = s2 < nsplit ? __shfl(w, s2, 64) : 0.f;
      a0 = fmaf(ws, ov[j].x, a0);
      a1 = fmaf(ws, ov[j].y, a1);
    }
    red[wave][lane] = float2{a0, a1};
    __syncthreads();
    if (wave == 0) {
      const float2 p1 = red[1][lane], p2 = red[2][lane], p3 = red[3][lane];
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      *(unsigned int*)(o + r * SWA_D + 2 * lane) = pack2bf((a0 + p1.x + p2.x + p3.x) * inv, (a1 + p1.y + p2.y + p3.y) * inv);
    }
    __syncthreads();
  }
}

// stand-alone ring append (ivl_swa_cache_append, and ivl_swa_fwd with append_new when no combine launch follows)
__global__ __launch_bounds__(256) void swa_cache_append_kernel(AppendArgs ap) { ring_append(ap, blockIdx.x, gridDim.x); }

// The 64-row kernel (QG = 1) serves decode / T <= 64 calls; longer calls run swa_prefill_kernel (the QG = 2 instantiation of
// round 2a - two 16-row groups per wave - is superseded by it and no longer launched).
static int swa_qg(int, int, int) { return 1; }

static int swa_base_nsplit(int B, int T, int Hq) {
  if (T > SWA_QT) {
    const long long base = (long long)B * ((T + PF_QT - 1) / PF_QT) * Hq;
    long long ns = (256 + base - 1) / base;
    if (ns < 1) ns = 1;
    if (ns > SWA_MAX_SPLIT) ns = SWA_MAX_SPLIT;
    return (int)ns;
  }
  const int qt = SWA_QT * swa_qg(B, T, Hq);
  const long long base = (long long)B * ((T + qt - 1) / qt) * Hq;
  long long ns = (512 + base - 1) / base;
  if (ns < 1) ns = 1;
  if (ns > SWA_MAX_SPLIT) ns = SWA_MAX_SPLIT;
  return (int)ns;
}

int swa_ring256_launch(const ivl_swa_args* a, hipStream_t st);      // swa_ring256.hip

}  // namespace ivl

using namespace ivl;

extern "C" size_t ivl_swa_workspace_bytes(int B, int T, int Hq, int d) {
  if (B <= 0 || T <= 0 || Hq <= 0 || d != SWA_D) return 0;
  int ns = swa_base_nsplit(B, T, Hq);
  if (T <= SWA_QT) ns = SWA_MAX_SPLIT_PACK;      // packed decode rows may use the maximum split
  // split-KV partials (+ the rotated q / k copies of the rope pre-pass: at most 2 Hq heads of bf16 rows)
  const size_t rot = T > SWA_QT ? (size_t)B * T * Hq * SWA_D * 2 * sizeof(bf16_t) : 0;
  if (ns == 1) return rot + 256;
  return (size_t)B * ns * T * Hq * (SWA_D + 2) * sizeof(float) + rot + 256;
}

extern "C" int ivl_swa_fwd(const ivl_swa_args* a, void* stream) {
  IVL_REQUIRE(a != nullptr, IVL_ERR_INVALID_ARG, "ivl_swa_fwd: NULL args");
  IVL_REQUIRE(a->q && a->k_new && a->v_new && a->o, IVL_ERR_INVALID_ARG, "ivl_swa_fwd: NULL q/k_new/v_new/o");
  IVL_REQUIRE(a->B > 0 && a->T > 0 && a->T_new >= a->T && a->Hq > 0 && a->Hkv > 0, IVL_ERR_INVALID_ARG,
              "ivl_swa_fwd: bad sizes B=%d T=%d T_new=%d Hq=%d Hkv=%d", a->B, a->T, a->T_new, a->Hq, a->Hkv);
  IVL_REQUIRE(a->d == SWA_D, IVL_ERR_UNSUPPORTED, "ivl_swa_fwd: head_dim %d unsupported (built for 128)", a->d);
  IVL_REQUIRE(a->Hq % a->Hkv == 0, IVL_ERR_INVALID_ARG, "ivl_swa_fwd: Hq=%d not a multiple of Hkv=%d", a->Hq, a->Hkv);
  IVL_REQUIRE(a->cache_capacity >= 0 && (a->cache_capacity == 0 || (a->k_cache && a->v_cache)), IVL_ERR_INVALID_ARG,
              "ivl_swa_fwd: cache_capacity=%d needs k_cache/v_cache", a->cache_capacity);
  IVL_REQUIRE(a->pos_dev != nullptr || a->pos >= 0, IVL_ERR_INVALID_ARG, "ivl_swa_fwd: negative pos");
  IVL_REQUIRE((a->rope_cos == nullptr) == (a->rope_sin == nullptr), IVL_ERR_INVALID_ARG, "ivl_swa_fwd: rope_cos and rope_sin go together");
  IVL_REQUIRE(a->rope_cos == nullptr || (a->T_new == a->T && a->rope_s0 % 8 == 0 && a->rope_s1 % 8 == 0 && a->rope_s0 >= 0 &&
                                          a->rope_s1 >= 0 && a->rope_s0 + a->rope_s1 <= 64),
              IVL_ERR_UNSUPPORTED, "ivl_swa_fwd: fused rope needs T_new == T and mrope sections that are multiples of 8 (got %d, %d)",
              a->rope_s0, a->rope_s1);
  IVL_REQUIRE(a->rope_cos == nullptr || a->mma_dtype == IVL_BF16 || !((long long)a->T * (a->Hq / a->Hkv) <= SWA_QT),
              IVL_ERR_UNSUPPORTED, "ivl_swa_fwd: the fp8 decode step takes rotated q / k (apply ivl_mrope_fwd first)");
  IVL_REQUIRE(a->mma_dtype == IVL_BF16 || a->mma_dtype == IVL_FP8_E4M3, IVL_ERR_INVALID_ARG,
              "ivl_swa_fwd: mma_dtype must be IVL_BF16 or IVL_FP8_E4M3 (got %d)", a->mma_dtype);
  IVL_REQUIRE(!a->append_new || (a->cache_capacity > 0 && a->T_new == a->T), IVL_ERR_INVALID_ARG,
              "ivl_swa_fwd: append_new needs a ring cache and T_new == T (got capacity %d, T_new %d, T %d)", a->cache_capacity,
              a->T_new, a->T);
  // new-key rows are addressed with 32-bit element offsets from the (batch, kv-head) base
  IVL_REQUIRE((long long)a->T_new * a->kn_st < (1LL << 32) && a->kn_st >= 0, IVL_ERR_UNSUPPORTED,
              "ivl_swa_fwd: T_new * kn_st = %lld elements exceeds the 32-bit row addressing of the kernel (split the call)",
              (long long)a->T_new * a->kn_st);
  // a long call over a FULL ring (the caller vouches for pos >= C): the 256-row form on a linear copy of the keys (swa_ring256.hip)
  if (a->pos_min >= a->cache_capacity && a->cache_capacity > 0 && a->window == a->cache_capacity + 1 && a->T_new == a->T &&
      a->mma_dtype == IVL_BF16 && a->pos_min >= 0 && (a->pos_dev != nullptr || a->pos >= a->cache_capacity)) {
    const size_t need = ivl_swa_ring256_workspace_bytes(a->B, a->T, a->Hq, a->Hkv, a->d, a->cache_capacity);
    if (need != 0 && a->workspace != nullptr && a->workspace_bytes >= need) return swa_ring256_launch(a, (hipStream_t)stream);
  }
  const int G = a->Hq / a->Hkv;
  const bool pack = (long long)a->T * G <= SWA_QT && G <= 16;
  // worst-case number of key tiles a workgroup walks (n_prev unknown under graph replay -> capacity)
  const long long max_prev = (long long)a->cache_capacity + (a->T_new - a->T);
  long long span = a->window > 0 ? (long long)a->window + 2 * SWA_QT : max_prev + a->T;
  if (span > max_prev + a->T) span = max_prev + a->T;
  const int max_tiles = (int)(span / SWA_KT) + 2;
  int nsplit = 1;
  if (pack) {
    nsplit = max_tiles;                    // decode: one key tile per workgroup (the K/V read is the whole cost)
    if (nsplit > SWA_MAX_SPLIT_PACK) nsplit = SWA_MAX_SPLIT_PACK;
  } else {
    nsplit = swa_base_nsplit(a->B, a->T, a->Hq);
    if (nsplit > max_tiles / 4) nsplit = max_tiles / 4;
    if (nsplit > SWA_MAX_SPLIT) nsplit = SWA_MAX_SPLIT;
  }
  if (nsplit < 1) nsplit = 1;

  SwaParams p;
  p.q = (const bf16_t*)a->q; p.k_new = (const bf16_t*)a->k_new; p.v_new = (const bf16_t*)a->v_new;
  p.k_cache = (const bf16_t*)a->k_cache; p.v_cache = (const bf16_t*)a->v_cache; p.o = (bf16_t*)a->o;
  p.q_sb = a->q_sb; p.q_st = a->q_st; p.q_sh = a->q_sh; p.kn_sb = a->kn_sb; p.kn_st = a->kn_st; p.kn_sh = a->kn_sh;
  p.vn_sb = a->kn_sb; p.vn_st = a->kn_st; p.vn_sh = a->kn_sh;
  p.B = a->B; p.T = a->T; p.T_new = a->T_new; p.Hq = a->Hq; p.Hkv = a->Hkv; p.C = a->cache_capacity; p.W = a->window;
  p.nsplit = nsplit; p.pos = a->pos; p.pos_dev = (const long long*)a->pos_dev; p.scaling = a->scaling;
  p.part_o = nullptr; p.part_ml = nullptr;
  p.rcos = (const bf16_t*)a->rope_cos; p.rsin = (const bf16_t*)a->rope_sin; p.rs0 = a->rope_s0; p.rs1 = a->rope_s1;
  if (nsplit > 1) {
    const size_t n_o = (size_t)a->B * nsplit * a->T * a->Hq * SWA_D;
    const size_t need = (n_o + (size_t)a->B * nsplit * a->T * a->Hq * 2) * sizeof(float);
    IVL_REQUIRE(a->workspace != nullptr && a->workspace_bytes >= need, IVL_ERR_WORKSPACE,
                "ivl_swa_fwd: workspace %zu bytes < required %zu (nsplit=%d)", a->workspace_bytes, need, nsplit);
    p.part_o = (float*)a->workspace;
    p.part_ml = p.part_o + n_o;
  }
  const int rows = pack ? a->T * G : a->T;
  const int qg = pack ? 1 : swa_qg(a->B, a->T, a->Hq);
  const bool prefill = !pack && a->T > SWA_QT;
  hipStream_t st = (hipStream_t)stream;
  // Fused M-RoPE of a prefill call: with split KV every split of a q-tile would rotate the same query rows, and in a long call
  // (one split) every q-tile would rotate the new keys it visits again (a 4096-token call over a full ring: each key tile
  // ~16 times, through registers instead of the LDS-DMA path) -- both cases rotate ONCE in a pre-pass; only short single-split
  // calls keep the rotation in the kernel (one launch less).
  // (measured again in round 3 with the tables requested in one round trip: the rotation of a 128-row query tile is ~700
  // VALU instructions per lane -- every product and sum is rounded to bf16 like the reference's -- and costs each of the eight
  // splits ~8,000 cycles that also slow its loader waves down: 34.7 vs 31.6 us per call at the step shape.)
  if (prefill && p.rcos != nullptr && (nsplit > 1 || a->T >= 4 * PF_QT)) {
    // rotate q and the call's keys once, into the workspace behind the partials (see swa_rope_prepass_kernel)
    // rounded up to 16 bytes: the pre-pass, the attention kernel and the ring append move q_rot / k_rot as 16-byte vectors
    // (B * nsplit * T * Hq * 130 floats is only 8-byte aligned for an odd row count; the +256 slack of
    // ivl_swa_workspace_bytes covers the padding)
    const size_t n_part = nsplit > 1 ? (((size_t)a->B * nsplit * a->T * a->Hq * (SWA_D + 2)) * sizeof(float) + 15) & ~(size_t)15 : 0;
    const size_t n_q = (size_t)a->B * a->T * a->Hq * SWA_D, n_k = (size_t)a->B * a->T * a->Hkv * SWA_D;
    IVL_REQUIRE(a->workspace != nullptr && a->workspace_bytes >= n_part + (n_q + n_k) * sizeof(bf16_t), IVL_ERR_WORKSPACE,
                "ivl_swa_fwd: workspace %zu bytes < required %zu (rope pre-pass)", a->workspace_bytes, n_part + (n_q + n_k) * sizeof(bf16_t));
    bf16_t* q_rot = (bf16_t*)((unsigned char*)a->workspace + n_part);
    bf16_t* k_rot = q_rot + n_q;
    const long long items = (long long)a->B * a->T * (a->Hq + a->Hkv) * 8;
    long long gb = (items + 255) / 256;
    if (gb > 2048) gb = 2048;
    hipLaunchKernelGGL(swa_rope_prepass_kernel, dim3((int)gb), dim3(256), 0, st, p.q, p.q_sb, p.q_st, p.q_sh, p.k_new, p.kn_sb, p.kn_st,
                       p.kn_sh, q_rot, k_rot, a->B, a->T, a->Hq, a->Hkv, p.rcos, p.rsin, p.rs0, p.rs1);
    int rc0 = check_launch("ivl_swa_fwd(rope pre-pass)");
    if (rc0 != IVL_OK) return rc0;
    p.q = q_rot; p.q_sb = (long long)a->T * a->Hq * SWA_D; p.q_st = (long long)a->Hq * SWA_D; p.q_sh = SWA_D;
    p.k_new = k_rot; p.kn_sb = (long long)a->T * a->Hkv * SWA_D; p.kn_st = (long long)a->Hkv * SWA_D; p.kn_sh = SWA_D;
    p.rcos = nullptr; p.rsin = nullptr;
  }
  p.n_qtiles = prefill ? (rows + PF_QT - 1) / PF_QT : (rows + SWA_QT * qg - 1) / (SWA_QT * qg);
  dim3 grid(p.n_qtiles * (pack ? a->Hkv : a->Hq) * a->B * nsplit);
  if (prefill) hipLaunchKernelGGL(swa_prefill_kernel, grid, dim3(P8_THREADS), 0, st, p.pos_dev, p);
  else if (pack && a->mma_dtype == IVL_FP8_E4M3)
    hipLaunchKernelGGL(swa_decode_fp8_kernel, dim3(a->Hkv * a->B * nsplit), dim3(256), 0, st, p.pos_dev, p);
  else if (pack) hipLaunchKernelGGL((swa_fwd_kernel<true, 1>), grid, dim3(256), 0, st, p.pos_dev, p);
  else hipLaunchKernelGGL((swa_fwd_kernel<false, 1>), grid, dim3(256), 0, st, p.pos_dev, p);
  int rc = check_launch("ivl_swa_fwd");
  if (rc != IVL_OK) return rc;
  AppendArgs ap;
  ap.first_block = -1;
  int append_blocks = 0;
  if (a->append_new) {
    ap.k_new = p.k_new; ap.v_new = p.v_new; ap.kn_sb = p.kn_sb; ap.kn_st = p.kn_st; ap.kn_sh = p.kn_sh;
    ap.vn_sb = p.vn_sb; ap.vn_st = p.vn_st; ap.vn_sh = p.vn_sh;
    ap.k_cache = (bf16_t*)a->k_cache; ap.v_cache = (bf16_t*)a->v_cache;
    ap.B = a->B; ap.T = a->T; ap.Hkv = a->Hkv; ap.C = a->cache_capacity; ap.pos = a->pos; ap.pos_dev = p.pos_dev;
    ap.rcos = p.rcos; ap.rsin = p.rsin; ap.rs0 = p.rs0; ap.rs1 = p.rs1;
    const int nt = a->T > a->cache_capacity ? a->cache_capacity : a->T;
    long long ab = ((long long)a->B * nt * a->Hkv * (SWA_D / 8) + 255) / 256;
    append_blocks = (int)(ab > 2048 ? 2048 : ab);
  }
  if (nsplit > 1) {
    const long long nrows = (long long)a->B * a->T * a->Hq;
    long long gb = (nrows * 64 + 255) / 256;
    if (gb > 4096) gb = 4096;
    if (append_blocks > 0) ap.first_block = (int)gb;
    const dim3 cg((int)gb + append_blocks);
    if (nsplit > 16) {
      long long wb = nrows > 4096 ? 4096 : nrows;                 // one workgroup per row
      if (append_blocks > 0) ap.first_block = (int)wb;
      hipLaunchKernelGGL(swa_combine_wide_kernel, dim3((int)wb + append_blocks), dim3(256), 0, st, p.part_o, p.part_ml, p.o, a->B,
                         a->T * a->Hq, nsplit, ap);
    }
    else if (prefill && nsplit <= 4) hipLaunchKernelGGL((swa_combine_kernel<4, true>), cg, dim3(256), 0, st, p.part_o, p.part_ml, p.o, a->B, a->T * a->Hq, nsplit, ap);
    else if (prefill && nsplit <= 8) hipLaunchKernelGGL((swa_combine_kernel<8, true>), cg, dim3(256), 0, st, p.part_o, p.part_ml, p.o, a->B, a->T * a->Hq, nsplit, ap);
    else if (prefill) hipLaunchKernelGGL((swa_combine_kernel<16, true>), cg, dim3(256), 0, st, p.part_o, p.part_ml, p.o, a->B, a->T * a->Hq, nsplit, ap);
    else if (nsplit <= 4) hipLaunchKernelGGL((swa_combine_kernel<4, false>), cg, dim3(256), 0, st, p.part_o, p.part_ml, p.o, a->B, a->T * a->Hq, nsplit, ap);
    else if (nsplit <= 8) hipLaunchKernelGGL((swa_combine_kernel<8, false>), cg, dim3(256), 0, st, p.part_o, p.part_ml, p.o, a->B, a->T * a->Hq, nsplit, ap);
    else hipLaunchKernelGGL((swa_combine_kernel<16, false>), cg, dim3(256), 0, st, p.part_o, p.part_ml, p.o, a->B, a->T * a->Hq, nsplit, ap);
    rc = check_launch("ivl_swa_fwd(combine)");
  } else if (append_blocks > 0) {
    ap.first_block = 0;
    hipLaunchKernelGGL(swa_cache_append_kernel, dim3(append_blocks), dim3(256), 0, st, ap);
    rc = check_launch("ivl_swa_fwd(append)");
  }
  return rc;
}

extern "C" int ivl_swa_cache_append(const void* k_new, const void* v_new, int64_t kn_sb, int64_t kn_st, int64_t kn_sh,
                                    void* k_cache, void* v_cache, int B, int T, int Hkv, int d, int cache_capacity,
                                    int64_t pos, const int64_t* pos_dev, const void* rope_cos, const void* rope_sin,
                                    int rope_s0, int rope_s1, void* stream) {
  IVL_REQUIRE(k_new && v_new && k_cache && v_cache, IVL_ERR_INVALID_ARG, "ivl_swa_cache_append: NULL pointer");
  IVL_REQUIRE(B > 0 && T > 0 && Hkv > 0 && cache_capacity > 0, IVL_ERR_INVALID_ARG, "ivl_swa_cache_append: bad sizes");
  IVL_REQUIRE(d == SWA_D, IVL_ERR_UNSUPPORTED, "ivl_swa_cache_append: head_dim %d unsupported", d);
  IVL_REQUIRE(pos_dev != nullptr || pos >= 0, IVL_ERR_INVALID_ARG, "ivl_swa_cache_append: negative pos");
  IVL_REQUIRE((rope_cos == nullptr) == (rope_sin == nullptr), IVL_ERR_INVALID_ARG, "ivl_swa_cache_append: rope_cos and rope_sin go together");
  IVL_REQUIRE(rope_cos == nullptr || (rope_s0 % 8 == 0 && rope_s1 % 8 == 0 && rope_s0 >= 0 && rope_s1 >= 0 && rope_s0 + rope_s1 <= 64),
              IVL_ERR_UNSUPPORTED, "ivl_swa_cache_append: mrope sections must be multiples of 8 (got %d, %d)", rope_s0, rope_s1);
  const int nt = T > cache_capacity ? cache_capacity : T;
  long long items = (long long)B * nt * Hkv * (SWA_D / 8);
  long long gb = (items + 255) / 256;
  if (gb > 2048) gb = 2048;
  AppendArgs ap;
  ap.k_new = (const bf16_t*)k_new; ap.v_new = (const bf16_t*)v_new; ap.kn_sb = kn_sb; ap.kn_st = kn_st; ap.kn_sh = kn_sh;
  ap.vn_sb = kn_sb; ap.vn_st = kn_st; ap.vn_sh = kn_sh;
  ap.k_cache = (bf16_t*)k_cache; ap.v_cache = (bf16_t*)v_cache; ap.B = B; ap.T = T; ap.Hkv = Hkv; ap.C = cache_capacity;
  ap.pos = (long long)pos; ap.pos_dev = (const long long*)pos_dev;
  ap.rcos = (const bf16_t*)rope_cos; ap.rsin = (const bf16_t*)rope_sin; ap.rs0 = rope_s0; ap.rs1 = rope_s1; ap.first_block = 0;
  hipLaunchKernelGGL(swa_cache_append_kernel, dim3((int)gb), dim3(256), 0, (hipStream_t)stream, ap);
  return check_launch("ivl_swa_cache_append");
}
