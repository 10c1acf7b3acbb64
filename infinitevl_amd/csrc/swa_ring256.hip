// Sliding-window attention of a LONG call over a FULL ring: the 256-row workgroup form.
//
// Where it runs: calls with a filled ring (pos >= C, asserted by the caller through ivl_swa_args.pos_min), W == C + 1,
// T a multiple of 256 and B * Hq * T / 256 >= 256 workgroups -- the 4096-token calls of BASELINE.json configs[3] after the first.
// Everything else stays on swa_prefill_kernel / swa_fwd_kernel (swa.hip).
//
// Two launches:
//  (1) swa_linearize_kernel -- the rope pre-pass of the long calls, re-cut: the ring in CHRONOLOGICAL order
//      followed by the call's rotated keys / values -> k_lin / v_lin [B, Hkv, C + T + 64, 128] (64 zero rows behind); and the ring
//      append itself (the thread that moves ring slot s out writes the call's token that lands in s: no reader of the old slot is
//      left, the attention kernel below reads the linear copy only).  With the keys linear in memory the attention kernel has NO
//      ring arithmetic, no wrap / seam tiles and no position: query row t of the call sees linear keys t .. t + C.
//  (2) swa_ring256_kernel -- 8 waves x 32 rows x ALL 64 keys of a tile (tools/proto/attn8_proto.hip modes 3 / 4): a 256-row q-tile of
//      one head walks NT = ceil((256 + C) / 64) key tiles starting at its own first row (tile-aligned: 256 | r0).  Tiles 0..3 cut
//      the lower band edge, tiles >= (C + 1) / 64 the upper one: they run a body with the band test; all tiles between are
//      interior for every row and run a body WITHOUT any mask or tile-kind code.  K / V tiles: unpadded 64 x 256 B images,
//      XOR-swizzled on the source side of the LDS-DMA (K 16-byte piece p of row r at p ^ (r & 15), V 64-byte granule g at
//      g ^ (r & 3)), 32 one-KB pieces per tile issued by the compute waves 0-3 (eight each) between their row maximum and their
//      exponentials, a 4-stage ring filled three tiles ahead, one barrier per tile; waves 4-7 (the SIMD partners of 0-3) request
//      nothing and run their phases rotated by one (softmax(t), PV(t), QK^T(t + 1)); no static priority (it costs 3-35 x here).
#include "ivl_common.h"
#include "swa_shared.h"
#include <type_traits>

namespace ivl {

constexpr int R2_ROWS = 256;                 // query rows per workgroup
constexpr int R2_KT = 64;                    // keys per tile
constexpr int R2_STAGE = 2 * R2_KT * 256;    // K image + V image = 32 KB
constexpr int R2_NST = 4;                    // stages
constexpr int R2_AHEAD = 3;                  // tiles requested ahead
constexpr int R2_LDS = R2_NST * R2_STAGE;    // 128 KB
#ifndef R2_DMA_BY
#define R2_DMA_BY 2                          // who requests the tiles of the loop: 0 every wave four pieces per tile; 1 / 2 only waves
                                             // 4-7 / 0-3, eight each (2 = default: -3 ... -6 % against 0 on three boxes; 1: +1.5 %)
#endif
#ifndef R2_DMA_AT
#define R2_DMA_AT 0                          // where the four DMA pieces of a tile are issued: 0 behind the row maximum, 1 behind the
#endif                                       // exponentials, 2 two and two
constexpr int R2_PAD_ROWS = 64;              // zero rows behind the linear keys (the last tile of the last q-tile ends 1 key late)

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// a wave-uniform pointer the compiler has lost track of (derived from blockIdx through divisions): back into SGPRs for the
// scalar-base operand of the LDS-DMA instruction
__device__ __forceinline__ const unsigned char* uniform_ptr(const unsigned char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  return (const unsigned char*)(((unsigned long long)hi << 32) | lo);
}

struct LinArgs {
  const bf16_t* q; long long q_sb, q_st, q_sh; bf16_t* q_rot; int Hq;      // q_rot == NULL: the attention kernel rotates q itself
  const bf16_t* k_new; const bf16_t* v_new; long long kn_sb, kn_st, kn_sh;
  bf16_t* k_cache; bf16_t* v_cache;
  bf16_t* k_lin; bf16_t* v_lin;
  int B, T, Hkv, C; long long pos; const long long* pos_dev;
  const bf16_t* rcos; const bf16_t* rsin; int rs0, rs1;
  int append;
};

// one work item = one pair of 16-byte pieces (channels 8c .. 8c+7 and 8c+64 .. 8c+71) of one row: the unit of the rotation
__global__ __launch_bounds__(256) void swa_linearize_kernel(LinArgs a) {
  const long long pos = a.pos_dev ? *a.pos_dev : a.pos;
  const int pos_slot = mod_pos(pos, a.C);
  const long long Lp = (long long)a.C + a.T + R2_PAD_ROWS;
  const long long nN = (long long)a.B * a.T * a.Hkv * 8, nR = (long long)a.B * a.Hkv * a.C * 8, nP = (long long)a.B * a.Hkv * R2_PAD_ROWS * 8,
                  nQ = a.q_rot != nullptr ? (long long)a.B * a.T * a.Hq * 8 : 0;
  const long long plane = (long long)a.B * a.T * SWA_D;
  const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < nN + nR + nP + nQ; idx += (long long)gridDim.x * blockDim.x) {
    if (idx < nN) {                                            // (b, t, hk, c): the call's keys / values behind the ring's
      const long long i2 = idx;
      const int c = (int)(i2 & 7);
      int hk, t;
      const long long bt = divmod_idx(i2 >> 3, a.Hkv, hk);
      const int b = (int)divmod_idx(bt, a.T, t);
      const long long so = (long long)b * a.kn_sb + (long long)t * a.kn_st + (long long)hk * a.kn_sh;
      u32x4 lo = *(const u32x4*)(a.k_new + so + 8 * c), hi = *(const u32x4*)(a.k_new + so + 8 * c + 64);
      if (a.rcos != nullptr) rope_pair(lo, hi, a.rcos, a.rsin, plane, bt * SWA_D, 8 * c, a.rs0, a.rs1);
      const long long d0 = (((long long)b * a.Hkv + hk) * Lp + a.C + t) * SWA_D;
      *(u32x4*)(a.k_lin + d0 + 8 * c) = lo;
      *(u32x4*)(a.k_lin + d0 + 8 * c + 64) = hi;
      *(u32x4*)(a.v_lin + d0 + 8 * c) = *(const u32x4*)(a.v_new + so + 8 * c);
      *(u32x4*)(a.v_lin + d0 + 8 * c + 64) = *(const u32x4*)(a.v_new + so + 8 * c + 64);
    } else if (idx < nN + nR) {                           // (b, hk, slot, c): ring slot out in chronological order, new token in
      const long long i2 = idx - nN;
      const int c = (int)(i2 & 7);
      int s, hk;
      const long long bh_ = divmod_idx(i2 >> 3, a.C, s);
      const int b = (int)divmod_idx(bh_, a.Hkv, hk);
      int j = s - pos_slot;                                    // position pos - C + j lives in slot s
      if (j < 0) j += a.C;
      const long long ro = (bh_ * a.C + s) * SWA_D, d0 = (bh_ * Lp + j) * SWA_D;
      *(u32x4*)(a.k_lin + d0 + 8 * c) = *(const u32x4*)(a.k_cache + ro + 8 * c);
      *(u32x4*)(a.k_lin + d0 + 8 * c + 64) = *(const u32x4*)(a.k_cache + ro + 8 * c + 64);
      *(u32x4*)(a.v_lin + d0 + 8 * c) = *(const u32x4*)(a.v_cache + ro + 8 * c);
      *(u32x4*)(a.v_lin + d0 + 8 * c + 64) = *(const u32x4*)(a.v_cache + ro + 8 * c + 64);
      if (a.append && j < a.T) {                               // token t goes to slot (pos + t) % C: t = j (mod C), the newest one < T
        const int t = j + ((a.T - 1 - j) / a.C) * a.C;
        const long long so = (long long)b * a.kn_sb + (long long)t * a.kn_st + (long long)hk * a.kn_sh;
        u32x4 lo = *(const u32x4*)(a.k_new + so + 8 * c), hi = *(const u32x4*)(a.k_new + so + 8 * c + 64);
        if (a.rcos != nullptr) rope_pair(lo, hi, a.rcos, a.rsin, plane, ((long long)b * a.T + t) * SWA_D, 8 * c, a.rs0, a.rs1);
        const u32x4 v0 = *(const u32x4*)(a.v_new + so + 8 * c), v1 = *(const u32x4*)(a.v_new + so + 8 * c + 64);
        *(u32x4*)(a.k_cache + ro + 8 * c) = lo;
        *(u32x4*)(a.k_cache + ro + 8 * c + 64) = hi;
        *(u32x4*)(a.v_cache + ro + 8 * c) = v0;
        *(u32x4*)(a.v_cache + ro + 8 * c + 64) = v1;
      }
    } else if (idx >= nN + nR + nP) {                          // (b, t, h, c): rotated q
      const long long i2 = idx - nN - nR - nP;
      const int c = (int)(i2 & 7);
      int h, t;
      const long long bt = divmod_idx(i2 >> 3, a.Hq, h);
      const int b = (int)divmod_idx(bt, a.T, t);
      const bf16_t* src = a.q + (long long)b * a.q_sb + (long long)t * a.q_st + (long long)h * a.q_sh;
      u32x4 lo = *(const u32x4*)(src + 8 * c), hi = *(const u32x4*)(src + 8 * c + 64);
      rope_pair(lo, hi, a.rcos, a.rsin, plane, bt * SWA_D, 8 * c, a.rs0, a.rs1);
      bf16_t* dst = a.q_rot + (bt * a.Hq + h) * SWA_D;
      *(u32x4*)(dst + 8 * c) = lo;
      *(u32x4*)(dst + 8 * c + 64) = hi;
    } else {                                                   // zero rows behind the linear keys
      const long long i2 = idx - nN - nR;
      const int c = (int)(i2 & 7);
      int r;
      const long long bh_ = divmod_idx(i2 >> 3, R2_PAD_ROWS, r);
      const long long d0 = (bh_ * Lp + a.C + a.T + r) * SWA_D;
      *(u32x4*)(a.k_lin + d0 + 8 * c) = zero;
      *(u32x4*)(a.k_lin + d0 + 8 * c + 64) = zero;
      *(u32x4*)(a.v_lin + d0 + 8 * c) = zero;
      *(u32x4*)(a.v_lin + d0 + 8 * c + 64) = zero;
    }
  }
}

struct Ring256Params {
  const bf16_t* q; long long q_sb, q_st, q_sh;          // the call's q (UN-rotated when rcos != NULL: rotated below, once per workgroup)
  const bf16_t* k_lin; const bf16_t* v_lin; bf16_t* o;
  int B, T, Hq, Hkv, C, ntiles, j_hi; long long lin_rows; float sc;
  const bf16_t* rcos; const bf16_t* rsin; int rs0, rs1;
};

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void swa_ring256_kernel(Ring256Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi5 = lane >> 5, l15 = lane & 15;
  // workgroup -> (b, kv head, q head of the group, q-tile): XCD x (= blockIdx % 8) takes a contiguous eighth of the work list,
  // i.e. whole kv heads (their K / V stay in one L2), neighbouring q-tiles (overlapping windows) next to each other
  const int NQ = p.T / R2_ROWS;
  const int total = (int)gridDim.x;
  int wid = (int)blockIdx.x;
  if ((total & 7) == 0) wid = (wid & 7) * (total >> 3) + (wid >> 3);
  const int qt = wid % NQ, bh = wid / NQ, h = bh % p.Hq, b = bh / p.Hq, hk = h / (p.Hq / p.Hkv);
  const int r0 = qt * R2_ROWS;
  const float sc = p.sc;
  const int NT = p.ntiles, j_hi = p.j_hi;

  const unsigned char* kbase = (const unsigned char*)(p.k_lin + (((long long)b * p.Hkv + hk) * p.lin_rows + r0) * SWA_D) + wave * 1024;
  const unsigned char* vbase = (const unsigned char*)(p.v_lin + (((long long)b * p.Hkv + hk) * p.lin_rows + r0) * SWA_D) + wave * 1024;
  const unsigned int lds_base = (unsigned int)(size_t)smem;
  // LDS-DMA: wave w moves chunks (4 rows x 256 B) w and w + 8 of the K image and of the V image of a tile
  const int r_in = lane >> 4, pp = lane & 15;
  const unsigned int k_src = (unsigned int)(r_in * 256 + ((pp ^ ((4 * wave + r_in) & 15)) << 4));
  const unsigned int v_src = (unsigned int)(r_in * 256 + (((((pp >> 2) ^ r_in) << 2) | (pp & 3)) << 4));
  auto dma_tile = [&](int jt, int j0, int j1) __attribute__((always_inline)) {
    const unsigned int dst0 = lds_base + (unsigned int)(jt & (R2_NST - 1)) * R2_STAGE + 1024u * wave;
#pragma unroll
    for (int j = j0; j < j1; ++j)
#pragma unroll
      for (int isv = 0; isv < 2; ++isv) {
        const unsigned char* base = uniform_ptr((isv ? vbase : kbase) + (size_t)jt * (R2_KT * 256) + j * 8192);
        const unsigned int dst = __builtin_amdgcn_readfirstlane(dst0 + (isv ? 16384u : 0u) + 8192u * j);
        unsigned int keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(isv ? v_src : k_src), "s"(dst), "s"(base) : "memory");
      }
  };
#if R2_DMA_BY
  // role split: only one half of the waves (R2_DMA_BY 1: the rotated half 4-7, 2: waves 0-3) requests the tiles of the loop, eight pieces
  // per wave and tile: wave w' = wave & 3 moves chunks w', w' + 4, w' + 8, w' + 12 of both images
  const bool dma_role = R2_DMA_BY == 1 ? wave >= 4 : wave < 4;       // (3: as 2, the K pieces behind the row maximum, the V pieces behind the exponentials)
  const unsigned int k_src8 = (unsigned int)(r_in * 256 + ((pp ^ ((4 * (wave & 3) + r_in) & 15)) << 4));
  auto dma_tile8 = [&](int jt, int v0 = 0, int v1 = 2) __attribute__((always_inline)) {
    const unsigned int dst0 = lds_base + (unsigned int)(jt & (R2_NST - 1)) * R2_STAGE + 1024u * (wave & 3);
    const long long wofs = (long long)jt * (R2_KT * 256) + ((wave & 3) - wave) * 1024;      // kbase / vbase carry wave * 1024
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int isv = v0; isv < v1; ++isv) {
        const unsigned char* base = uniform_ptr((isv ? vbase : kbase) + wofs + j * 4096);
        const unsigned int dst = __builtin_amdgcn_readfirstlane(dst0 + (isv ? 16384u : 0u) + 4096u * j);
        unsigned int keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(isv ? v_src : k_src8), "s"(dst), "s"(base) : "memory");
      }
  };
#endif
  dma_tile(0, 0, 2);
  dma_tile(1, 0, 2);
  dma_tile(2, 0, 2);

  // Q^T fragments (B operand): lane = query row, d = 16 kd + 8 hi .. +7
  const int trow = r0 + 32 * wave + l31;
  const bf16_t* qrow = p.q + (long long)b * p.q_sb + (long long)trow * p.q_st + (long long)h * p.q_sh;
  u32x4 qf[8];
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) qf[kd] = *(const u32x4*)(qrow + 16 * kd + 8 * hi5);
  // fused M-RoPE of the query rows (the 64-apart partner of fragment kd is the lane's own fragment kd + 4): rope_pair's arithmetic,
  // bit-identical to the pre-pass of the 128-row path; ~700 VALU instructions per lane once per 68-tile workgroup
  if (p.rcos != nullptr) {
#pragma unroll
    for (int kd = 0; kd < 4; ++kd)
      rope_pair(qf[kd], qf[kd + 4], p.rcos, p.rsin, (long long)p.B * p.T * SWA_D, ((long long)b * p.T + trow) * SWA_D, 16 * kd + 8 * hi5, p.rs0, p.rs1);
  }
  // the compiler's vmcnt bookkeeping must see the Q loads complete BEFORE the loop (it does not model the inline-asm DMA: a
  // pending load at the loop header makes it wait in front of the first MFMAs of every iteration); its wait here also covers the
  // three tiles requested above (older requests)
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) asm volatile("" : "+v"(qf[kd]));
  f32x16 oacc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[mt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // fragment addresses: K row 32 kh + l31, piece (2 kd + hi5) ^ l15 == base ^ (kd << 5); V granule (mt ^ rr) == base ^ (mt << 6)
  const unsigned int k_fb = (unsigned int)(l31 * 256 + ((l15 >> 1) << 5) + ((hi5 ^ (l15 & 1)) << 4));
  const unsigned int v_fb = (unsigned int)(16384 + (4 * hi5 + (l15 >> 2)) * 256 + ((l15 >> 2) << 6) + 32 * ((lane >> 4) & 1) + 8 * (l15 & 3));
  unsigned int ak[8], av[4];
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) ak[kd] = k_fb ^ (unsigned int)(kd << 5);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) av[mt] = v_fb ^ (unsigned int)(mt << 6);
  // band of this lane's row rr = 32 wave + l31 (tile-relative key kk = c_r + 4 hi5 of tile j is visible iff rr <= 64 j + kk <= rr + C)
  const int band_lo = 32 * wave + l31 - 4 * hi5;

  f32x16 s[2];
  u32x4 pf[2][2];
  auto qk = [&](int jt) __attribute__((always_inline)) {
    const unsigned int sb = (unsigned int)(jt & (R2_NST - 1)) * R2_STAGE;
    u32x4 fr[2][8];
#pragma unroll
    for (int kd = 0; kd < 8; ++kd)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) fr[kh][kd] = *(const u32x4*)(smem + ak[kd] + sb + 32 * kh * 256);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kh][r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 8; ++kd)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) s[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_mfma(fr[kh][kd]), as_mfma(qf[kd]), s[kh], 0, 0, 0);
  };
  // row maximum + (rare) rescale.  MASK: band test first (invisible scores -> -inf), rows that have not seen a key yet stay at
  // m_run = -inf with a factor of 1.
  auto smax = [&](auto mask_tag, int jt) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mask_tag)::value;
    if (MASK) {
      const int lo = band_lo - 64 * jt, hi = lo + p.C;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c_r = 32 * kh + (r & 3) + 8 * (r >> 2);
          s[kh][r] = (c_r >= lo && c_r <= hi) ? s[kh][r] : -INFINITY;
        }
    }
    float rmax = vmax2(__builtin_fmaxf(s[0][0], s[0][1]), s[1][0]);
    rmax = vmax3(rmax, s[0][2], s[0][3]);
#pragma unroll
    for (int r = 4; r < 16; r += 2) rmax = vmax3(rmax, s[0][r], s[0][r + 1]);
    rmax = vmax2(rmax, s[1][1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) rmax = vmax3(rmax, s[1][r], s[1][r + 1]);
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
    rmax = vmax2(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * sc;
    const float m_new = vmax2(m_run, rmax);
    // lazy exponent reference (as in swa_prefill_kernel): m_run follows the maximum only when some row of the wave outgrew it by
    // more than 2^8, so the probabilities are 2^(s - m_run) <= 2^8 and the 65 multiplies of the rescale are rare
    if (__any(m_new > m_run + 8.0f)) {
      float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      if (MASK) alpha = m_new == -INFINITY ? 1.0f : alpha;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) oacc[mt] *= alpha;
      l_run *= alpha;
      m_run = m_new;
    }
  };
  auto sexp = [&](auto mask_tag) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const float mu = (MASK && m_run == -INFINITY) ? 0.f : m_run;
    float rsum = 0.f;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kh][r], sc, -mu));
        s[kh][r] = pr;
        rsum += pr;
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        pf[kh][ks] = u32x4{pack2bf(s[kh][8 * ks + 0], s[kh][8 * ks + 1]), pack2bf(s[kh][8 * ks + 2], s[kh][8 * ks + 3]),
                           pack2bf(s[kh][8 * ks + 4], s[kh][8 * ks + 5]), pack2bf(s[kh][8 * ks + 6], s[kh][8 * ks + 7])};
    }
    l_run += rsum;
  };
  auto pv = [&](int jt) __attribute__((always_inline)) {
    const unsigned int sb = (unsigned int)(jt & (R2_NST - 1)) * R2_STAGE;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 fv[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const unsigned char* vp = smem + av[mt] + sb + (32 * kh + 16 * ks) * 256;
          const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)vp);
          const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp + 8 * 256));
          u32x2 w0, w1;
          __builtin_memcpy(&w0, &a0, 8);
          __builtin_memcpy(&w1, &a1, 8);
          fv[mt] = u32x4{w0.x, w0.y, w1.x, w1.y};
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_mfma(fv[mt]), as_mfma(pf[kh][ks]), oacc[mt], 0, 0, 0);
      }
  };
  auto tile_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // one tile.  The request of tile t + 3 overwrites the stage of tile t - 1, whose last readers (PV of the waves, QK^T of the
  // rotated half one iteration earlier) passed the barrier that closed iteration t - 1.  At the end of iteration t the wave's own
  // pieces of the tiles <= t + 2 have landed (the four newest requests may fly), so behind the barrier the rotated half may read
  // K(t + 2) in iteration t + 1.  In the last iterations nothing is requested and the wait is for everything.
  // tile j holds no visible key for ANY row of this wave (band edges only): the wave skips its products and softmax, not the
  // requests, the wait and the barrier
  auto vis = [&](int j) __attribute__((always_inline)) {
    return 64 * j + 63 >= 32 * wave && 64 * j <= 32 * wave + 31 + p.C;
  };
  auto body = [&](auto rot_tag, auto mask_tag, int t) __attribute__((always_inline)) {
    constexpr bool ROT = decltype(rot_tag)::value;
    constexpr bool MASK = decltype(mask_tag)::value;
    const bool on = !MASK || vis(t);
    if (on) {
      if (!ROT) qk(t);
      smax(mask_tag, t);
    }
    const bool more = !MASK || t + R2_AHEAD < NT;
#if R2_DMA_BY == 3
    if (more && dma_role) dma_tile8(t + R2_AHEAD, 0, 1);
#elif R2_DMA_BY
    if (more && dma_role) dma_tile8(t + R2_AHEAD);
#elif !defined(R2_NO_DMA) && R2_DMA_AT == 0
    if (more) dma_tile(t + R2_AHEAD, 0, 2);
#elif !defined(R2_NO_DMA) && R2_DMA_AT == 2
    if (more) dma_tile(t + R2_AHEAD, 0, 1);
#endif
    if (on) sexp(mask_tag);
#if R2_DMA_BY == 3
    if (more && dma_role) dma_tile8(t + R2_AHEAD, 1, 2);
#endif
#if !defined(R2_NO_DMA) && R2_DMA_AT == 1
    if (more) dma_tile(t + R2_AHEAD, 0, 2);
#elif !defined(R2_NO_DMA) && R2_DMA_AT == 2
    if (more) dma_tile(t + R2_AHEAD, 1, 2);
#endif
    if (on) pv(t);
    if (ROT && t + 1 < NT && vis(t + 1)) qk(t + 1);
#if R2_DMA_BY
    if (more && dma_role) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    tile_barrier();
  };
  auto run = [&](auto rot_tag) __attribute__((always_inline)) {
    constexpr bool ROT = decltype(rot_tag)::value;
    if (ROT) {
#ifdef R2_PRIO      // static priority 1 for the rotated half (what swa_prefill_kernel does) makes THIS kernel 35x slower (5.57 ms
      __builtin_amdgcn_s_setprio(1);      // vs 157 us per launch, same box): the prio-0 waves also issue the LDS-DMA the others wait for
#endif
      if (vis(0)) qk(0);
    }
#ifdef R2_PRIO0
    if (!ROT) __builtin_amdgcn_s_setprio(1);
#endif
    const int n_lo = NT < 4 ? NT : 4;
    int t = 0;
    for (; t < n_lo; ++t) body(rot_tag, std::true_type{}, t);
    for (; t < j_hi; ++t) body(rot_tag, std::false_type{}, t);
    for (; t < NT; ++t) body(rot_tag, std::true_type{}, t);
#ifdef R2_PRIO
    if (ROT) __builtin_amdgcn_s_setprio(0);
#endif
#ifdef R2_PRIO0
    if (!ROT) __builtin_amdgcn_s_setprio(0);
#endif
  };
#ifdef R2_NO_ROT
  run(std::false_type{});
#else
  if (wave < 4) run(std::false_type{});
  else run(std::true_type{});
#endif

  // ---- normalise and store O [B,T,Hq,128]: lane (row l31, half hi5) owns d = 32 mt + 8 qd + 4 hi5 .. +3
  {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    // through LDS (all stages are free behind the last barrier; every wave uses its own 8 KB): rows land as whole 256-byte rows,
    // 16-byte piece P of row r at P ^ (r & 15), and leave as 16-byte stores of four complete rows per instruction
    unsigned char* ob = smem + wave * 8192;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int P = 4 * mt + qd;
        *(u32x2*)(ob + l31 * 256 + ((P ^ l15) << 4) + 8 * hi5) =
            u32x2{pack2bf(oacc[mt][4 * qd] * inv, oacc[mt][4 * qd + 1] * inv), pack2bf(oacc[mt][4 * qd + 2] * inv, oacc[mt][4 * qd + 3] * inv)};
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bf16_t* orow = p.o + (((long long)b * p.T + r0 + 32 * wave) * p.Hq + h) * SWA_D;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 4 * i + (lane >> 4), P = lane & 15;
      const u32x4 v = *(const u32x4*)(ob + row * 256 + ((P ^ (row & 15)) << 4));
      store_out16(orow + (long long)row * p.Hq * SWA_D + 8 * P, v);
    }
  }
}

}  // namespace ivl

using namespace ivl;

// Workspace of the 256-row path: rotated q, linear keys and values.  0: the shape does not take this path.
extern "C" size_t ivl_swa_ring256_workspace_bytes(int B, int T, int Hq, int Hkv, int d, int cache_capacity) {
  if (B <= 0 || T <= 0 || Hq <= 0 || Hkv <= 0 || d != SWA_D || cache_capacity < 511) return 0;
  if (T % R2_ROWS != 0 || Hq % Hkv != 0 || (long long)B * Hq * (T / R2_ROWS) < 256) return 0;
  const size_t lin = (size_t)B * Hkv * ((size_t)cache_capacity + T + R2_PAD_ROWS) * SWA_D * sizeof(bf16_t);
  return 2 * lin + (size_t)B * T * Hq * SWA_D * sizeof(bf16_t) + 256;       // + rotated q
}

namespace ivl {
// Called by ivl_swa_fwd (swa.hip) for the calls that qualify; returns IVL_OK after both launches.
int swa_ring256_launch(const ivl_swa_args* a, hipStream_t st) {
  const size_t need = ivl_swa_ring256_workspace_bytes(a->B, a->T, a->Hq, a->Hkv, a->d, a->cache_capacity);
  IVL_REQUIRE(need != 0 && a->workspace != nullptr && a->workspace_bytes >= need, IVL_ERR_WORKSPACE,
              "ivl_swa_fwd(256-row path): workspace %zu bytes < required %zu", a->workspace_bytes, need);
  const long long Lp = (long long)a->cache_capacity + a->T + R2_PAD_ROWS;
  bf16_t* k_lin = (bf16_t*)(((size_t)a->workspace + 15) & ~(size_t)15);
  bf16_t* v_lin = k_lin + (size_t)a->B * a->Hkv * Lp * SWA_D;
  // the query rotation stays in the pre-pass: inside the attention kernel (Ring256Params.rcos) it sits in front of every workgroup's
  // first product with nothing to overlap it -- +9 us per launch for -7.5 us of pre-pass (same box, in a configs[3] call)
  bf16_t* q_rot = a->rope_cos != nullptr ? v_lin + (size_t)a->B * a->Hkv * Lp * SWA_D : nullptr;
  LinArgs la;
  la.q = (const bf16_t*)a->q; la.q_sb = a->q_sb; la.q_st = a->q_st; la.q_sh = a->q_sh; la.q_rot = q_rot; la.Hq = a->Hq;
  la.k_new = (const bf16_t*)a->k_new; la.v_new = (const bf16_t*)a->v_new; la.kn_sb = a->kn_sb; la.kn_st = a->kn_st; la.kn_sh = a->kn_sh;
  la.k_cache = (bf16_t*)a->k_cache; la.v_cache = (bf16_t*)a->v_cache; la.k_lin = k_lin; la.v_lin = v_lin;
  la.B = a->B; la.T = a->T; la.Hkv = a->Hkv; la.C = a->cache_capacity; la.pos = a->pos; la.pos_dev = (const long long*)a->pos_dev;
  la.rcos = (const bf16_t*)a->rope_cos; la.rsin = (const bf16_t*)a->rope_sin; la.rs0 = a->rope_s0; la.rs1 = a->rope_s1;
  la.append = a->append_new ? 1 : 0;
  const long long items = ((long long)a->B * a->T * (a->Hkv + (q_rot ? a->Hq : 0)) + (long long)a->B * a->Hkv * (a->cache_capacity + R2_PAD_ROWS)) * 8;
  long long gb = (items + 255) / 256;
  if (gb > 4096) gb = 4096;
  hipLaunchKernelGGL(swa_linearize_kernel, dim3((int)gb), dim3(256), 0, st, la);
  int rc = check_launch("ivl_swa_fwd(linearize pre-pass)");
  if (rc != IVL_OK) return rc;
  Ring256Params p;
  p.q = (const bf16_t*)a->q; p.q_sb = a->q_sb; p.q_st = a->q_st; p.q_sh = a->q_sh;
  p.rcos = nullptr; p.rsin = nullptr; p.rs0 = p.rs1 = 0;
  if (q_rot != nullptr) { p.q = q_rot; p.q_sb = (long long)a->T * a->Hq * SWA_D; p.q_st = (long long)a->Hq * SWA_D; p.q_sh = SWA_D; }
  p.k_lin = k_lin; p.v_lin = v_lin; p.o = (bf16_t*)a->o;
  p.B = a->B; p.T = a->T; p.Hq = a->Hq; p.Hkv = a->Hkv; p.C = a->cache_capacity;
  p.ntiles = (R2_ROWS + a->cache_capacity + R2_KT - 1) / R2_KT;
  p.j_hi = (a->cache_capacity + 1) / R2_KT;
  p.lin_rows = Lp; p.sc = a->scaling * LOG2E;
  static bool attr_set[64];                       // per device: a function attribute belongs to the device's copy of the code object
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!__atomic_load_n(&attr_set[dev & 63], __ATOMIC_ACQUIRE)) {
    (void)hipFuncSetAttribute((const void*)swa_ring256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, R2_LDS);
    __atomic_store_n(&attr_set[dev & 63], true, __ATOMIC_RELEASE);
  }
  hipLaunchKernelGGL(swa_ring256_kernel, dim3(a->B * a->Hq * (a->T / R2_ROWS)), dim3(512), R2_LDS, st, p);
  return check_launch("ivl_swa_fwd(256-row attention)");
}
}  // namespace ivl
