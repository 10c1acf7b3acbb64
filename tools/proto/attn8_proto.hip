// PROTOTYPE (developer experiment, not part of libivl_hip.so): the 8-wave x 32-row x all-64-keys attention tile loop of
// cdna_hip_programming.md Appendix B ("8-warp 32x32 ladder"), as a stand-alone kernel -- what VERDICT r4 #2 asks to be measured
// BEFORE the product kernel is rebuilt around it.
//   workgroup = 256 query rows of one head = 8 waves x 32 rows (two waves per SIMD, <= 256 VGPRs, no loader waves);
//   a wave owns its 32 rows against ALL 64 keys of a tile: S^T = K Q^T on 2 x 8 v_mfma_f32_32x32x16_bf16 (two key halves, one
//   shared row maximum), P^T packed in place, O^T += V^T P^T on 16 MFMAs -- 32 MFMAs per wave and tile between two barriers
//   (the 128-row product kernel: 8 + 8 in two barrier segments, and a merge of the key halves at the end);
//   K / V tiles: padded LDS images (K rows 272 B, V rows 320 B), a 3-stage ring filled by LDS-DMA issued by the COMPUTE waves
//   (5 one-KB pieces per wave and tile, counted vmcnt, one barrier per tile).
// MODE 0: the tile stays resident in LDS (compute-loop ceiling); 1: every tile is fetched again by DMA (the real data path, L2
// hits), pieces issued right behind the barrier; 3: pieces issued between the row maximum and the exponentials; 4: MODE 3 + the
// second half of the waves runs its phases rotated by one (softmax, PV, QK^T of the next tile).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __bf16 bf16_native2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) { const bf16_native2 v = {(__bf16)lo, (__bf16)hi}; return __builtin_bit_cast(unsigned int, v); }
__device__ __forceinline__ mfma_bf16x8 mf(u32x4 v) { mfma_bf16x8 r; __builtin_memcpy(&r, &v, 16); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

constexpr int KS = 272, VS = 320, KT = 64, D = 128;
constexpr int K_BYTES = KT * KS, V_BYTES = KT * VS, STAGE = K_BYTES + V_BYTES;      // 17,408 + 20,480 = 37,888 B
constexpr int NST = 3;
constexpr int DUMMY = NST * STAGE;                                                  // 1 KB landing area of the padding DMA pieces
constexpr int LDS_BYTES = DUMMY + 1024;
constexpr int NPIECE = 5;                                                           // DMA instructions per wave and tile (37 real + 3 dummy)

template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn8_proto(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                   bf16_t* __restrict__ o, int ntiles, float sc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi5 = lane >> 5, l15 = lane & 15;
  const int row0 = blockIdx.x * 256 + wave * 32;

  // ---- DMA pieces of this wave: global piece index i = wave + 8 j (j < 5): K image pieces 0..16, V image pieces 17..36, >= 37 dummy
  unsigned int src_off[NPIECE];         // per-lane byte offset into the [64][256 B] source tile (k or v)
  unsigned int lds_off[NPIECE];         // wave-uniform byte offset inside a stage (or DUMMY - stage base: handled below)
  bool is_v[NPIECE], dummy[NPIECE];
#pragma unroll
  for (int j = 0; j < NPIECE; ++j) {
    const int i = wave + 8 * j;
    dummy[j] = i >= 37;
    is_v[j] = i >= 17;
    const int c = is_v[j] ? i - 17 : i;
    const int ppr = is_v[j] ? VS / 16 : KS / 16;
    const int qd = 64 * c + lane, r = qd / ppr, col = qd % ppr;
    const bool pad = col >= 16 || r >= KT || dummy[j];
    src_off[j] = pad ? 0u : (unsigned int)(r * 256 + col * 16);
    lds_off[j] = (unsigned int)((is_v[j] ? K_BYTES : 0) + 1024 * c);
  }
  const unsigned int lds_base = (unsigned int)(size_t)smem;
  auto dma_tile = [&](int st) {
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const bf16_t* base = is_v[j] ? v : k;
      const unsigned int dst = dummy[j] ? lds_base + DUMMY : lds_base + (unsigned int)st * STAGE + lds_off[j];
      unsigned int keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src_off[j]), "s"(dst), "s"(base) : "memory");
    }
  };

  // ---- Q^T fragments (B operand): lane = query row, d = 16 kd + 8 hi .. +7
  u32x4 qf[8];
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) qf[kd] = *(const u32x4*)(q + (size_t)(row0 + l31) * D + 16 * kd + 8 * hi5);
  // the compiler's vmcnt bookkeeping must see the Q loads complete BEFORE the loop: it does not model the inline-asm DMA, and a
  // pending load at the loop header makes it place vmcnt(7..0) waits in front of the first MFMAs of every iteration -- which in
  // the steady state wait for the DMA pieces issued a few instructions earlier
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) asm volatile("" : "+v"(qf[kd]));
  f32x16 oacc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[mt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  if (MODE == 0) {
    for (int i = tid; i < KT * 16; i += 512) {
      const int r = i >> 4, c = i & 15;
      *(u32x4*)(smem + r * KS + c * 16) = *(const u32x4*)(k + r * D + c * 8);
      *(u32x4*)(smem + K_BYTES + r * VS + c * 16) = *(const u32x4*)(v + r * D + c * 8);
    }
  } else {
    dma_tile(0);
    dma_tile(1);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");       // tile 0 landed (tile 1 may fly)
  }
  __syncthreads();

  const int k_off = l31 * KS + 16 * hi5;
  const int v_off = K_BYTES + (4 * hi5 + (l15 >> 2)) * VS + (16 * ((lane >> 4) & 1) + 4 * (l15 & 3)) * 2;

  auto qk = [&](int st, f32x16 (&s)[2]) {
    u32x4 fr[2][8];
#pragma unroll
    for (int kd = 0; kd < 8; ++kd)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) fr[kh][kd] = *(const u32x4*)(smem + st * STAGE + 32 * kh * KS + k_off + 32 * kd);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kh][r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 8; ++kd)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) s[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fr[kh][kd]), mf(qf[kd]), s[kh], 0, 0, 0);
  };
  auto smax = [&](f32x16 (&s)[2]) {
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
    if (__any(m_new > m_run + 8.0f)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) oacc[mt] *= alpha;
      l_run *= alpha;
      m_run = m_new;
    }
  };
  auto sexp = [&](f32x16 (&s)[2], u32x4 (&pf)[2][2]) {
    const float mu = m_run;
    float rsum = 0.f;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kh][r], sc, -mu));
        s[kh][r] = p;
        rsum += p;
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        pf[kh][ks] = u32x4{pack2bf(s[kh][8 * ks + 0], s[kh][8 * ks + 1]), pack2bf(s[kh][8 * ks + 2], s[kh][8 * ks + 3]),
                           pack2bf(s[kh][8 * ks + 4], s[kh][8 * ks + 5]), pack2bf(s[kh][8 * ks + 6], s[kh][8 * ks + 7])};
    }
    l_run += rsum;
  };
  auto softmax = [&](f32x16 (&s)[2], u32x4 (&pf)[2][2]) { smax(s); sexp(s, pf); };
  auto pv = [&](int st, const u32x4 (&pf)[2][2]) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 fv[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const unsigned char* vp = smem + st * STAGE + v_off + (32 * kh + 16 * ks) * VS + 64 * mt;
          const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)vp);
          const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp + 8 * VS));
          u32x2 w0, w1;
          __builtin_memcpy(&w0, &a0, 8);
          __builtin_memcpy(&w1, &a1, 8);
          fv[mt] = u32x4{w0.x, w0.y, w1.x, w1.y};
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fv[mt]), mf(pf[kh][ks]), oacc[mt], 0, 0, 0);
      }
  };
  auto tile_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x16 sA[2], sB[2];
  u32x4 pA[2][2];
  int st = 0;
  if (MODE <= 1) {
    for (int t = 0; t < ntiles; ++t) {
      const int st2 = st == 0 ? 2 : st - 1;                 // (t + 2) % 3
      const int st1 = st == 2 ? 0 : st + 1;                 // (t + 1) % 3
      if (MODE != 0) dma_tile(st2);                         // tile t + 2 over tile t - 1 (its readers passed the barrier behind tile t - 1)
      qk(MODE == 0 ? 0 : st, sA);
      softmax(sA, pA);
      pv(MODE == 0 ? 0 : st, pA);
      if (MODE != 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // own pieces of tile t + 1 landed (tile t + 2 may fly)
      tile_barrier();
      st = st1;
    }
  } else if (MODE == 3 || wave < 4) {
    // DMA pieces issued between the row maximum and the exponentials (VALU-only stretch)
    for (int t = 0; t < ntiles; ++t) {
      const int st2 = st == 0 ? 2 : st - 1, st1 = st == 2 ? 0 : st + 1;
      qk(st, sA);
      smax(sA);
      dma_tile(st2);
      sexp(sA, pA);
      pv(st, pA);
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      tile_barrier();
      st = st1;
    }
  } else {
    // MODE 4, second half of the workgroup (the SIMD partners of waves 0-3): the same work ROTATED by one phase -- softmax(t),
    // PV(t), QK(t + 1) -- so that the partner's QK^T faces this wave's softmax and the partner's softmax this wave's PV
    qk(0, sA);
    for (int t = 0; t < ntiles; t += 2) {
      {
        const int st2 = st == 0 ? 2 : st - 1, st1 = st == 2 ? 0 : st + 1;
        smax(sA);
        dma_tile(st2);
        sexp(sA, pA);
        pv(st, pA);
        qk(st1, sB);                                        // tile t + 1: complete since the barrier behind tile t - 1
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        tile_barrier();
        st = st1;
      }
      {
        const int st2 = st == 0 ? 2 : st - 1, st1 = st == 2 ? 0 : st + 1;
        smax(sB);
        dma_tile(st2);
        sexp(sB, pA);
        pv(st, pA);
        if (t + 2 < ntiles) qk(st1, sA);
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        tile_barrier();
        st = st1;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    const float inv = 1.0f / l;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16_t* op = o + (size_t)(row0 + l31) * D + 32 * mt + 8 * qd + 4 * hi5;
        *(u32x2*)op = u32x2{pack2bf(oacc[mt][4 * qd] * inv, oacc[mt][4 * qd + 1] * inv),
                            pack2bf(oacc[mt][4 * qd + 2] * inv, oacc[mt][4 * qd + 3] * inv)};
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// 4-wave form: workgroup = 128 query rows = 4 waves x 32 rows x all 64 keys, ONE wave per SIMD and workgroup, TWO workgroups per
// CU (two 32 KB stages each): the two waves of a SIMD belong to different workgroups and drift apart freely -- the stagger of
// MODE 4 without a second code path, at the granularity (128 rows) that balances a causal call.  K / V images unpadded
// (64 x 256 B), XOR-swizzled on the SOURCE side of the DMA: K 16-byte piece p of row r at p ^ (r & 15), V 64-byte granule g of
// row r at g ^ (r & 3): 32 one-KB pieces per tile = 8 per wave, none wasted.
constexpr int STAGE4 = 2 * KT * 256;                     // 32 KB
template <int NSTG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn4_proto(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                                             const bf16_t* __restrict__ v, bf16_t* __restrict__ o, int ntiles, float sc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi5 = lane >> 5, l15 = lane & 15;
  const int row0 = blockIdx.x * 128 + wave * 32;
  const unsigned int lds_base = (unsigned int)(size_t)smem;
  // DMA: chunks c = wave + 4 j (rows 4c .. 4c + 3) of both images
  const int r_in = lane >> 4, pp = lane & 15;
  const unsigned int k_src = (unsigned int)(r_in * 256 + ((pp ^ (4 * wave + r_in)) << 4));
  const unsigned int v_src = (unsigned int)(r_in * 256 + (((((pp >> 2) ^ r_in) << 2) | (pp & 3)) << 4));
  auto dma_tile = [&](int st) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = wave + 4 * j;
#pragma unroll
      for (int isv = 0; isv < 2; ++isv) {
        const unsigned char* base = (const unsigned char*)(isv ? v : k) + (size_t)c * 1024;      // 4 rows x 256 B per chunk
        const unsigned int dst = lds_base + (unsigned int)st * STAGE4 + (isv ? 16384u : 0u) + 1024u * c;
        unsigned int keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(isv ? v_src : k_src), "s"(dst), "s"(base) : "memory");
      }
    }
  };
  u32x4 qf[8];
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) qf[kd] = *(const u32x4*)(q + (size_t)(row0 + l31) * D + 16 * kd + 8 * hi5);
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) asm volatile("" : "+v"(qf[kd]));
  f32x16 oacc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[mt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  dma_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // fragment addresses: K row 32 kh + l31, piece (2 kd + hi5) ^ l15  ==  base ^ (kd << 5);  V granule (mt ^ rr)  ==  base ^ (mt << 6)
  const unsigned int k_base = (unsigned int)(l31 * 256 + ((l15 >> 1) << 5) + ((hi5 ^ (l15 & 1)) << 4));
  const unsigned int v_base = (unsigned int)(16384 + (4 * hi5 + (l15 >> 2)) * 256 + ((l15 >> 2) << 6) + 32 * ((lane >> 4) & 1) + 8 * (l15 & 3));
  unsigned int ak[8], av[4];
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) ak[kd] = k_base ^ (unsigned int)(kd << 5);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) av[mt] = v_base ^ (unsigned int)(mt << 6);

  auto qk = [&](int st, f32x16 (&s)[2]) {
    u32x4 fr[2][8];
#pragma unroll
    for (int kd = 0; kd < 8; ++kd)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) fr[kh][kd] = *(const u32x4*)(smem + ak[kd] + st * STAGE4 + 32 * kh * 256);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kh][r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 8; ++kd)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) s[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fr[kh][kd]), mf(qf[kd]), s[kh], 0, 0, 0);
  };
  auto smax = [&](f32x16 (&s)[2]) {
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
    if (__any(m_new > m_run + 8.0f)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) oacc[mt] *= alpha;
      l_run *= alpha;
      m_run = m_new;
    }
  };
  auto sexp = [&](f32x16 (&s)[2], u32x4 (&pf)[2][2]) {
    const float mu = m_run;
    float rsum = 0.f;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kh][r], sc, -mu));
        s[kh][r] = p;
        rsum += p;
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        pf[kh][ks] = u32x4{pack2bf(s[kh][8 * ks + 0], s[kh][8 * ks + 1]), pack2bf(s[kh][8 * ks + 2], s[kh][8 * ks + 3]),
                           pack2bf(s[kh][8 * ks + 4], s[kh][8 * ks + 5]), pack2bf(s[kh][8 * ks + 6], s[kh][8 * ks + 7])};
    }
    l_run += rsum;
  };
  auto pv = [&](int st, const u32x4 (&pf)[2][2]) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 fv[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const unsigned char* vp = smem + av[mt] + st * STAGE4 + (32 * kh + 16 * ks) * 256;
          const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)vp);
          const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp + 8 * 256));
          u32x2 w0, w1;
          __builtin_memcpy(&w0, &a0, 8);
          __builtin_memcpy(&w1, &a1, 8);
          fv[mt] = u32x4{w0.x, w0.y, w1.x, w1.y};
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fv[mt]), mf(pf[kh][ks]), oacc[mt], 0, 0, 0);
      }
  };
  auto tile_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x16 sA[2];
  u32x4 pA[2][2];
  auto step = [&](auto st_tag, bool more) {
    constexpr int ST = decltype(st_tag)::value;
    qk(ST, sA);
    smax(sA);
    if (more) dma_tile(ST ^ 1);                            // tile t + 1 over tile t - 1 (its readers passed the barrier behind tile t - 1)
    sexp(sA, pA);
    pv(ST, pA);
    tile_barrier();                                        // own pieces of tile t + 1 landed; everybody is done with tile t
  };
  for (int t = 0; t < ntiles; t += 2) {
    step(std::integral_constant<int, 0>{}, t + 1 < ntiles);
    step(std::integral_constant<int, 1>{}, t + 2 < ntiles);
  }
  {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    const float inv = 1.0f / l;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16_t* op = o + (size_t)(row0 + l31) * D + 32 * mt + 8 * qd + 4 * hi5;
        *(u32x2*)op = u32x2{pack2bf(oacc[mt][4 * qd] * inv, oacc[mt][4 * qd + 1] * inv),
                            pack2bf(oacc[mt][4 * qd + 2] * inv, oacc[mt][4 * qd + 3] * inv)};
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// MODE 7: 8 waves on 128 rows -- 4 row groups x 2 key halves like the product kernel (balanced for causal calls: 128-row units),
// but WITHOUT loader waves (DMA from the compute waves, 4 pieces per wave and tile), ONE barrier per tile, and the key-half-1
// waves (the SIMD partners) rotated by one phase.  Unpadded swizzled images as in the 4-wave form; 3 stages of 32 KB.
template <int DUMMYARG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn8h_proto(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                                              const bf16_t* __restrict__ v, bf16_t* __restrict__ o, int ntiles, float sc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi5 = lane >> 5, l15 = lane & 15;
  const int rg = wave & 3, kh = wave >> 2;
  const int row0 = blockIdx.x * 128 + rg * 32;
  const unsigned int lds_base = (unsigned int)(size_t)smem;
  // DMA: wave w takes K chunks c = (w & 3) + 4 j for j = 2 (w >> 2) .. + 1, and the same V chunks
  const int r_in = lane >> 4, pp = lane & 15;
  const unsigned int k_src = (unsigned int)(r_in * 256 + ((pp ^ (4 * (wave & 3) + r_in)) << 4));
  const unsigned int v_src = (unsigned int)(r_in * 256 + (((((pp >> 2) ^ r_in) << 2) | (pp & 3)) << 4));
  auto dma_tile = [&](int st) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = (wave & 3) + 4 * (2 * (wave >> 2) + j);
#pragma unroll
      for (int isv = 0; isv < 2; ++isv) {
        const unsigned char* base = (const unsigned char*)(isv ? v : k) + (size_t)c * 1024;
        const unsigned int dst = lds_base + (unsigned int)st * STAGE4 + (isv ? 16384u : 0u) + 1024u * c;
        unsigned int keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(isv ? v_src : k_src), "s"(dst), "s"(base) : "memory");
      }
    }
  };
  u32x4 qf[8];
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) qf[kd] = *(const u32x4*)(q + (size_t)(row0 + l31) * D + 16 * kd + 8 * hi5);
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) asm volatile("" : "+v"(qf[kd]));
  f32x16 oacc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[mt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  dma_tile(0);
  dma_tile(1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  const unsigned int k_base = (unsigned int)((32 * kh + l31) * 256 + ((l15 >> 1) << 5) + ((hi5 ^ (l15 & 1)) << 4));
  const unsigned int v_base = (unsigned int)(16384 + (32 * kh + 4 * hi5 + (l15 >> 2)) * 256 + ((l15 >> 2) << 6) + 32 * ((lane >> 4) & 1) + 8 * (l15 & 3));
  unsigned int ak[8], av[4];
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) ak[kd] = k_base ^ (unsigned int)(kd << 5);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) av[mt] = v_base ^ (unsigned int)(mt << 6);
  auto qk = [&](int st, f32x16& s) {
    u32x4 fr[8];
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) fr[kd] = *(const u32x4*)(smem + ak[kd] + st * STAGE4);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fr[kd]), mf(qf[kd]), s, 0, 0, 0);
  };
  auto smax = [&](f32x16& s) {
    float rmax = vmax2(__builtin_fmaxf(s[0], s[1]), s[2]);
    rmax = vmax2(rmax, s[3]);
#pragma unroll
    for (int r = 4; r < 16; r += 2) rmax = vmax3(rmax, s[r], s[r + 1]);
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
    rmax = vmax2(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * sc;
    const float m_new = vmax2(m_run, rmax);
    if (__any(m_new > m_run + 8.0f)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) oacc[mt] *= alpha;
      l_run *= alpha;
      m_run = m_new;
    }
  };
  auto sexp = [&](f32x16& s, u32x4 (&pf)[2]) {
    const float mu = m_run;
    float rsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sc, -mu));
      s[r] = p;
      rsum += p;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      pf[ks] = u32x4{pack2bf(s[8 * ks + 0], s[8 * ks + 1]), pack2bf(s[8 * ks + 2], s[8 * ks + 3]),
                     pack2bf(s[8 * ks + 4], s[8 * ks + 5]), pack2bf(s[8 * ks + 6], s[8 * ks + 7])};
    l_run += rsum;
  };
  auto pv = [&](int st, const u32x4 (&pf)[2]) {
    u32x4 fv[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const unsigned char* vp = smem + av[mt] + st * STAGE4 + 16 * ks * 256;
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
      for (int mt = 0; mt < 4; ++mt) oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fv[ks][mt]), mf(pf[ks]), oacc[mt], 0, 0, 0);
  };
  auto tile_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x16 sA, sB;
  u32x4 pA[2];
  // stage of tile t = t % 3; three explicit bodies so that every LDS offset is an immediate
  if (kh == 0) {
    auto body = [&](auto st_tag) {
      constexpr int ST = decltype(st_tag)::value;
      qk(ST, sA);
      smax(sA);
      dma_tile((ST + 2) % 3);
      sexp(sA, pA);
      pv(ST, pA);
      tile_barrier();
    };
    for (int t = 0; t < ntiles; t += 3) {
      body(std::integral_constant<int, 0>{});
      if (t + 1 < ntiles) body(std::integral_constant<int, 1>{});
      if (t + 2 < ntiles) body(std::integral_constant<int, 2>{});
    }
  } else {
    qk(0, sA);
    auto body = [&](auto st_tag, f32x16& cur, f32x16& nxt) {
      constexpr int ST = decltype(st_tag)::value;
      smax(cur);
      dma_tile((ST + 2) % 3);
      sexp(cur, pA);
      pv(ST, pA);
      qk((ST + 1) % 3, nxt);
      tile_barrier();
    };
    for (int t = 0; t < ntiles; t += 6) {
      body(std::integral_constant<int, 0>{}, sA, sB);
      if (t + 1 < ntiles) body(std::integral_constant<int, 1>{}, sB, sA);
      if (t + 2 < ntiles) body(std::integral_constant<int, 2>{}, sA, sB);
      if (t + 3 < ntiles) body(std::integral_constant<int, 0>{}, sB, sA);
      if (t + 4 < ntiles) body(std::integral_constant<int, 1>{}, sA, sB);
      if (t + 5 < ntiles) body(std::integral_constant<int, 2>{}, sB, sA);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // merge of the key halves through LDS (stage 0 is free behind the last barrier... keep it simple: one more barrier)
  {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  __syncthreads();
  float* img = (float*)smem + (size_t)(rg * 32 + l31) * 132 + 4 * hi5;          // row stride 528 B
  float* ml = (float*)(smem + 128 * 528) + (rg * 32 + l31) * 2;
  if (kh == 1) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
#pragma unroll
        for (int e = 0; e < 4; ++e) img[32 * mt + 8 * qd + e] = oacc[mt][4 * qd + e];
    if (hi5 == 0) { ml[0] = m_run; ml[1] = l_run; }
  }
  __syncthreads();
  if (kh == 0) {
    const float m1 = ml[0], l1 = ml[1];
    const float m = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
    const float inv = 1.0f / (l_run * a0 + l1 * a1);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (oacc[mt][4 * qd + e] * a0 + img[32 * mt + 8 * qd + e] * a1) * inv;
        bf16_t* op = o + (size_t)(row0 + l31) * D + 32 * mt + 8 * qd + 4 * hi5;
        *(u32x2*)op = u32x2{pack2bf(x[0], x[1]), pack2bf(x[2], x[3])};
      }
  }
}
// ------------------------------------------------------------------------------------------------------------------------------
// MODE 8/9/10 (attn8r_proto<PLACE>): MODE 7 on REAL K / V streams (2 kv heads x ntiles x 64 keys, a different tile per iteration:
// the DMA comes from the L2s / HBM, not from an L1-resident tile), 4 stages (the rotated half reads K(t + 1) one barrier early),
// and the placement of the four DMA pieces of a wave as the template argument: 0 = all behind the row maximum, 1 = one piece
// behind every fourth MFMA of the two products, 2 = only the key-half-0 waves stage (8 pieces each, behind every second MFMA).
// MODE 7: 8 waves on 128 rows -- 4 row groups x 2 key halves like the product kernel (balanced for causal calls: 128-row units),
// but WITHOUT loader waves (DMA from the compute waves, 4 pieces per wave and tile), ONE barrier per tile, and the key-half-1
// waves (the SIMD partners) rotated by one phase.  Unpadded swizzled images as in the 4-wave form; 3 stages of 32 KB.
template <int PLACE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn8r_proto(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                                              const bf16_t* __restrict__ v, bf16_t* __restrict__ o, int ntiles, float sc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi5 = lane >> 5, l15 = lane & 15;
  const int rg = wave & 3, kh = wave >> 2;
  const int row0 = blockIdx.x * 128 + rg * 32;
  const unsigned int lds_base = (unsigned int)(size_t)smem;
  // DMA: wave w takes K chunks c = (w & 3) + 4 j for j = 2 (w >> 2) .. + 1, and the same V chunks
  const int r_in = lane >> 4, pp = lane & 15;
  const unsigned int k_src = (unsigned int)(r_in * 256 + ((pp ^ (4 * (wave & 3) + r_in)) << 4));
  const unsigned int v_src = (unsigned int)(r_in * 256 + (((((pp >> 2) ^ r_in) << 2) | (pp & 3)) << 4));
  const int kvh = ((int)blockIdx.x & 7) >> 2;                         // XCDs 0-3 read kv head 0, XCDs 4-7 kv head 1
  const size_t head_bytes = (size_t)ntiles * 64 * 256;
  // piece i (0..3 for a wave that stages 4, 0..7 for PLACE 2's key-half-0 waves) of tile `tile` into stage `st`
  auto dma_piece = [&](int tile, int st, int i) __attribute__((always_inline)) {
    const int isv = i & 1, j = i >> 1;
    const int c = PLACE == 2 ? (wave & 3) + 4 * j : (wave & 3) + 4 * (2 * (wave >> 2) + j);
    const unsigned char* base = (const unsigned char*)(isv ? v : k) + kvh * head_bytes + (size_t)tile * 16384 + (size_t)c * 1024;
    const unsigned int dst = lds_base + (unsigned int)st * STAGE4 + (isv ? 16384u : 0u) + 1024u * c;
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(isv ? v_src : k_src), "s"(dst), "s"(base) : "memory");
  };
  constexpr int NP = PLACE == 2 ? 8 : 4;                              // pieces of a staging wave per tile
  const bool stager = PLACE != 2 || kh == 0;
  auto dma_tile = [&](int tile, int st) __attribute__((always_inline)) {
    if (stager) {
#pragma unroll
      for (int i = 0; i < NP; ++i) dma_piece(tile, st, i);
    }
  };
  u32x4 qf[8];
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) qf[kd] = *(const u32x4*)(q + (size_t)(row0 + l31) * D + 16 * kd + 8 * hi5);
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) asm volatile("" : "+v"(qf[kd]));
  f32x16 oacc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[mt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  dma_tile(0, 0);
  dma_tile(1, 1);
  dma_tile(2, 2);
  if (PLACE == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  const unsigned int k_base = (unsigned int)((32 * kh + l31) * 256 + ((l15 >> 1) << 5) + ((hi5 ^ (l15 & 1)) << 4));
  const unsigned int v_base = (unsigned int)(16384 + (32 * kh + 4 * hi5 + (l15 >> 2)) * 256 + ((l15 >> 2) << 6) + 32 * ((lane >> 4) & 1) + 8 * (l15 & 3));
  unsigned int ak[8], av[4];
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) ak[kd] = k_base ^ (unsigned int)(kd << 5);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) av[mt] = v_base ^ (unsigned int)(mt << 6);
  // `hook(i)` is called behind MFMA i of a product (i = 0..7): where the spread placements put their DMA pieces
  auto qk = [&](int st, f32x16& s, auto hook) __attribute__((always_inline)) {
    u32x4 fr[8];
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) fr[kd] = *(const u32x4*)(smem + ak[kd] + st * STAGE4);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fr[kd]), mf(qf[kd]), s, 0, 0, 0);
      hook(kd);
    }
  };
  auto smax = [&](f32x16& s) {
    float rmax = vmax2(__builtin_fmaxf(s[0], s[1]), s[2]);
    rmax = vmax2(rmax, s[3]);
#pragma unroll
    for (int r = 4; r < 16; r += 2) rmax = vmax3(rmax, s[r], s[r + 1]);
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
    rmax = vmax2(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * sc;
    const float m_new = vmax2(m_run, rmax);
    if (__any(m_new > m_run + 8.0f)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) oacc[mt] *= alpha;
      l_run *= alpha;
      m_run = m_new;
    }
  };
  auto sexp = [&](f32x16& s, u32x4 (&pf)[2]) {
    const float mu = m_run;
    float rsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sc, -mu));
      s[r] = p;
      rsum += p;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      pf[ks] = u32x4{pack2bf(s[8 * ks + 0], s[8 * ks + 1]), pack2bf(s[8 * ks + 2], s[8 * ks + 3]),
                     pack2bf(s[8 * ks + 4], s[8 * ks + 5]), pack2bf(s[8 * ks + 6], s[8 * ks + 7])};
    l_run += rsum;
  };
  auto pv = [&](int st, const u32x4 (&pf)[2], auto hook) __attribute__((always_inline)) {
    u32x4 fv[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const unsigned char* vp = smem + av[mt] + st * STAGE4 + 16 * ks * 256;
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
      for (int mt = 0; mt < 4; ++mt) {
        oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fv[ks][mt]), mf(pf[ks]), oacc[mt], 0, 0, 0);
        hook(4 * ks + mt);
      }
  };
  auto tile_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    if (PLACE == 2 && kh == 0) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (PLACE == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x16 sA;
  u32x4 pA[2];
  auto nohook = [](int) {};
  // tile t in stage t & 3; tile t + 3 is requested during tile t (over the stage tile t - 1 left at the previous barrier) and has
  // landed -- every wave waits for its OWN pieces -- at the barrier that ends tile t + 1
  if (kh == 0) {
#pragma nounroll
    for (int t = 0; t < ntiles; ++t) {
      const int st = t & 3, st3 = (t + 3) & 3;
      const bool more = t + 3 < ntiles;
      // PLACE 1: pieces 0, 1 behind MFMAs 3, 7 of QK^T, pieces 2, 3 behind MFMAs 3, 7 of PV;  PLACE 2: pieces 0-3 behind MFMAs 1, 3, 5, 7 of QK^T, 4-7 of PV
      auto hq = [&](int i) { if (PLACE == 1 && more && (i & 3) == 3) dma_piece(t + 3, st3, i >> 2); if (PLACE == 2 && more && (i & 1)) dma_piece(t + 3, st3, i >> 1); };
      auto hp = [&](int i) { if (PLACE == 1 && more && (i & 3) == 3) dma_piece(t + 3, st3, 2 + (i >> 2)); if (PLACE == 2 && more && (i & 1)) dma_piece(t + 3, st3, 4 + (i >> 1)); };
      qk(st, sA, hq);
      smax(sA);
      if (PLACE == 0 && more) dma_tile(t + 3, st3);
      sexp(sA, pA);
      pv(st, pA, hp);
      tile_barrier();
    }
  } else {
    qk(0, sA, nohook);
#pragma nounroll
    for (int t = 0; t < ntiles; ++t) {
      const int st = t & 3, st1 = (t + 1) & 3, st3 = (t + 3) & 3;
      const bool more = t + 3 < ntiles;
      auto hp = [&](int i) { if (PLACE == 1 && more && (i & 3) == 3) dma_piece(t + 3, st3, i >> 2); };
      auto hq = [&](int i) { if (PLACE == 1 && more && (i & 3) == 3) dma_piece(t + 3, st3, 2 + (i >> 2)); };
      smax(sA);
      if (PLACE == 0 && more) dma_tile(t + 3, st3);
      sexp(sA, pA);
      pv(st, pA, hp);
      if (t + 1 < ntiles) qk(st1, sA, hq);
      else if (PLACE == 1 && more) { dma_piece(t + 3, st3, 2); dma_piece(t + 3, st3, 3); }
      tile_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // merge of the key halves through LDS (stage 0 is free behind the last barrier... keep it simple: one more barrier)
  {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  __syncthreads();
  float* img = (float*)smem + (size_t)(rg * 32 + l31) * 132 + 4 * hi5;          // row stride 528 B
  float* ml = (float*)(smem + 128 * 528) + (rg * 32 + l31) * 2;
  if (kh == 1) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
#pragma unroll
        for (int e = 0; e < 4; ++e) img[32 * mt + 8 * qd + e] = oacc[mt][4 * qd + e];
    if (hi5 == 0) { ml[0] = m_run; ml[1] = l_run; }
  }
  __syncthreads();
  if (kh == 0) {
    const float m1 = ml[0], l1 = ml[1];
    const float m = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
    const float inv = 1.0f / (l_run * a0 + l1 * a1);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (oacc[mt][4 * qd + e] * a0 + img[32 * mt + 8 * qd + e] * a1) * inv;
        bf16_t* op = o + (size_t)(row0 + l31) * D + 32 * mt + 8 * qd + 4 * hi5;
        *(u32x2*)op = u32x2{pack2bf(x[0], x[1]), pack2bf(x[2], x[3])};
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// MODE 11 / 12 (attn8p_proto<ACC2>): mode 8's data path (real K / V streams, 4 stages, no loader waves) with the compute of the full
// 256-VGPR design: a register block each for the K and the V^T fragments, every product's fragments requested ONE PHASE AHEAD
// (K(t + 1) before the PV(t) MFMAs, V(t) right behind the QK^T(t) MFMAs), every wave's 4 DMA pieces behind MFMAs 1, 3, 5, 7 of its PV
// product; ACC2: QK^T on two accumulators (two dependent chains of four).
template <int ACC2>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn8p_proto(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                                              const bf16_t* __restrict__ v, bf16_t* __restrict__ o, int ntiles, float sc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi5 = lane >> 5, l15 = lane & 15;
  const int rg = wave & 3, kh = wave >> 2;
  const int row0 = blockIdx.x * 128 + rg * 32;
  const unsigned int lds_base = (unsigned int)(size_t)smem;
  const int r_in = lane >> 4, pp = lane & 15;
  const unsigned int k_src = (unsigned int)(r_in * 256 + ((pp ^ (4 * (wave & 3) + r_in)) << 4));
  const unsigned int v_src = (unsigned int)(r_in * 256 + (((((pp >> 2) ^ r_in) << 2) | (pp & 3)) << 4));
  const int kvh = ((int)blockIdx.x & 7) >> 2;
  const size_t head_bytes = (size_t)ntiles * 64 * 256;
  auto dma_piece = [&](int tile, int st, int i) __attribute__((always_inline)) {
    const int isv = i & 1, j = i >> 1;
    const int c = (wave & 3) + 4 * (2 * (wave >> 2) + j);
    const unsigned char* base = (const unsigned char*)(isv ? v : k) + kvh * head_bytes + (size_t)tile * 16384 + (size_t)c * 1024;
    const unsigned int dst = lds_base + (unsigned int)st * STAGE4 + (isv ? 16384u : 0u) + 1024u * c;
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(isv ? v_src : k_src), "s"(dst), "s"(base) : "memory");
  };
  u32x4 qf[8];
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) qf[kd] = *(const u32x4*)(q + (size_t)(row0 + l31) * D + 16 * kd + 8 * hi5);
#pragma unroll
  for (int kd = 0; kd < 8; ++kd) asm volatile("" : "+v"(qf[kd]));
  f32x16 oacc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[mt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  for (int t0 = 0; t0 < 3; ++t0)
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(t0, t0, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned int k_base = (unsigned int)((32 * kh + l31) * 256 + ((l15 >> 1) << 5) + ((hi5 ^ (l15 & 1)) << 4));
  const unsigned int v_base = (unsigned int)(16384 + (32 * kh + 4 * hi5 + (l15 >> 2)) * 256 + ((l15 >> 2) << 6) + 32 * ((lane >> 4) & 1) + 8 * (l15 & 3));
  u32x4 fr[8], fv[8];
  auto load_k = [&](int st) __attribute__((always_inline)) {
    const unsigned int kb = k_base + (unsigned int)st * STAGE4;
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) fr[kd] = *(const u32x4*)(smem + (kb ^ (unsigned int)(kd << 5)));
  };
  auto load_v = [&](int st) __attribute__((always_inline)) {
    const unsigned int vb = v_base + (unsigned int)st * STAGE4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const unsigned char* vp = smem + (vb ^ (unsigned int)(mt << 6)) + 16 * ks * 256;
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)vp);
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp + 8 * 256));
        u32x2 w0, w1;
        __builtin_memcpy(&w0, &a0, 8);
        __builtin_memcpy(&w1, &a1, 8);
        fv[4 * ks + mt] = u32x4{w0.x, w0.y, w1.x, w1.y};
      }
  };
  auto qk_mfma = [&](f32x16& s) __attribute__((always_inline)) {
    if constexpr (ACC2 != 0) {
      f32x16 s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
      for (int kd = 0; kd < 8; kd += 2) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fr[kd]), mf(qf[kd]), s, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fr[kd + 1]), mf(qf[kd + 1]), s1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] += s1[r];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int kd = 0; kd < 8; ++kd) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fr[kd]), mf(qf[kd]), s, 0, 0, 0);
    }
  };
  auto smax = [&](f32x16& s) __attribute__((always_inline)) {
    float rmax = vmax2(__builtin_fmaxf(s[0], s[1]), s[2]);
    rmax = vmax2(rmax, s[3]);
#pragma unroll
    for (int r = 4; r < 16; r += 2) rmax = vmax3(rmax, s[r], s[r + 1]);
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
    rmax = vmax2(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * sc;
    const float m_new = vmax2(m_run, rmax);
    if (__any(m_new > m_run + 8.0f)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) oacc[mt] *= alpha;
      l_run *= alpha;
      m_run = m_new;
    }
  };
  auto sexp = [&](f32x16& s, u32x4 (&pf)[2]) __attribute__((always_inline)) {
    const float mu = m_run;
    float rsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sc, -mu));
      s[r] = p;
      rsum += p;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      pf[ks] = u32x4{pack2bf(s[8 * ks + 0], s[8 * ks + 1]), pack2bf(s[8 * ks + 2], s[8 * ks + 3]),
                     pack2bf(s[8 * ks + 4], s[8 * ks + 5]), pack2bf(s[8 * ks + 6], s[8 * ks + 7])};
    l_run += rsum;
  };
  auto pv_mfma = [&](const u32x4 (&pf)[2], int tile3, int st3, bool more) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf(fv[4 * ks + mt]), mf(pf[ks]), oacc[mt], 0, 0, 0);
        const int i = 4 * ks + mt;
        if (more && (i & 1)) dma_piece(tile3, st3, i >> 1);
      }
  };
  auto tile_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x16 sA;
  u32x4 pA[2];
  if (kh == 0) {
    load_k(0);
#pragma nounroll
    for (int t = 0; t < ntiles; ++t) {
      const int st = t & 3, st1 = (t + 1) & 3, st3 = (t + 3) & 3;
      qk_mfma(sA);
      load_v(st);
      smax(sA);
      sexp(sA, pA);
      if (t + 1 < ntiles) load_k(st1);
      pv_mfma(pA, t + 3, st3, t + 3 < ntiles);
      if (t + 3 >= ntiles) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_piece(t, st3, i);             // (keeps the vmcnt constant: re-fetch into a dead stage)
      }
      tile_barrier();
    }
  } else {
    load_k(0);
    qk_mfma(sA);
    load_v(0);
#pragma nounroll
    for (int t = 0; t < ntiles; ++t) {
      const int st1 = (t + 1) & 3, st3 = (t + 3) & 3;
      smax(sA);
      sexp(sA, pA);
      if (t + 1 < ntiles) load_k(st1);
      pv_mfma(pA, t + 3, st3, t + 3 < ntiles);
      if (t + 3 >= ntiles) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_piece(t, st3, i);
      }
      if (t + 1 < ntiles) {
        qk_mfma(sA);
        load_v(st1);
      }
      tile_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  __syncthreads();
  float* img = (float*)smem + (size_t)(rg * 32 + l31) * 132 + 4 * hi5;
  float* ml = (float*)(smem + 128 * 528) + (rg * 32 + l31) * 2;
  if (kh == 1) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
#pragma unroll
        for (int e = 0; e < 4; ++e) img[32 * mt + 8 * qd + e] = oacc[mt][4 * qd + e];
    if (hi5 == 0) { ml[0] = m_run; ml[1] = l_run; }
  }
  __syncthreads();
  if (kh == 0) {
    const float m1 = ml[0], l1 = ml[1];
    const float m = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
    const float inv = 1.0f / (l_run * a0 + l1 * a1);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (oacc[mt][4 * qd + e] * a0 + img[32 * mt + 8 * qd + e] * a1) * inv;
        bf16_t* op = o + (size_t)(row0 + l31) * D + 32 * mt + 8 * qd + 4 * hi5;
        *(u32x2*)op = u32x2{pack2bf(x[0], x[1]), pack2bf(x[2], x[3])};
      }
  }
}

extern "C" int attn8_proto_launch(const void* q, const void* k, const void* v, void* o, int rows, int ntiles, float sc, int mode, void* stream) {
  dim3 grid(rows / 256), block(512);
#define GO(M) { (void)hipFuncSetAttribute((const void*)attn8_proto<M>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
    hipLaunchKernelGGL((attn8_proto<M>), grid, block, LDS_BYTES, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ntiles, sc); }
  if (mode == 5 || mode == 6) {         // 4-wave form: 5 = 64 KB of LDS (two workgroups per CU), 6 = 96 KB (one per CU)
    const int lds = mode == 5 ? 2 * STAGE4 : 3 * STAGE4;
    (void)hipFuncSetAttribute((const void*)attn4_proto<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn4_proto<2>), dim3(rows / 128), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ntiles, sc);
    return (int)hipGetLastError();
  }
  if (mode == 11 || mode == 12) {
    const int lds = 4 * STAGE4;
    if (mode == 11) {
      (void)hipFuncSetAttribute((const void*)attn8p_proto<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((attn8p_proto<0>), dim3(rows / 128), dim3(512), lds, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ntiles, sc);
    } else {
      (void)hipFuncSetAttribute((const void*)attn8p_proto<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((attn8p_proto<1>), dim3(rows / 128), dim3(512), lds, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ntiles, sc);
    }
    return (int)hipGetLastError();
  }
  if (mode >= 8 && mode <= 10) {
    const int lds = 4 * STAGE4;
#define GO8(P) { (void)hipFuncSetAttribute((const void*)attn8r_proto<P>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    hipLaunchKernelGGL((attn8r_proto<P>), dim3(rows / 128), dim3(512), lds, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ntiles, sc); }
    if (mode == 8) GO8(0) else if (mode == 9) GO8(1) else GO8(2)
    return (int)hipGetLastError();
  }
  if (mode == 7) {
    const int lds = 3 * STAGE4;
    (void)hipFuncSetAttribute((const void*)attn8h_proto<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn8h_proto<0>), dim3(rows / 128), dim3(512), lds, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ntiles, sc);
    return (int)hipGetLastError();
  }
  if (mode == 0) GO(0) else if (mode == 1) GO(1) else if (mode == 3) GO(3) else GO(4)
  return (int)hipGetLastError();
}
