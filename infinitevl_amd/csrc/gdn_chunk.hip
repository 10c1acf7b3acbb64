// Gated DeltaNet, chunkwise form (chunk C = 64 tokens) for gfx950, K = 128, V = 256.
// Replaces fla's six-kernel pipeline (fla:ops/gated_delta_rule/chunk.py:18-71: l2norm, cumsum, WY transform
// wy_fast.py:114-238, state scan chunk_delta_h.py:32-124, output chunk_o.py:92-113) -- and, for the fused call, the three
// short convolutions and the gate glue in front of it -- with two launches, or ONE (gdn_chunk_single_kernel: both bodies in
// one grid, the scan workgroups waiting on in-kernel flags; small grids).
//
//  (1) pre-pass (gdn_chunk_prepare_body) -- chunk-PARALLEL, one 512-thread workgroup per (chunk, batch*head).  Everything
//      that depends on q, k, g, beta only (the "K side"; fused call: incl. conv + SiLU of q and k and the gate math):
//        q_hat, k_hat = l2norm -> bf16;  gamma = cumsum(g);  L = tril(bf16(beta k_hat) k_hat^T, -1);
//        Tw = (I+L)^-1 in fp32 (16x16 diagonal blocks by a Neumann product, the rest by block elimination, all on the
//        exact-fp32 MFMA v_mfma_f32_16x16x4_f32 with intermediates chained in registers);
//        Tu = Tw * e^{gamma_i-gamma_j};  w = bf16(Tw) bf16(beta k_hat);  A = tril((q_hat k_hat^T) * Gamma).
//      It leaves one 61 KB record per chunk: the four A-operand matrices of the serial pass, already decayed / negated and
//      stored as a sequence of 1 KB MFMA FRAGMENT BLOCKS (below), e^gamma, beta and Tu.  In the single-launch form a chunk's
//      pre-pass may be split over two workgroups (ROLE: k side | q side).
//  (2) scan (gdn_chunk_scan_body) -- SERIAL over chunks.  One STATE WAVE owns a 16-column slab of the state for all 128
//      rows: S[128 x 16] fp32 lives in 32 accumulator registers for the whole call and never visits LDS.  The
//      MFMA C layout (lane = column, registers = rows) IS the B-operand layout of the next product once the
//      contraction index is permuted inside each block of 32 (slot 8g+e <-> index 4g+e | 16+4g+(e-4)); the
//      records are written with that permutation, so per chunk
//          v_new  = u - (w e^gamma) S            (16 MFMA, u as C input, S as B operand straight from the accumulators)
//          S      = e^{gamma_L} S + (k_hat e^{gamma_L-gamma})^T v_new      (16 MFMA, v_new straight from accumulators)
//          o^T    = scale ((S^T q_hat^T) e^gamma + v_new^T A^T)            (22 MFMA: the OUTPUT WAVE of the pair)
//      with no LDS traffic on the S -> v_new -> S chain.  A workgroup = NCW state waves + NCW output waves (32 or 64 state
//      columns) + 4 LOADER waves that do nothing but stream the chunks' operand images into LDS by LDS-DMA
//      (global_load_lds_dwordx4, counted vmcnt) + 4 V WAVES that form u = bf16(Tu (beta v)) -- with the conv + SiLU of v in
//      the fused call -- one chunk ahead of the state waves (u does not depend on the state).  The image is split in two
//      halves by phase (H1: Wn, q_hat, e^gamma; H2: Kd^T, Aqk), each double buffered and refilled right after its last
//      reader's barrier: two barriers per chunk.
//
// Fragment block (v_mfma_f32_16x16x32_bf16): 16 rows x 32 contraction slots = 64 x 16 bytes, piece (g, i) at byte
// 16 (16 g + i) holds row i, slots 8g..8g+7  <->  contraction indices 4g+e (e < 4), 16+4g+(e-4) (e >= 4).  A wave
// reads a block lane-linearly (lane l = 16 g + i reads bytes [16 l, 16 l + 16)): conflict-free ds_read_b128, and
// the DMA that fills it is a plain linear copy of the record.
#include <mutex>

#include "ivl_common.h"
#include <type_traits>

namespace ivl {
IVL_TRACE_DECL(gdn)
// scan timeline slots: thread 0 of the FIRST scan workgroup (block (0, 0) of the scan launch; a later block of the single launch)
#ifdef IVL_TRACE
#define IVL_TOUT_WG(first, slot, value)                                             \
  do {                                                                              \
    long long* tb_ = ivl_trace_buf;                                                 \
    if (tb_ != nullptr && threadIdx.x == 0 && (first)) tb_[slot] = (long long)(value); \
  } while (0)
#else
#define IVL_TOUT_WG(first, slot, value) ((void)0)
#endif

typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int GC = 64;        // chunk length
constexpr int GK = 128;       // key head dim
constexpr int GV = 256;       // value head dim
constexpr int G_SEG_CHUNKS = 64;   // chunks per workspace segment (4096 tokens)

// ---- workspace record per (batch*head, chunk) ------------------------------------------------------------------
// 54 fragment blocks (1 KB each in bf16, 512 B in fp8 e4m3), then e^gamma / beta, then 6 bf16 blocks of Tu
//   WN  : -(bf16(w) e^gamma)            blocks (m, s)  = 4 m + s    rows: time   contraction: k
//   QH  : q_hat                         blocks (m, s)  = 4 m + s    rows: time   contraction: k
//   KDT : (k_hat e^{gl-gamma})^T        blocks (t, s2) = 2 t + s2   rows: k      contraction: time
//   AQK : tril((q k^T) Gamma)           blocks tri_blk(m, s2)       rows: time   contraction: time
//   EG  : f32 e^gamma[64]; EGL: f32 e^gamma_last; BETA: bf16 beta[64] (0 for the padded rows of a ragged last chunk)
//   TU  : bf16((I + L Gamma)^-1)        blocks tri_blk(m, s2)       rows: time   contraction: time   (always bf16: u = Tu (beta v)
//         is rounded to bf16 like the reference's, whatever the operand format of the scan's products)
// A lower-triangular 64 x 64 matrix has six non-zero 16 x 32 blocks: (m, s2) = (0,0) (1,0) (2,0) (2,1) (3,0) (3,1); the two
// blocks strictly above the diagonal are neither stored nor multiplied.
// Round 3: u itself is no longer part of the record.  The value side (v -> [conv + SiLU] -> beta v -> u = Tu (beta v)) runs in
// the scan's V waves, one chunk ahead of the state waves (u does not depend on the state): the pre-pass neither loads v nor
// writes the 32 KB of u per chunk, and the scan reads v where it read u.
template <bool F8>
struct Rec {
  static constexpr int BLK = F8 ? 512 : 1024;       // bytes per fragment block; a piece = BLK / 64 bytes per lane
  static constexpr int WN = 0, QH = 16 * BLK, KDT = 32 * BLK, AQK = 48 * BLK, EG = 54 * BLK, EGL = EG + 256, BETA = EG + 512;
  static constexpr int TU = EG + 1024;
  static constexpr size_t STRIDE = TU + 6 * 1024;   // bf16: 62,464   fp8: 34,816
};
__host__ __device__ constexpr int tri_blk(int m, int s2) { return m < 2 ? m : 2 + 2 * (m - 2) + s2; }

__device__ __forceinline__ mfma_bf16x8 mf(u32x4 v) {
  mfma_bf16x8 r;
  __builtin_memcpy(&r, &v, 16);
  return r;
}
__device__ __forceinline__ int crow32(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ __forceinline__ u32x4 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
  return u32x4{pack2bf(a0, a1), pack2bf(a2, a3), pack2bf(a4, a5), pack2bf(a6, a7)};
}

// four floats -> four OCP e4m3 bytes (gfx950 hardware conversion; clamped to the finite range +-448 first)
__device__ __forceinline__ unsigned int pack4_fp8(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f);
  b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f);
  d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (unsigned int)r;
}
// piece `idx` (lane-linear position inside a sequence of fragment blocks) <- eight consecutive contraction slots
// Record stores.  DEV (single-launch form: the record is read by other workgroups of the SAME launch, possibly through another
// die's L2): device-scope stores (sc1: written through this die's write-back L2 to the level every die sees) -- hipcc does not
// count them, the pre-pass drains them with an explicit s_waitcnt vmcnt(0) before it raises its flag.  A write-back of the L2
// (the release fence the memory model prescribes) walks the whole cache: ~8 us, more than the second launch it would save.
template <bool DEV> __device__ __forceinline__ void rec_st(void* p, u32x4 v) {
  // s_nop: a vector-memory store of more than 64 bits must not be followed at once by a VALU write of its data registers
  // (the compiler's hazard recognizer does not look inside an asm statement)
  if constexpr (DEV) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else *(u32x4*)p = v;
}
template <bool DEV> __device__ __forceinline__ void rec_st(void* p, u32x2 v) {
  if constexpr (DEV) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else *(u32x2*)p = v;
}
template <bool DEV> __device__ __forceinline__ void rec_st(void* p, float v) {
  if constexpr (DEV) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else *(float*)p = v;
}
template <bool DEV> __device__ __forceinline__ void rec_st(void* p, bf16_t v) {
  if constexpr (DEV) asm volatile("global_store_short %0, %1, off sc1" ::"v"(p), "v"((unsigned int)v) : "memory");
  else *(bf16_t*)p = v;
}
template <bool F8, bool DEV>
__device__ __forceinline__ void put_piece(unsigned char* base, int idx, float a0, float a1, float a2, float a3, float a4, float a5,
                                          float a6, float a7) {
  if (F8) rec_st<DEV>(base + idx * 8, u32x2{pack4_fp8(a0, a1, a2, a3), pack4_fp8(a4, a5, a6, a7)});
  else rec_st<DEV>(base + idx * 16, pack8(a0, a1, a2, a3, a4, a5, a6, a7));
}

// ---- single-launch forms: the sync area (protocol: see scan_wait_records below) -----------------------------------
constexpr int SYNC_HEAD_WORDS = 64;       // flag words per head: one per chunk of a workspace segment (G_SEG_CHUNKS)
enum { SYNC_E_START = 1, SYNC_E_GATE = 2, SYNC_E_KREAD = 3 };   // which wait ran out (error word, low byte)
struct ScanSync {
  unsigned int* flags = nullptr;            // flags[64 bh + c]: pre-pass workgroups of (bh, chunk c) that have published
  unsigned int* headdone = nullptr;
  unsigned int* kread = nullptr;            // kread[bh]: the q side of chunk 0 has read the old conv state of k (split pre-pass)
  unsigned int* err = nullptr;              // err[0]: sticky error code (0 = healthy); err[1]: (bh << 16) | chunk of the first failure
  unsigned int* host_err = nullptr;         // the same two words in pinned host memory (NULL: not available)
  int BH = 0;
  int nsplit = 0;                           // chunks [0, nsplit) of the segment come from a split pre-pass (a k side and a q side: two
                                            // increments of their flag), the others from one workgroup (one increment)
};
// one lane: record the first failure (device word for the workgroups of this and of later launches, host word for the C ABI)
__device__ __forceinline__ void sync_fail(const ScanSync& sy, unsigned int code, int bh, int c) {
  if (atomicCAS(sy.err, 0u, code) == 0u) {
    const unsigned int where = ((unsigned int)bh << 16) | (unsigned int)(c & 0xffff);
    __hip_atomic_store(sy.err + 1, where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sy.host_err != nullptr) {
      __hip_atomic_store(sy.host_err + 1, where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(sy.host_err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
constexpr int SYNC_SPIN_BOUND = 1 << 22;    // polls of ~0.2 us each (measured: 2^18 of them ~50 ms): ~1 s before a wait is declared failed

// ==================================================================================================
// (1) chunk-parallel pre-pass
// ==================================================================================================
constexpr int P_LDK = 136;                         // bf16 elements per row of k_hat / q_hat / beta k_hat (272 B)
constexpr int P_LDF = 68;                          // f32 elements per row of L / T
constexpr int P_KH = 0;                            // k_hat            [64][136] bf16
constexpr int P_QH = P_KH + GC * P_LDK * 2;        // q_hat            [64][136]
constexpr int P_KB = P_QH + GC * P_LDK * 2;        // bf16(beta k_hat) [64][136]
constexpr int P_L = P_KB + GC * P_LDK * 2;         // L, inverted IN PLACE to T = (I+L)^-1   [64][68] f32
constexpr int P_SM = P_L + GC * P_LDF * 4;         // gam[64], eg[64], dec[64], beta[64]
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

// Fused front end (FUSED = true): the pre-pass reads the mixer's ONE projection buffer itself and applies the three causal
// short convolutions (kernel 4, + SiLU, carry-in from / carry-out to the conv states, fla:modules/convolution.py via
// std:1253-1283) and the gate math (std:1293-1294) on the way in - the q / k / v / g / beta tensors of the unfused path never
// exist in HBM and the separate prologue launch disappears.  Same arithmetic, same bf16 rounding points as
// ivl_gdn_prologue_fwd followed by the plain pre-pass: the two paths agree bit for bit.
struct PrepFused {
  const bf16_t* proj; long long ld;                 // [B*T, ld] bf16
  int col_q, col_k, col_v, col_a, col_b;            // first column of q / k / v (head 0) and of the a / b gate inputs
  const bf16_t* w[3];                               // conv taps [D, 1, 4] bf16 of q, k, v
  const bf16_t* st_in[3];                           // conv states [B, D, 4] bf16 (NULL: zero history)
  bf16_t* st_out[3];                                // new conv states (may alias st_in; NULL: not wanted)
  const float* A_log; const float* dt_bias;         // [H] fp32
};

// causal 4-tap conv + SiLU of NR consecutive tokens x 8 channels: xr[0..2] = the three tokens before the run, xr[3..] the
// run; taps of channel c = (w[c / 2] .x/.y | .z/.w).  fp32 accumulation in tap order, output rounded to bf16 (the unfused
// path's tensors are bf16).
template <int NR>
__device__ __forceinline__ void conv4_silu(const u32x4* xr, const u32x4* w, u32x4* out) {
  // two channels per instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32; component-wise IEEE, so the results are those
  // of the scalar form): the front end of a 64-token chunk is ~9,000 VALU cycles per SIMD, half of them the two
  // transcendentals of the SiLU, which have no packed form
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const unsigned int ww[16] = {w[0].x, w[0].y, w[0].z, w[0].w, w[1].x, w[1].y, w[1].z, w[1].w,
                               w[2].x, w[2].y, w[2].z, w[2].w, w[3].x, w[3].y, w[3].z, w[3].w};
  f32x2 wf[4][4];                                   // [channel pair][tap]
#pragma unroll
  for (int c2 = 0; c2 < 4; ++c2) {
    const int c = 2 * c2;
    wf[c2][0] = f32x2{bflo(ww[2 * c]), bflo(ww[2 * c + 2])};
    wf[c2][1] = f32x2{bfhi(ww[2 * c]), bfhi(ww[2 * c + 2])};
    wf[c2][2] = f32x2{bflo(ww[2 * c + 1]), bflo(ww[2 * c + 3])};
    wf[c2][3] = f32x2{bfhi(ww[2 * c + 1]), bfhi(ww[2 * c + 3])};
  }
  auto unpack = [](const u32x4 x, f32x2* d) {
    d[0] = f32x2{bflo(x.x), bfhi(x.x)}; d[1] = f32x2{bflo(x.y), bfhi(x.y)};
    d[2] = f32x2{bflo(x.z), bfhi(x.z)}; d[3] = f32x2{bflo(x.w), bfhi(x.w)};
  };
  f32x2 win[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k) unpack(xr[k], win[k]);
  const f32x2 nl2e = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.f, 1.f};
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    f32x2 cur[4], o[4];
    unpack(xr[k + 3], cur);
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
      f32x2 a = wf[c2][0] * win[0][c2];
      a = __builtin_elementwise_fma(wf[c2][1], win[1][c2], a);
      a = __builtin_elementwise_fma(wf[c2][2], win[2][c2], a);
      a = __builtin_elementwise_fma(wf[c2][3], cur[c2], a);
      // a * sigmoid(a), sigmoid = rcp(1 + 2^(-a log2 e)) as in sigmoidf_ (ivl_common.h)
      f32x2 e = a * nl2e;
      e = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + one;
      o[c2] = a * f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
      win[0][c2] = win[1][c2]; win[1][c2] = win[2][c2]; win[2][c2] = cur[c2];
    }
    out[k] = u32x4{pack2bf(o[0][0], o[0][1]), pack2bf(o[1][0], o[1][1]), pack2bf(o[2][0], o[2][1]), pack2bf(o[3][0], o[3][1])};
  }
}

// The pre-pass of one (chunk ci, batch*head bh) by the first 512 threads of the workgroup.  `done` (single-launch form only):
// the word the scan workgroups of this head poll -- raised once the workgroup's part of the record is visible device-wide.
// ROLE (single-launch form, when the chip has room for twice the pre-pass workgroups): 0 = the whole pre-pass;
//   1 = the k side: conv + l2norm of k ONLY (all eight waves, runs of two tokens), gates, L, T = (I + L)^-1, w, Tu, Kd -- the
//       chain the scan waits for, without the q half of the front end and without Aqk / q_hat;
//   2 = the q side: conv + l2norm of q and k (as ROLE 0), q_hat, Aqk.
// (The gate step P1a stays behind wave 1's conv work: a ninth wave for it starts ~2,000 cycles after wave 0 -- the waves of a
// workgroup are launched one after the other -- and arrived at B1 later than wave 1 does with both jobs.)
// The conv state of k is read by BOTH workgroups of chunk 0 and written (possibly in place) by the k side: the q side raises
// `kread` once its loads of the old state have returned, the k side writes the new state at its very end, behind that word.
template <bool F8, bool FUSED, bool DEV, int ROLE, bool LOOPED = false>
__device__ __forceinline__ void gdn_chunk_prepare_body(
    unsigned char* smem, const int ci, const int bh,
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const float* __restrict__ g,
    const bf16_t* __restrict__ beta, const PrepFused& pf, unsigned char* __restrict__ ws, int T, int H, int t_seg0, int nt_seg, int l2norm,
    unsigned int* done, unsigned int* kread, const ScanSync* sy, const int tid_off = 0) {
  static_assert(ROLE == 0 || FUSED, "the split pre-pass exists for the fused front end only");
  constexpr bool KONLY = ROLE == 1, DO_K = ROLE != 2, DO_Q = ROLE != 1;
  using R = Rec<F8>;
  bf16_t* s_kh = (bf16_t*)(smem + P_KH);
  bf16_t* s_qh = (bf16_t*)(smem + P_QH);
  bf16_t* s_kb = (bf16_t*)(smem + P_KB);
  float* s_L = (float*)(smem + P_L);          // L, then T in place
  float* s_gam = (float*)(smem + P_SM);
  float* s_eg = s_gam + GC;
  float* s_dec = s_eg + GC;
  float* s_beta = s_dec + GC;

  IVL_T(tp0);
#ifdef IVL_TRACE
  const unsigned long long rt_p0 = __builtin_amdgcn_s_memrealtime();
#endif
  // (the thread index passes through an empty asm statement: inside the persistent loop of the long-call form nothing derived
  // from it is hoisted out of the body and kept alive across it -- the body stays at its straight-line register count)
  int tid_opaque = (int)threadIdx.x - tid_off;          // (tid_off: the long-call form runs two bodies per 1024-thread workgroup)
  if constexpr (LOOPED) asm volatile("" : "+v"(tid_opaque));
  const int tid = tid_opaque, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int l31 = lane & 31, hi = lane >> 5;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int b = bh / H, h = bh % H;            // ci = chunk within the segment
  const int t0 = t_seg0 + ci * GC;            // first token of the chunk
  const int nvalid = min(GC, T - t0);
  unsigned char* rec = ws + ((size_t)bh * nt_seg + ci) * R::STRIDE;

  // ---- P0: every global load of the chunk is issued up front (clamped rows, zeroed later) ------------------
  // plain : thread -> rows r0, r0 + 32 of q AND k (r0 = tid / 16), 16-byte column octet
  // fused : waves 0-3 -> rows 4 r0 .. 4 r0 + 3 of q, waves 4-7 -> the same rows of k (r0 < 16): runs of consecutive tokens, so
  //         that the three tokens in front of a run are loaded once; conv + SiLU of the run; the value side is not touched
  //         here at all (round 3: it belongs to the scan's V waves)
  const int oct = tid & 15;
  const bool is_k = KONLY || wave_u >= 4;            // fused: the array this wave converts (wave-uniform)
  // SIMD s hosts waves s and s + 4.  Three pieces of single-wave work sit in front of B1: the conv-state hand-over of q (the
  // threads that hold rows 0-3: wave 0), of k (wave 6: the k runs are rotated by 8) and the gate / cumsum step P1a (wave 1):
  // three different SIMDs.
  const int r0 = FUSED ? (KONLY ? (tid >> 4) & 31 : (is_k ? ((tid >> 4) + 8) & 15 : (tid >> 4) & 15)) : tid >> 4;
  constexpr int NQK = FUSED ? (KONLY ? 2 : 4) : 2;   // rows per thread (fused: of ONE array)
  u32x4 kraw[NQK], qraw[NQK];                        // fused: only the wave's own array is populated
  bf16_t braw[NQK];
  constexpr int P1A_WAVE = 1;
  u32x4 keep_tl[4], keep_h[3];                       // ROLE 1: the k conv-state hand-over is written at the end (see above)
  // the P1a wave also fetches the chunk's gate inputs here (consumed in P1a, behind its conv work: requested there they were
  // a memory round trip of their own on the wave every other one waits for at B1)
  float p1_g = 0.f, p1_b = 0.f, p1_dt = 0.f, p1_A = 0.f;
  if (wave_u == P1A_WAVE) {
    const size_t tok1 = ((size_t)b * T + t0 + min(lane, nvalid - 1)) * H + h;
    if constexpr (FUSED) {
      const bf16_t* row = pf.proj + ((size_t)b * T + t0 + min(lane, nvalid - 1)) * pf.ld;
      p1_g = bf2f(row[pf.col_a + h]);
      p1_b = bf2f(row[pf.col_b + h]);
      p1_dt = pf.dt_bias[h];
      p1_A = pf.A_log[h];
    } else {
      p1_g = g[tok1];
      p1_b = bf2f(beta[tok1]);
    }
  }
  // ---- P1a (one wave): g -> chunk-local inclusive cumsum -> e^gamma, decay to the chunk end; beta -------------
  // Fused front end: called BEHIND the issue of the wave's loads and IN FRONT of its conv work -- the gate inputs were requested
  // first and return first (in order), so the step runs while the wave's row loads are still in flight instead of making
  // wave 1 the last arrival at B1 by its whole length (~1,800 cycles).
  auto p1a = [&]() {
    float g_ld, b_ld;
    if constexpr (FUSED) {                           // g = -exp(A_log) softplus(a + dt_bias), beta = bf16(sigmoid(b)) (std:1293-1294)
      const float av = p1_g + p1_dt;
      const float sp = av > 20.f ? av : log1pf(expf(av));
      g_ld = -expf(p1_A) * sp;
      b_ld = bf2f(f2bf(sigmoid_exact_(p1_b)));
    } else {
      g_ld = p1_g;
      b_ld = p1_b;
    }
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
    s_beta[lane] = bv;                               // 0 for padded rows
    if constexpr (DO_K) {
      rec_st<DEV>((float*)(rec + R::EG) + lane, e);
      rec_st<DEV>((bf16_t*)(rec + R::BETA) + lane, f2bf(bv));     // the scan's V waves scale v with it (beta is a bf16 value: exact)
      if (lane == 0) rec_st<DEV>(rec + R::EGL, __expf(gl));
    }
  };
  auto qk_row = [&](int rr) { return FUSED ? NQK * r0 + rr : r0 + 32 * rr; };
  if constexpr (!FUSED) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = min(r0 + 32 * rr, nvalid - 1);
      const size_t tok = ((size_t)b * T + t0 + row) * H + h;
      kraw[rr] = *(const u32x4*)(k + tok * GK + 8 * oct);
      qraw[rr] = *(const u32x4*)(q + tok * GK + 8 * oct);
      braw[rr] = beta[tok];
    }
  } else {
    const bf16_t* xb = pf.proj + (size_t)b * T * pf.ld;              // row t at xb + t * ld (32-bit element offsets: host-checked)
    const unsigned int ld32 = (unsigned int)pf.ld;
    // rows of a run of NR tokens starting at chunk row `first`: global times t0 + first - 3 .. t0 + first + NR - 1,
    // clamped into the sequence (times < 0 come from the conv state, rows past the end are zeroed downstream)
    auto load_run = [&](u32x4* xr, int first, auto nr_tag, int col) {
      constexpr int NR = decltype(nr_tag)::value;
#pragma unroll
      for (int kk = 0; kk < NR + 3; ++kk) {
        int tg = t0 + first - 3 + kk;
        tg = tg < 0 ? 0 : (tg > T - 1 ? T - 1 : tg);
        xr[kk] = *(const u32x4*)(xb + ((unsigned int)tg * ld32 + (unsigned int)col));
      }
    };
    // the history of a run that starts at time 0 is the conv state: state[c][1..3] = times -3, -2, -1
    // (the state is REQUESTED by history_load together with the run's own loads and only consumed by history: requested
    // where it is consumed it costs the workgroup that holds time 0 - the one every launch waits for - a second full
    // memory round trip, ~3,400 cycles)
    auto history_load = [&](u32x4* s4, const bf16_t* st_in) {
      s4[0] = s4[1] = s4[2] = s4[3] = u32x4{0u, 0u, 0u, 0u};
      if (st_in != nullptr) {
        const u32x4* sp = (const u32x4*)(st_in + ((size_t)b * H * GK + (size_t)h * GK + 8 * oct) * 4);
        s4[0] = sp[0]; s4[1] = sp[1]; s4[2] = sp[2]; s4[3] = sp[3];
      }
    };
    auto history = [&](u32x4* xr, const u32x4* s4) {
      const unsigned int ss[16] = {s4[0].x, s4[0].y, s4[0].z, s4[0].w, s4[1].x, s4[1].y, s4[1].z, s4[1].w,
                                   s4[2].x, s4[2].y, s4[2].z, s4[2].w, s4[3].x, s4[3].y, s4[3].z, s4[3].w};
      // channel c: ss[2c] = (state[c][0], state[c][1]), ss[2c + 1] = (state[c][2], state[c][3]); row kk takes tap kk + 1 of the
      // eight channels: one v_perm_b32 per pair of channels (the hand-over runs on a few lanes of ONE wave while the whole
      // workgroup waits at B1, and shares its SIMD with a wave that is converting: every instruction costs ~10 cycles)
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int w = (kk + 1) >> 1;
        const unsigned int sel = ((kk + 1) & 1) ? 0x07060302u : 0x05040100u;         // the high | the low halves of (a, b)
        unsigned int d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = __builtin_amdgcn_perm(ss[2 * (2 * j + 1) + w], ss[2 * (2 * j) + w], sel);
        xr[kk] = u32x4{d[0], d[1], d[2], d[3]};
      }
    };
    // new conv state = the last four inputs of the sequence ([old state, x] when T < 4): written by the thread that read
    // the old one (chunk 0, run 0), so that st_out may alias st_in.  tail_load issues the four row loads together with the
    // run's own loads (four dependent round trips otherwise, on the workgroup every other one waits for).
    auto tail_load = [&](u32x4* tail, int col) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int tg = T - 4 + j;                                            // ext[T + j] with ext = [state(4), x(T)]
        tg = tg < 0 ? 0 : tg;
        tail[j] = *(const u32x4*)(xb + ((unsigned int)tg * ld32 + (unsigned int)col));
      }
    };
    auto put_state = [&](const u32x4* tail, const u32x4* hist3, bf16_t* st_out) {
      if (st_out == nullptr) return;
      const int D = H * GK, d0 = h * GK + 8 * oct;
      u32x4 row[4];
      if (T >= 4) {                                                    // (uniform) the last four inputs, as loaded
#pragma unroll
        for (int j = 0; j < 4; ++j) row[j] = tail[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = T + j;                                         // e >= 4: x[e - 4]; e = 1..3 (T < 4): old state[.][e]
          row[j] = e >= 4 ? tail[j] : (e == 1 ? hist3[0] : (e == 2 ? hist3[1] : hist3[2]));
        }
      }
      // row[j] holds (channel 0..7) of ext[T + j]; the state is [channel][4 taps]: transpose 4 x 8 halfwords
      const unsigned int r0w[4] = {row[0].x, row[0].y, row[0].z, row[0].w}, r1w[4] = {row[1].x, row[1].y, row[1].z, row[1].w};
      const unsigned int r2w[4] = {row[2].x, row[2].y, row[2].z, row[2].w}, r3w[4] = {row[3].x, row[3].y, row[3].z, row[3].w};
      u32x4* op = (u32x4*)(st_out + ((size_t)b * D + d0) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {                                    // channels 2i, 2i + 1
        const unsigned int lo01 = __builtin_amdgcn_perm(r1w[i], r0w[i], 0x05040100u), lo23 = __builtin_amdgcn_perm(r3w[i], r2w[i], 0x05040100u);
        const unsigned int hi01 = __builtin_amdgcn_perm(r1w[i], r0w[i], 0x07060302u), hi23 = __builtin_amdgcn_perm(r3w[i], r2w[i], 0x07060302u);
        op[i] = u32x4{lo01, lo23, hi01, hi23};
      }
    };
    const bool own_state = t0 == 0;                                    // this workgroup holds time 0
    {
      const int a = is_k ? 1 : 0;                                      // 0: q, 1: k (wave-uniform)
      u32x4 xr[NQK + 3], wt[4];
      const int col = (is_k ? pf.col_k : pf.col_q) + h * GK + 8 * oct;
      load_run(xr, NQK * r0, std::integral_constant<int, NQK>{}, col);
      const u32x4* wp = (const u32x4*)(pf.w[a] + ((size_t)h * GK + 8 * oct) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) wt[i] = wp[i];
      bf16_t bin[NQK] = {};
      if (is_k) {
#pragma unroll
        for (int rr = 0; rr < NQK; ++rr) {
          const int tg = min(t0 + NQK * r0 + rr, T - 1);
          bin[rr] = xb[(unsigned int)tg * ld32 + (unsigned int)(pf.col_b + h)];
        }
      }
      // every load of the thread is in flight before the first one is consumed (in-order return: waiting for any of them
      // waits for all issued before it)
      u32x4 tl[4], st4[4];
      // (k side: runs of two tokens -- the run that starts at time 2 has time -1, the state's last tap, in front of it too)
      const bool second_run = KONLY && own_state && r0 == 1;
      if (own_state && r0 == 0) {
        tail_load(tl, col);
        history_load(st4, pf.st_in[a]);
      } else if (second_run) {
        history_load(st4, pf.st_in[a]);
      }
      if (wave_u == P1A_WAVE) p1a();                                   // (consumes the wave's OLDEST loads only)
      if (is_k) {
#pragma unroll
        for (int rr = 0; rr < NQK; ++rr) braw[rr] = f2bf(sigmoid_exact_(bf2f(bin[rr])));     // beta = bf16(sigmoid(b)) (std:1293)
      }
      IVL_T(tf0);
      IVL_TOUT(8, tf0 - tp0);
      if (own_state && r0 == 0) {
        history(xr, st4);
        if (ROLE == 0 || (ROLE == 2 && !is_k)) {
          put_state(tl, xr, pf.st_out[a]);
        } else if (ROLE == 1) {                                        // written at the end, behind the q side's `kread`
#pragma unroll
          for (int j = 0; j < 4; ++j) keep_tl[j] = tl[j];
#pragma unroll
          for (int j = 0; j < 3; ++j) keep_h[j] = xr[j];
        } else if (pf.st_out[1] != nullptr) {                          // ROLE 2, k waves: the old state has been read: history
          // consumed it.  The explicit wait keeps the store behind the RETURN of the loads whatever the compiler schedules (a
          // relaxed atomic is not ordered against ordinary loads); it costs nothing, st4 was the wave's youngest load
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (oct == 0) __hip_atomic_store(kread, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (second_run) {
        u32x4 hx[3];
        history(hx, st4);
        xr[0] = hx[2];                                                 // time -1
      }
      IVL_T(tf1);
      if (is_k) conv4_silu<NQK>(xr, wt, kraw);
      else conv4_silu<NQK>(xr, wt, qraw);
      IVL_T(tf2);
      IVL_TOUT(9, tf1 - tf0); IVL_TOUT(10, tf2 - tf1);
    }
  }
  if constexpr (!FUSED) {
    if (wave_u == P1A_WAVE) p1a();
  }
  // ---- P1b: l2norm -> k_hat, q_hat (bf16);  bf16(beta k_hat) -------------------------------------------------
  auto norm_row = [&](u32x4 xv, int row, bool ok, float* f) {        // f[0..7] = x / |x| (fp32), 0 for a padded row
    const unsigned int xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) { f[2 * c] = bflo(xw[c]); f[2 * c + 1] = bfhi(xw[c]); }
    float rs = 1.f;
    if (l2norm) {
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) ss = fmaf(f[c], f[c], ss);
      ss = row16_sum(ss);
      rs = __builtin_amdgcn_rsqf(ss + 1e-6f);
    }
    rs = ok ? rs : 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] *= rs;
  };
  auto put_k = [&](const float* f, int row, float bt) {
    float kf[8], kb[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { kf[c] = bf_round(f[c]); kb[c] = kf[c] * bt; }
    *(u32x4*)(s_kh + row * P_LDK + 8 * oct) = pack8(kf[0], kf[1], kf[2], kf[3], kf[4], kf[5], kf[6], kf[7]);
    *(u32x4*)(s_kb + row * P_LDK + 8 * oct) = pack8(kb[0], kb[1], kb[2], kb[3], kb[4], kb[5], kb[6], kb[7]);
  };
#pragma unroll
  for (int rr = 0; rr < NQK; ++rr) {
    const int row = qk_row(rr);
    const bool ok = row < nvalid;
    float f[8];
    if (!FUSED || is_k) {
      norm_row(kraw[rr], row, ok, f);
      put_k(f, row, ok ? bf2f(braw[rr]) : 0.f);
    }
    if (!FUSED || !is_k) {
      norm_row(qraw[rr], row, ok, f);
      *(u32x4*)(s_qh + row * P_LDK + 8 * oct) = pack8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
    }
  }
  IVL_T(tb1);
  IVL_TOUT(15, tb1 - tp0); IVL_TOUT_AT(448, 29, tb1 - tp0); IVL_TOUT_AT(64, 30, tb1 - tp0);
  __syncthreads();                                   // B1
  IVL_T(tp1);

  // Copy-out of the two operands that are plain re-orderings of the LDS tiles, piece `idx` of 2048:
  //   [0,1024)    QH : piece (block 4m+s, g, i) = q_hat[16m+i][32s + {4g..4g+3, 16+4g..+3}]
  //   [1024,2048) KDT: piece (block 2t+s2, g, i) = k_hat[32s2 + {4g.., 16+4g..}][16t+i] * dec[time]  (LDS transpose read)
  auto copy_piece = [&](int idx) {
    const int i = idx & 15, gg = (idx >> 4) & 3;
    if (idx < 1024) {
      const int blk = idx >> 6, m = blk >> 2, s = blk & 3;
      const bf16_t* src = s_qh + (16 * m + i) * P_LDK + 32 * s + 4 * gg;
      const u32x2 lo = *(const u32x2*)src, hi2 = *(const u32x2*)(src + 16);
      put_piece<F8, DEV>(rec + R::QH, idx, bflo(lo.x), bfhi(lo.x), bflo(lo.y), bfhi(lo.y), bflo(hi2.x), bfhi(hi2.x), bflo(hi2.y), bfhi(hi2.y));
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
      put_piece<F8, DEV>(rec + R::KDT, id2, bflo(w0.x) * d0[0], bfhi(w0.x) * d0[1], bflo(w0.y) * d0[2], bfhi(w0.y) * d0[3],
                    bflo(w1.x) * d1[0], bfhi(w1.x) * d1[1], bflo(w1.y) * d1[2], bfhi(w1.y) * d1[3]);
    }
  };

  // ---- P2: waves 0-2: L = tril(kb kh^T, -1) -> s_L;  waves 3-5: A^T = kh qh^T -> Aqk blocks;  waves 6-7: first copy-outs -----
  if (DO_K && wave_u < 3) {
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
  } else if (DO_Q && wave_u >= 3 && wave_u < 6) {
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
    const int pc0 = tri_blk(2 * mi + (l31 >> 4), nj) * 64 + hi * 16 + (l31 & 15);        // piece (block, g = hi, i)
#pragma unroll
    for (int p = 0; p < 2; ++p)
      put_piece<F8, DEV>(rec + R::AQK, pc0 + 32 * p, val[4 * p], val[4 * p + 1], val[4 * p + 2], val[4 * p + 3], val[8 + 4 * p], val[9 + 4 * p],
                    val[10 + 4 * p], val[11 + 4 * p]);
  } else if (ROLE == 0) {
    const int t2 = tid - 384;                        // 0..127
#pragma unroll
    for (int z = 0; z < 4; ++z) copy_piece(t2 + 128 * z);                 // QH pieces 0..511
  } else {
    // split pre-pass: the five waves without a product copy the side's re-ordered operand out (1024 pieces over 320 threads)
    const int t2 = ROLE == 1 ? tid - 192 : (tid < 192 ? tid : tid - 192);   // k side: waves 3-7; q side: waves 0-2, 6-7
#pragma unroll
    for (int z = 0; z < 4; ++z) {
      const int idx = t2 + 320 * z;
      if (idx < 1024) copy_piece((ROLE == 1 ? 1024 : 0) + idx);
    }
  }
  if constexpr (ROLE == 2) {                          // the q side is complete: q_hat and Aqk are on their way
    if (done != nullptr) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) atomicAdd(done, 1u);
    }
#ifdef IVL_TRACE
    if (ivl_trace_buf != nullptr && tid == 0) {       // q side: last start (47), last publish (45)
      atomicMax((unsigned long long*)ivl_trace_buf + 45, (unsigned long long)__builtin_amdgcn_s_memrealtime());
      atomicMax((unsigned long long*)ivl_trace_buf + 47, rt_p0);
    }
#endif
    return;
  }
  __syncthreads();                                   // B2: s_L complete
  IVL_T(tp2);

  // ---- P3: T = (I + L)^-1 in place (waves 0-3); waves 4-7 finish the copy-outs meanwhile -----------------------
  // level 0: the 16x16 diagonal block `wave`, D = I + N with N strictly lower (N^16 = 0):
  //          D^-1 = (I - N)(I + N^2)(I + N^4)(I + N^8), eight 16x16x16 products on v_mfma_f32_16x16x4_f32 (fp32 operands).
  //          A product takes its left factor in the A layout (lane (g, j): M[j][4g + s]) and its right factor in the B layout
  //          (M[4g + s][j]), which is also the layout of a result; the A layout of M is the result layout of M^T, so the
  //          squarings carry both M (N2C, N4C) and M^T (N2A, N4A, N8A: "M in the A layout") along.  (A row-by-row forward
  //          substitution on 16 lanes was ~2,500 cycles of dependent FMAs; this is ~600.)
  if (wave_u < 4) {
    const int bb = 16 * wave_u;
    const f32x4 NA = *(const f32x4*)(s_L + (bb + l15) * P_LDF + bb + 4 * g4);
    f32x4 NB;
#pragma unroll
    for (int s = 0; s < 4; ++s) NB[s] = s_L[(bb + 4 * g4 + s) * P_LDF + bb + l15];
    auto mm = [](const f32x4 A, const f32x4 B, f32x4 C) {
#pragma unroll
      for (int s = 0; s < 4; ++s) C = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s], B[s], C, 0, 0, 0);
      return C;
    };
    const f32x4 Z0 = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 N2C = mm(NA, NB, Z0), N2A = mm(NB, NA, Z0);
    const f32x4 N4C = mm(N2A, N2C, Z0), N4A = mm(N2C, N2A, Z0);
    const f32x4 N8A = mm(N4C, N4A, Z0);
    f32x4 Rr;
#pragma unroll
    for (int s = 0; s < 4; ++s) Rr[s] = (4 * g4 + s == l15 ? 1.f : 0.f) - NB[s];      // I - N
    Rr = mm(N2A, Rr, Rr);
    Rr = mm(N4A, Rr, Rr);
    Rr = mm(N8A, Rr, Rr);
#pragma unroll
    for (int r = 0; r < 4; ++r) s_L[(bb + 4 * g4 + r) * P_LDF + bb + l15] = Rr[r];
  } else if (ROLE == 0) {
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

  // ---- P4a: Tu = Tw * e^{gamma_i - gamma_j} -> bf16, straight into its fragment blocks of the record (8 elements of one
  //           row per thread = two 8-byte half pieces; the blocks strictly above the diagonal do not exist) ----------------
  {
    const int row = tid >> 3, c0 = 8 * (tid & 7);
    const int m = row >> 4, i = row & 15, s2 = c0 >> 5, tt = c0 & 31;
    if (s2 == 0 || m >= 2) {
      const float gi = s_gam[row];
      float tv[8];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x4 tq = *(const f32x4*)(s_L + row * P_LDF + c0 + 4 * hh);
        const f32x4 gj = *(const f32x4*)(s_gam + c0 + 4 * hh);
#pragma unroll
        for (int c = 0; c < 4; ++c) tv[4 * hh + c] = row >= c0 + 4 * hh + c ? tq[c] * __expf(gi - gj[c]) : 0.f;
      }
      // times tt..tt+3 -> lane group g = (tt & 15) / 4, times tt+4..tt+7 -> g + 1; both in slots 0-3 (tt < 16) or 4-7 of a piece
      const int gq = (tt & 15) >> 2, half = tt >> 4;
      unsigned char* blk = rec + R::TU + tri_blk(m, s2) * 1024 + half * 8;
      rec_st<DEV>(blk + (16 * gq + i) * 16, u32x2{pack2bf(tv[0], tv[1]), pack2bf(tv[2], tv[3])});
      rec_st<DEV>(blk + (16 * (gq + 1) + i) * 16, u32x2{pack2bf(tv[4], tv[5]), pack2bf(tv[6], tv[7])});
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
    const int pc0 = ((2 * mi + (l31 >> 4)) * 4 + s) * 64 + hi * 16 + (l31 & 15);
#pragma unroll
    for (int p = 0; p < 2; ++p)
      put_piece<F8, DEV>(rec + R::WN, pc0 + 32 * p, bf_round(acc[4 * p]) * negeg, bf_round(acc[4 * p + 1]) * negeg, bf_round(acc[4 * p + 2]) * negeg,
                    bf_round(acc[4 * p + 3]) * negeg, bf_round(acc[8 + 4 * p]) * negeg, bf_round(acc[9 + 4 * p]) * negeg,
                    bf_round(acc[10 + 4 * p]) * negeg, bf_round(acc[11 + 4 * p]) * negeg);
  }
  IVL_T(tp4);
#ifdef IVL_TRACE
  // spread over the workgroups of the launch on the shared 100 MHz clock: first start (slot 26), last end (27), longest (28)
  if (ivl_trace_buf != nullptr && threadIdx.x == 0) {
    const unsigned long long rt_p1 = __builtin_amdgcn_s_memrealtime();
    atomicMin((unsigned long long*)ivl_trace_buf + 26, rt_p0);
    atomicMax((unsigned long long*)ivl_trace_buf + 27, rt_p1);
    atomicMax((unsigned long long*)ivl_trace_buf + 28, rt_p1 - rt_p0);
  }
#endif
  IVL_TOUT(0, tp0); IVL_TOUT(1, tp1 - tp0); IVL_TOUT(2, tp2 - tp1); IVL_TOUT(3, tp3a - tp2); IVL_TOUT(4, tp3b - tp3a);
  IVL_TOUT(5, tp3 - tp3b); IVL_TOUT(6, tp4 - tp3); IVL_TOUT(7, tp4);
  IVL_T(tq1);
  if (done != nullptr) {
    // publish: every thread's (device-scope) record stores have been acknowledged, then one thread raises the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IVL_T(tq2);
    __syncthreads();
    IVL_T(tq3);
    if (tid == 0) atomicAdd(done, 1u);
    IVL_TOUT(49, tq1 - tp4); IVL_TOUT(50, tq2 - tq1); IVL_TOUT(51, tq3 - tq2);
  }
  if constexpr (ROLE == 1) {
    // the new conv state of k (chunk 0, the sixteen threads that read the old one): the q side's workgroup has read the old
    // state too once `kread` is up (bounded wait, as scan_wait_records) -- cleared again for the next launch.  Behind the
    // publish: the scan does not read the conv state, and the device-scope load of the word is a full memory round trip
    if (t0 == 0 && tid < 16 && pf.st_out[1] != nullptr) {
      bool read = false;
      for (int spin = 0; spin < SYNC_SPIN_BOUND; ++spin) {
        if (__hip_atomic_load(kread, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { read = true; break; }
        if ((spin & 31) == 31 && __hip_atomic_load(sy->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (!read) {                                                     // (uniform over the 16 threads) the old state may still be
        if (tid == 0) sync_fail(*sy, SYNC_E_KREAD, bh, 0);             // unread: it is NOT overwritten, the failure is reported
        return;
      }
      if (tid == 0) __hip_atomic_store(kread, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (put_state is defined in the front-end scope: the same transposition, inlined here)
      const int D = H * GK, d0 = h * GK + 8 * oct;
      u32x4 row[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = T + j;
        row[j] = (T >= 4 || e >= 4) ? keep_tl[j] : (e == 1 ? keep_h[0] : (e == 2 ? keep_h[1] : keep_h[2]));
      }
      const unsigned int r0w[4] = {row[0].x, row[0].y, row[0].z, row[0].w}, r1w[4] = {row[1].x, row[1].y, row[1].z, row[1].w};
      const unsigned int r2w[4] = {row[2].x, row[2].y, row[2].z, row[2].w}, r3w[4] = {row[3].x, row[3].y, row[3].z, row[3].w};
      u32x4* op = (u32x4*)(pf.st_out[1] + ((size_t)b * D + d0) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned int lo01 = __builtin_amdgcn_perm(r1w[i], r0w[i], 0x05040100u), lo23 = __builtin_amdgcn_perm(r3w[i], r2w[i], 0x05040100u);
        const unsigned int hi01 = __builtin_amdgcn_perm(r1w[i], r0w[i], 0x07060302u), hi23 = __builtin_amdgcn_perm(r3w[i], r2w[i], 0x07060302u);
        op[i] = u32x4{lo01, lo23, hi01, hi23};
      }
    }
  }
#ifdef IVL_TRACE
  if (ivl_trace_buf != nullptr && tid == 0) {         // last publish (46), last start (48)
    atomicMax((unsigned long long*)ivl_trace_buf + 46, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    atomicMax((unsigned long long*)ivl_trace_buf + 48, rt_p0);
  }
#endif
}

template <bool F8, bool FUSED>
__global__ __launch_bounds__(512, 2) void gdn_chunk_prepare_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const float* __restrict__ g,
    const bf16_t* __restrict__ beta, PrepFused pf, unsigned char* __restrict__ ws, int T, int H, int t_seg0, int nt_seg, int l2norm) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  gdn_chunk_prepare_body<F8, FUSED, false, 0>(smem, (int)blockIdx.x, (int)blockIdx.y, q, k, g, beta, pf, ws, T, H, t_seg0, nt_seg, l2norm, nullptr,
                                              nullptr, nullptr);
}

// ==================================================================================================
// (2) serial scan + output
// ==================================================================================================
// LDS: two operand images, bytes [0, Rec::EG + 1024) of the record each (Wn, q_hat, Kd^T, Aqk, e^gamma / beta).  H1 = {Wn,
// q_hat, e^gamma} is read in the first half of a chunk, H2 = {Kd^T, Aqk} in the second.  bf16: 55 KB per image; fp8: 28 KB.
// Behind the images: the workgroup's slab of u (2 KB per pair, written by the V waves, one buffer), per pair the sb / v_new
// exchange (6 fragment blocks), then the V waves' staging area (beta v as B-operand fragment blocks, 2 KB per pair).
template <bool F8>
struct Img {
  static constexpr int BYTES = Rec<F8>::EG + 1024;                                         // one operand image
  static constexpr int XCH_PAIR = 6 * Rec<F8>::BLK;
  __host__ __device__ static constexpr int uslab(int ncw) { return 2 * BYTES; }              // u: 2 KB per pair, ONE buffer
  __host__ __device__ static constexpr int xch(int ncw) { return uslab(ncw) + ncw * 2048; } // sb (4 fragments) | v_new (2 fragments) per pair
  __host__ __device__ static constexpr int stg(int ncw) { return xch(ncw) + ncw * XCH_PAIR; } // beta v: 2 KB per pair
  __host__ __device__ static constexpr int dummy(int ncw) { return stg(ncw) + ncw * 2048; }   // 256 B: landing zone of the touches
  // NCW = 2 only (the LDS of the 64-column workgroup is full): the raw value rows of a chunk, t0 - 3 .. t0 + 63 (68 row slots of
  // 64 bytes), two buffers -- fetched by the loaders' LDS-DMA instead of 28 small vector loads per chunk by the V waves
  static constexpr int VT_ROWS = 80, VT_BYTES = VT_ROWS * 64;     // five 1 KB DMA instructions per tile
  __host__ __device__ static constexpr int vt(int ncw) { return dummy(ncw) + 256; }
  // ... and Tu (six 1 KB blocks) and beta (64 bf16) of a chunk, two buffers each: the V waves of the 32-column workgroup issue
  // no vector-memory instruction at all (each costs its wave ~100 cycles beside the DMA stream)
  static constexpr int TU_BYTES = 6 * 1024, BETA_BYTES = 256;
  __host__ __device__ static constexpr int tub(int ncw) { return vt(ncw) + 2 * VT_BYTES; }
  __host__ __device__ static constexpr int betab(int ncw) { return tub(ncw) + 2 * TU_BYTES; }
  __host__ __device__ static constexpr int total(int ncw) { return ncw == 2 ? betab(ncw) + 2 * BETA_BYTES : vt(ncw); }
};
__host__ __device__ constexpr int scan_lds_bytes(int ncw, bool f8) { return f8 ? Img<true>::total(ncw) : Img<false>::total(ncw); }
static_assert(scan_lds_bytes(4, false) <= 160 * 1024, "scan LDS budget");

// LDS-DMA, NP consecutive 1 KB pieces: global [gsrc + 1024 p + 16 lane] -> LDS [lds_dst + 1024 p + 16 lane], p = 0..NP-1
// (the instruction offset is added to both addresses).  gsrc and lds_dst are wave-uniform (SGPRs); hipcc does not count
// these operations: completion is awaited with explicit counted s_waitcnt vmcnt and published by the following barrier.
// DEV (single-launch forms): the source is a record that ANOTHER workgroup of the same launch has written (device-scope
// `sc1` stores, flag raised behind their acknowledgement).  The loads carry `sc1` as well: they are served by the L2, never by
// a line this CU's vector L1 may hold -- the consumer half of the hand-off (MI355X_MICROARCH.md, inter-workgroup visibility:
// "sc1 loads may replace the acquire only when the producer stored sc1"; cdna_hip_programming.md Guideline 16, R1).
#define IVL_DMA_BODY(NP_, SC_)                                                                                   \
  if constexpr (NP_ == 4)                                                                                        \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"                                          \
                 "global_load_lds_dwordx4 %1, %3" SC_ "\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024" SC_ "\n\t"  \
                 "global_load_lds_dwordx4 %1, %3 offset:2048" SC_ "\n\tglobal_load_lds_dwordx4 %1, %3 offset:3072" SC_ "\n\t" \
                 "s_mov_b32 m0, %0"                                                                              \
                 : "=&s"(keep) : "v"(lane16), "s"(lds_dst), "s"(gsrc) : "memory");                              \
  else if constexpr (NP_ == 3)                                                                                   \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"                                          \
                 "global_load_lds_dwordx4 %1, %3" SC_ "\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024" SC_ "\n\t"  \
                 "global_load_lds_dwordx4 %1, %3 offset:2048" SC_ "\n\t"                                          \
                 "s_mov_b32 m0, %0"                                                                              \
                 : "=&s"(keep) : "v"(lane16), "s"(lds_dst), "s"(gsrc) : "memory");                              \
  else if constexpr (NP_ == 2)                                                                                   \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"                                          \
                 "global_load_lds_dwordx4 %1, %3" SC_ "\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024" SC_ "\n\t"  \
                 "s_mov_b32 m0, %0"                                                                              \
                 : "=&s"(keep) : "v"(lane16), "s"(lds_dst), "s"(gsrc) : "memory");                              \
  else                                                                                                           \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3" SC_ "\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep) : "v"(lane16), "s"(lds_dst), "s"(gsrc) : "memory")
template <int NP, bool DEV = false>
__device__ __forceinline__ void dma_pieces(const unsigned char* gsrc, unsigned int lds_dst, unsigned int lane16) {
  static_assert(NP >= 1 && NP <= 4, "pieces per issue");
  unsigned int keep;
  if constexpr (DEV) { IVL_DMA_BODY(NP, " sc1"); }
  else { IVL_DMA_BODY(NP, ""); }
}
// workgroup barrier that waits for this wave's LDS traffic only (no vector-memory drain)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The SCAN_NL loader waves take the DMA units (4 consecutive 1 KB pieces; the last unit of a region may be shorter) of a
// half image round-robin (unit u -> loader u % SCAN_NL).  A DMA instruction costs its wave ~50-100 cycles of issue time: with
// two loaders the loaders arrived last at the M barrier (state waves waited ~400 cycles per chunk there, ~240 with four).
//   H1(ci): Wn | q_hat (contiguous: 32 pieces in bf16, 16 in fp8) + 1 piece e^gamma / beta (loader 0)
//   H2(ci): Kd^T | Aqk (contiguous: 22 pieces in bf16, 11 in fp8)
__host__ __device__ constexpr int h1_pieces(bool f8) { return f8 ? 16 : 32; }
__host__ __device__ constexpr int h2_pieces(bool f8) { return f8 ? 11 : 22; }
constexpr int SCAN_NL = 4;
__host__ __device__ constexpr int unit_size(int u, int n) { return n - 4 * u >= 4 ? 4 : (n - 4 * u > 0 ? n - 4 * u : 0); }
__host__ __device__ constexpr int loader_pieces(int L, int n) {               // pieces of an n-piece region issued by loader L
  int c = 0;
  for (int u = L; 4 * u < n; u += SCAN_NL) c += unit_size(u, n);
  return c;
}
__host__ __device__ constexpr int loader_n1(int L, bool f8) { return loader_pieces(L, h1_pieces(f8)) + (L == 0 ? 1 : 0); }
__host__ __device__ constexpr int loader_n2(int L, bool f8) { return loader_pieces(L, h2_pieces(f8)); }

template <int L, int NPIECES, bool DEV, int U = 0>
__device__ __forceinline__ void load_region(const unsigned char* gsrc, unsigned int lds_dst, unsigned int lane16) {
  if constexpr (4 * U < NPIECES) {
    if constexpr (U % SCAN_NL == L) dma_pieces<unit_size(U, NPIECES), DEV>(gsrc + U * 4096, lds_dst + (unsigned int)(U * 4096), lane16);
    load_region<L, NPIECES, DEV, U + 1>(gsrc, lds_dst, lane16);
  }
}
template <int L, bool F8, bool DEV>
__device__ __forceinline__ void load_h1(const unsigned char* rec, unsigned int img, unsigned int lane16) {
  using R = Rec<F8>;
  load_region<L, h1_pieces(F8), DEV>(rec + R::WN, img + (unsigned int)R::WN, lane16);
  if (L == 0) dma_pieces<1, DEV>(rec + R::EG, img + (unsigned int)R::EG, lane16);
}
template <int L, bool F8, bool DEV>
__device__ __forceinline__ void load_h2(const unsigned char* rec, unsigned int img, unsigned int lane16) {
  using R = Rec<F8>;
  load_region<L, h2_pieces(F8), DEV>(rec + R::KDT, img + (unsigned int)R::KDT, lane16);
}

// Barrier protocol (every wave of the workgroup executes the same sequence [PF,] PA, P0, P, T(0), M(0), T(1), M(1), ..., F):
//   PF    : single-launch forms only: loader 0 -- the workgroup's one polling wave -- has seen the flags of the records the loaders
//           request first (or its wait failed: abort word 0, every wave leaves)
//   PA    : the raw value tile of chunk 0 has landed (NCW = 2)
//   P0    : beta v of chunk 0 is staged (V waves)
//   P     : H1(0) has landed;  u(0) is in image 0
//   T(ci) : H2(ci) has landed;  every wave has finished chunk ci - 1      -> H2(ci + 1) may be issued (image (ci+1) & 1)
//           beta v of chunk ci + 1 is staged
//   M(ci) : H1(ci + 1) has landed;  every wave has read H1(ci)            -> H1(ci + 2) may be issued (image ci & 1)
//           u(ci + 1) is in image (ci + 1) & 1;  the staging area is free
// so each half image is requested one whole chunk before its barrier.  vmcnt retires in issue order: "landed" = at most
// the pieces issued AFTER the awaited half are still outstanding.
// L2 warm-up for the V waves ("touch"): a V wave reads its value rows, beta and Tu with ordinary loads ONE chunk ahead (two
// register sets; a deeper register pipeline does not fit), which hides an L2 hit but not a trip to HBM / the other dies'
// L2s.  Loader 2 therefore requests one dword of every 128-byte line of the workgroup's value rows of chunk c, loader 3 of
// Tu(c) and beta(c), five chunks ahead, as LDS-DMA into a 256-byte dummy area: two instructions per chunk and workgroup.
struct ScanTouch {
  const unsigned char* vrow0;                          // first byte of the workgroup's columns in row 0 of this batch's values
  unsigned int row_bytes;                              // bytes between consecutive tokens
  int T, t_seg0;
  unsigned int dummy;                                  // LDS byte address of the dummy area
  unsigned int vt, tub, betab;                         // LDS byte addresses of the value-tile / Tu / beta buffers (NCW = 2)
};
// WHAT = 2: the value rows, 3: Tu and beta, anything else: nothing
template <int L, bool F8>
__device__ __forceinline__ void touch_chunk(const ScanTouch& tc, const unsigned char* ws_bh, int c, int nt_seg, int lane) {
  using R = Rec<F8>;
  if constexpr (L != 2 && L != 3) return;
  c = c < nt_seg ? c : nt_seg - 1;                                 // always issued (static wait counts): re-touches the last chunk
  unsigned int keep;

  if constexpr (L == 2) {
    int t = tc.t_seg0 + c * GC + lane;
    t = t > tc.T - 1 ? tc.T - 1 : t;
    const unsigned int off = (unsigned int)(t - tc.t_seg0) * tc.row_bytes;
    const unsigned char* base = tc.vrow0 + (size_t)tc.t_seg0 * tc.row_bytes;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(tc.dummy), "s"(base) : "memory");
  } else {
    const unsigned char* base = ws_bh + (size_t)c * R::STRIDE + R::EG;
    const unsigned int off = lane < 48 ? 1024u + 128u * (unsigned int)lane : 512u + 128u * (unsigned int)(lane & 1);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(tc.dummy), "s"(base) : "memory");
  }
}
// NCW = 2: the workgroup's raw value rows of chunk c (rows t0 - 3 .. t0 + 63, clamped into the batch; 64 bytes = four 16-byte
// pieces per row) into tile buffer c & 1: piece i = 4 row + piece-of-row lands at byte 16 i.  Loader 2 issues pieces 0..191,
// loader 3 the remaining 76 (+ padding that repeats the last row).
template <int L>
__host__ __device__ constexpr int vt_instrs() { return L == 2 ? 3 : (L == 3 ? 2 : 0); }
template <int L, bool F8>
__device__ __forceinline__ void load_vt(const ScanTouch& tc, int c, int nt_seg, int lane) {
  if constexpr (vt_instrs<L>() == 0) return;
  c = c < nt_seg ? c : nt_seg - 1;                                 // always issued (static wait counts)
  const int tfirst = tc.t_seg0 + c * GC - 3;
  const int tbase = tfirst < 0 ? 0 : tfirst;
  const unsigned char* base = tc.vrow0 + (size_t)tbase * tc.row_bytes;
  constexpr int K0 = L == 2 ? 0 : 3, K1 = L == 2 ? 3 : 5;
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const int i = 64 * k + lane;
    int row = i >> 2;
    row = row > 66 ? 66 : row;
    int t = tfirst + row;
    t = t < 0 ? 0 : (t > tc.T - 1 ? tc.T - 1 : t);
    const unsigned int off = (unsigned int)(t - tbase) * tc.row_bytes + 16u * (unsigned int)(i & 3);
    const unsigned int dst = tc.vt + (unsigned int)((c & 1) * Img<F8>::VT_BYTES + 1024 * k);
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(dst), "s"(base) : "memory");
  }
}

// NCW = 2: Tu(c) (six 1 KB pieces: loader 1 takes three, loader 2 two, loader 3 one) and beta(c) (one dword instruction,
// loader 3) into buffer c & 1
template <int L>
__host__ __device__ constexpr int tu_instrs() { return L == 1 ? 3 : (L == 2 ? 2 : (L == 3 ? 1 : 0)); }
template <int L, bool F8, bool DEV>
__device__ __forceinline__ void load_tu(const ScanTouch& tc, const unsigned char* ws_bh, int c, int nt_seg, unsigned int lane16) {
  if constexpr (tu_instrs<L>() > 0) {
    c = c < nt_seg ? c : nt_seg - 1;
    constexpr int P0_ = L == 1 ? 0 : (L == 2 ? 3 : 5);
    const unsigned char* src = ws_bh + (size_t)c * Rec<F8>::STRIDE + Rec<F8>::TU + P0_ * 1024;
    const unsigned int dst = tc.tub + (unsigned int)((c & 1) * Img<F8>::TU_BYTES + P0_ * 1024);
    dma_pieces<tu_instrs<L>(), DEV>(src, dst, lane16);
  }
}
template <int L, bool F8, bool DEV>
__device__ __forceinline__ void load_beta(const ScanTouch& tc, const unsigned char* ws_bh, int c, int nt_seg, int lane) {
  if constexpr (L != 3) return;
  c = c < nt_seg ? c : nt_seg - 1;
  const unsigned char* src = ws_bh + (size_t)c * Rec<F8>::STRIDE + Rec<F8>::BETA;
  const unsigned int dst = tc.betab + (unsigned int)((c & 1) * Img<F8>::BETA_BYTES);
  const unsigned int off = 4u * (unsigned int)lane;
  unsigned int keep;
  if constexpr (DEV)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(dst), "s"(src) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(dst), "s"(src) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Single-launch forms: the scan workgroups start together with the pre-pass workgroups of the same launch, fetch what does
// not depend on the records (value tiles, the initial state), and their loader waves wait here for the flags of the head's
// records.  flags[64 bh + c] is raised by the pre-pass workgroup(s) of (chunk c, head bh) and cleared again by the LAST of the
// head's scan workgroups to have passed its waits (counted in headdone[bh], which it also clears): the area is all-zero between
// launches -- the caller zeroes it once.
//
// What the forms rest on, and what they do NOT rest on:
//  * LIVENESS: a workgroup only ever waits for workgroups with LOWER block ids (pre-pass ids < scan ids; the q side of chunk 0,
//    whose `kread` the k side awaits, has the lowest ids of all; the long-call pre-pass is a set of PERSISTENT workgroups in
//    front of the scan workgroups).  The hardware dispatches a grid in id order, so every awaited workgroup has been
//    dispatched -- and, never waiting itself for anything younger, finishes -- whatever share of the chip the launch gets (a
//    second stream, a second process, a partitioned device).  HIP does not promise the order (MI355X_MICROARCH.md, Workgroup
//    dispatch): it is relied on for progress only, never for results.
//  * SAFETY does not rest on it: every wait is bounded, and a wait that runs out (or that finds the area already failed) raises
//    a STICKY error word in the sync area and in a host-visible status word, and the workgroup stops: it stores no output
//    and no state computed from records it has not seen.  ivl_gdn_chunk_fused_fwd refuses to launch on a failed area
//    (IVL_ERR_SYNC), ivl_gdn_sync_status reports it, ivl_gdn_sync_reset re-arms the area.
//  * VISIBILITY (cdna_hip_programming.md Guideline 16, R1): records are stored write-through (`sc1`), every storing wave drains
//    (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane raises the flag (agent-scope atomic); the consumer polls relaxed with
//    agent-scope loads and reads the records with `sc1` loads (LDS-DMA), which the vector L1 cannot serve.
// the loader waves' wait at the start: every record of the call (small grids: the pre-pass workgroups all finish together), or,
// for long calls (`progressive`), the records of chunks 0..2 (what the loaders request in front of the first chunk step) -- the
// later chunks are then awaited by V wave 0, three chunks ahead of the state waves (scan_vwave).
// Lane c watches chunk c, lane 63 the area's error word.  false: the wait ran out or the area had failed before (the caller
// then loads nothing from the records and tells its workgroup through the abort words in LDS).
template <bool PROGRESSIVE>
__device__ __forceinline__ bool scan_wait_records(const ScanSync& sy, int bh, int nt_seg, int lane) {
  const int nw = PROGRESSIVE ? (nt_seg < 3 ? nt_seg : 3) : nt_seg;         // (all at once: the host keeps nt_seg <= 63)
  const int cw = lane < nw ? lane : 0;                                      // the chunk this lane watches
  const unsigned int want = lane == 63 ? 0u : (cw < sy.nsplit ? 2u : 1u);
  const unsigned int* p = lane == 63 ? sy.err : sy.flags + bh * SYNC_HEAD_WORDS + cw;
  // (Three polls in flight, a third of a round trip apart, were measured: the flags are seen earlier, but the extra device-scope
  // loads slow the pre-pass workgroups they wait for -- 16.3 -> 16.6 us at the step shape.  ONE wave per workgroup polls.)
  for (int spin = 0; spin < SYNC_SPIN_BOUND; ++spin) {
    const unsigned int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long bad = __builtin_amdgcn_ballot_w64(v != want);
    if (bad == 0ull) return true;
    if (bad >> 63) {                                                       // the area has failed before: no use waiting
      // ... and the host hears of it again: should ANOTHER area ever share this area's status slot (status_slot: only when all 64
      // slots hold failed areas), its reset has cleared the slot while this area is still failed
      if (lane == 63 && sy.host_err != nullptr) __hip_atomic_store(sy.host_err, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
    __builtin_amdgcn_s_sleep(4);
  }
  if (lane == 0) sync_fail(sy, SYNC_E_START, bh, 0);
  return false;
}
// Abort words of a scan workgroup (LDS, the first dwords of the otherwise unused `dummy` area of the 32-column workgroup):
// word 0 = the wait at the start failed (written by loader 0 in front of the barrier PF, 0 or 1; words 1-3 zero), word 4 = the
// gate wave's (V wave 0, long calls: 0 before PF, 1 when a later chunk's wait runs out).  Read by every wave behind PF, by the
// output waves at every chunk step and by the state waves in front of their state store: nothing computed from unseen records
// leaves the workgroup.
template <bool F8> __device__ __forceinline__ unsigned int* scan_abort_words(unsigned char* smem) {
  return (unsigned int*)(smem + Img<F8>::dummy(2));
}
template <bool F8, bool WITH_GATE> __device__ __forceinline__ bool scan_aborted(unsigned char* smem) {
  const u32x4 a = *(const u32x4*)scan_abort_words<F8>(smem);
  unsigned int any = a.x | a.y | a.z | a.w;
  if constexpr (WITH_GATE) any |= scan_abort_words<F8>(smem)[4];
  return __builtin_amdgcn_readfirstlane(any) != 0u;
}

// SYNC: 0 = records from an earlier launch; 1 = single launch, all records awaited at the start; 2 = long calls (progressive)
template <int L, int NCW, bool F8, int SYNC>
__device__ __forceinline__ void scan_loader(const unsigned char* ws_bh, int nt_seg, unsigned char* smem, unsigned int lds0, unsigned int lane16,
                                            const ScanTouch& tc, int lane, const ScanSync& sy, int bh, bool trace_wg) {
  (void)trace_wg;
  constexpr int N1 = loader_n1(L, F8), N2 = loader_n2(L, F8);               // pieces per half image issued by this loader
  // NCW = 4: the V waves load their rows / Tu / beta themselves, loaders 2, 3 warm the L2 for them (one touch per chunk)
  // NCW = 2: the loaders fetch the value tile, beta (with H2: the "T batch") and Tu (with H1: the "M batch") for real
  constexpr bool VD = NCW == 2;
  constexpr bool DEV = SYNC != 0;                                           // the records come from workgroups of this launch: sc1 loads
  constexpr int TW = VD ? 0 : (L == 2 || L == 3 ? L : 0);                   // what this loader touches (2: value rows, 3: Tu / beta)
  constexpr int NT = TW != 0 ? 1 : 0;                                       // touch instructions per chunk
  constexpr int NVB = VD ? vt_instrs<L>() + (L == 3 ? 1 : 0) : 0;           // value tile + beta instructions per chunk
  constexpr int NTU = VD ? tu_instrs<L>() : 0;
  constexpr int NTB = N2 + NVB, NMB = N1 + NTU + NT;                        // T batch, M batch (full)
  auto rec = [&](int ci) { return ws_bh + (size_t)ci * Rec<F8>::STRIDE; };
  auto img = [&](int ci) { return lds0 + (unsigned int)((ci & 1) * Img<F8>::BYTES); };
  auto vset = [&](int c) {                                                   // value tile, beta (and Tu at the start) of chunk c
    if constexpr (VD) { load_vt<L, F8>(tc, c, nt_seg, lane); load_beta<L, F8, DEV>(tc, ws_bh, c, nt_seg, lane); }
  };
  // issue order at the start: chunk 0's value tile, beta, Tu (the V waves' conversion + product of chunk 0 stand between their
  // arrival and the first chunk step) | H1(0) | chunk 1's set | H2(0) | H1(1) | touches
  if constexpr (SYNC != 0) {
    // single launch: the value tiles of chunks 0 and 1 do not come from the pre-pass -- requested before the wait
    static_assert(VD, "the single-launch form is built on the 32-column workgroup");
    constexpr int NB = L == 3 ? 1 : 0;
    load_vt<L, F8>(tc, 0, nt_seg, lane);
    load_vt<L, F8>(tc, 1, nt_seg, lane);
    // ONE wave of the workgroup (loader 0) polls -- 512 polling waves on the scan's CUs slowed the pre-pass they wait for --
    // and the barrier PF carries what it saw to every other wave
    if constexpr (L == 0) {
      const bool ok = scan_wait_records<SYNC == 2>(sy, bh, nt_seg, lane);
      if (lane == 0) *(u32x4*)scan_abort_words<F8>(smem) = u32x4{ok ? 0u : 1u, 0u, 0u, 0u};
    }
    lds_barrier();                                               // PF: the records the loaders request first are published (or the wait failed)
    if (scan_aborted<F8, false>(smem)) {                         // (workgroup-uniform) nothing is read from the records: every wave stops here
      wait_vm<0>();
      return;
    }
#ifdef IVL_TRACE
    if (ivl_trace_buf != nullptr && lane == 0 && trace_wg && L == 0) ivl_trace_buf[42] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    load_beta<L, F8, DEV>(tc, ws_bh, 0, nt_seg, lane);
    load_tu<L, F8, DEV>(tc, ws_bh, 0, nt_seg, lane16);
    load_h1<L, F8, DEV>(rec(0), img(0), lane16);
    load_beta<L, F8, DEV>(tc, ws_bh, 1, nt_seg, lane);
    load_tu<L, F8, DEV>(tc, ws_bh, 1, nt_seg, lane16);
    load_h2<L, F8, DEV>(rec(0), img(0), lane16);
    if (nt_seg > 1) load_h1<L, F8, DEV>(rec(1), img(1), lane16);
    if (nt_seg > 1) wait_vm<N1 + NB + NTU + N2 + N1>();          // chunk 0's beta / Tu have landed (its value tile long before)
    else wait_vm<N1 + NB + NTU + N2>();
  } else {
    vset(0);
    if constexpr (VD) load_tu<L, F8, DEV>(tc, ws_bh, 0, nt_seg, lane16);
    load_h1<L, F8, DEV>(rec(0), img(0), lane16);
    vset(1);
    if constexpr (VD) load_tu<L, F8, DEV>(tc, ws_bh, 1, nt_seg, lane16);
    load_h2<L, F8, DEV>(rec(0), img(0), lane16);
    if (nt_seg > 1) load_h1<L, F8, DEV>(rec(1), img(1), lane16);
    touch_chunk<TW, F8>(tc, ws_bh, 3, nt_seg, lane);
    touch_chunk<TW, F8>(tc, ws_bh, 4, nt_seg, lane);
    if (nt_seg > 1) wait_vm<N1 + NVB + NTU + N2 + N1 + 2 * NT>();   // chunk 0's value tile / beta / Tu have landed
    else wait_vm<N1 + NVB + NTU + N2 + 2 * NT>();
  }
  lds_barrier();                                               // PA
#ifdef IVL_TRACE
  if (ivl_trace_buf != nullptr && lane == 0 && trace_wg && L == 0) ivl_trace_buf[43] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  lds_barrier();                                               // P0: the V waves are done with tile 0
  vset(2);
  if (nt_seg > 1) wait_vm<N2 + N1 + 2 * NT + NVB>();           // H1(0) and chunk 1's set have landed
  else wait_vm<N2 + 2 * NT + NVB>();
  lds_barrier();                                               // P
  if constexpr (VD) load_tu<L, F8, DEV>(tc, ws_bh, 2, nt_seg, lane16);   // Tu buffer 0 is free: u(0) has been formed
  IVL_TVAR(lt_vmT); IVL_TVAR(lt_wT); IVL_TVAR(lt_iss2); IVL_TVAR(lt_vmM); IVL_TVAR(lt_wM); IVL_TVAR(lt_iss1);
  for (int ci = 0; ci < nt_seg; ++ci) {
    IVL_T(l0);
    // the T batch of this chunk (H2(ci), value tile / beta of chunk ci + 2) has landed: only the M batch issued behind it
    // (H1(ci+1), Tu(ci+2), a touch) may be in flight; ci = 0: only Tu(2) was issued behind value tile 2
    if (VD && ci == 0) wait_vm<NTU>();
    else if (ci + 1 < nt_seg) wait_vm<NMB>();
    else wait_vm<NTU + NT>();
    IVL_T(l1);
    lds_barrier();                                             // T(ci)
    IVL_T(l2);
    if (ci + 1 < nt_seg) load_h2<L, F8, DEV>(rec(ci + 1), img(ci + 1), lane16);
    vset(ci + 3);
    IVL_T(l3);
    if (ci + 1 < nt_seg) wait_vm<NTB + NTU + NT>();            // H1(ci+1) has landed: the rest of its M batch and the T batch behind it
    IVL_T(l4);
    lds_barrier();                                             // M(ci)
    IVL_T(l5);
    if (ci + 2 < nt_seg) load_h1<L, F8, DEV>(rec(ci + 2), img(ci + 2), lane16);
    if constexpr (VD) load_tu<L, F8, DEV>(tc, ws_bh, ci + 3, nt_seg, lane16);
    touch_chunk<TW, F8>(tc, ws_bh, ci + 5, nt_seg, lane);
    IVL_T(l6);
    IVL_TACC(lt_vmT, l1, l0); IVL_TACC(lt_wT, l2, l1); IVL_TACC(lt_iss2, l3, l2); IVL_TACC(lt_vmM, l4, l3); IVL_TACC(lt_wM, l5, l4);
    IVL_TACC(lt_iss1, l6, l5);
  }
#ifdef IVL_TRACE
  if (ivl_trace_buf != nullptr && lane == 0 && trace_wg) {
    long long* tb = ivl_trace_buf + 64 + 8 * L;
    tb[0] = lt_vmT; tb[1] = lt_wT; tb[2] = lt_iss2; tb[3] = lt_vmM; tb[4] = lt_wM; tb[5] = lt_iss1;
  }
#endif
  wait_vm<0>();                                                // no LDS-DMA may outlive the wave (the dummy area belongs to the workgroup)
  lds_barrier();                                               // F: the state waves' transposing store tail may use the images
}

// ---- the value side: V waves ------------------------------------------------------------------------------------------
// u = bf16(Tu (beta v)) does not depend on the state, so it is produced inside the scan by NCW dedicated waves that run one
// chunk AHEAD of the state waves and touch nothing the serial chain waits for:
//   conv phase  (between M(ci) and T(ci+1)): the V waves split the workgroup's 64 x (16 NCW) value tile of chunk ci + 2 by
//               rows (a thread = 4 consecutive tokens x 4 channels: whole 32 / 128-byte row pieces per load), apply the causal
//               width-4 convolution + SiLU when the call hands over the raw projection (VCONV; std:1253-1283, carry-in from /
//               carry-out to the conv state) and the beta scaling, and stage bf16(beta v) in LDS as B-operand fragment blocks;
//   mma phase   (between T(ci) and M(ci)): V wave p multiplies pair p's two staged blocks by the six non-zero blocks of Tu
//               (loaded lane-linearly from the record into registers, a chunk ahead), rounds to bf16 and stores the result
//               in the accumulator layout into the u slab of the next image - exactly where the state wave reads it.
// All memory operations of a V wave are ordinary compiler-tracked loads (no LDS-DMA): the counted vmcnt bookkeeping of the
// loader waves is not mixed with them.
struct ScanV {
  const bf16_t* v; long long ld;                       // value rows: v[(b T + t) ld + col0 + h GV + c], bf16
  int col0;
  const bf16_t* w;                                     // VCONV: conv taps [H GV, 1, 4] bf16
  const bf16_t* st_in; bf16_t* st_out;                 // VCONV: conv state [B, H GV, 4] bf16 (NULL: zero history / not wanted)
};

constexpr int SCAN_NV = 4;                             // V waves per workgroup: one per 16-row time tile of the chunk

template <int NCW, bool F8, bool VCONV, int SYNC>
__device__ __forceinline__ void scan_vwave(const ScanV sv, const unsigned char* ws_bh, unsigned char* smem, int vw, int slab_wg,
                                           int b, int h, int H, int T, int t_seg0, int nt_seg, int lane, bool trace_wg,
                                           const ScanSync& sy, int bh) {
  using R = Rec<F8>;
  (void)trace_wg;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  // V wave vw owns chunk rows 16 vw .. 16 vw + 15 of the workgroup's 16 NCW columns: a thread = 4 consecutive tokens (run) x
  // NCW channels (NP = NCW / 2 packed bf16 pairs: one 8- or 4-byte load per row), 4 runs x 16 channel groups per wave.
  constexpr int COLS = 16 * NCW, NP = NCW / 2;
  const int run = lane >> 4, cgp = lane & 15;
  const int row0 = vw * 16 + 4 * run;                              // chunk row of the thread's first token
  const int cw = NCW * cgp;                                        // first of its columns inside the workgroup's slab
  const int chan = slab_wg * COLS + cw;                            // ... inside the head
  unsigned char* stg = smem + Img<F8>::stg(NCW);                   // beta v as B-operand fragment blocks: (pair, s2) at 2048 pair + 1024 s2
  unsigned char* uslab = smem + Img<F8>::uslab(NCW) + vw * 512 + lane * 8;
  // staging address of the thread's four tokens: block s2 = vw / 2, lane group g = run, slots 0-3 (vw even) | 4-7 (vw odd)
  unsigned char* stg_t = stg + (vw >> 1) * 1024 + (run * 16) * 16 + (vw & 1) * 8;

  // Every global load of a V wave is issued at ONE point per chunk (right behind the consumption point of the previous
  // batch) and consumed one whole chunk later: vector-memory results return in order and the compiler's wait-count
  // bookkeeping across the loop's back edge is conservative, so a second consumption point per iteration would wait for
  // loads that are only half a chunk old.  Two register sets: the `_n` set is the load destination, take() moves it to the
  // working set (the moves are where the wave waits for memory) and the next batch is issued straight away.
  // The u product is split by ROW TILES over the V waves (wave vw: time tile m = vw, all pairs): a wave needs one or two
  // blocks of Tu instead of all six.
  // The data of chunk c lives in register set c & 1 (the loop below is unrolled by two so that the set index is a compile-time
  // constant): no copies between a "load" and a "working" set.
  unsigned int xr[2][7][NP];                                       // raw rows t-3 .. t+3 of the thread's run (VCONV: 7, else the last 4)
  u32x2 bt[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}};                    // beta of the thread's four tokens (bf16)
  u32x4 tu[2][2];                                                  // Tu blocks (vw, 0), (vw, 1); (vw < 2, 1) is zero and not loaded
  const long long ld = sv.ld;
  auto ld_row = [&](const bf16_t* p, unsigned int* d) {
    if constexpr (NP == 2) { const u32x2 x = *(const u32x2*)p; d[0] = x.x; d[1] = x.y; }
    else d[0] = *(const unsigned int*)p;
  };
  // row t - 3 of the thread's run in chunk 0 of the segment (may lie in front of the tensor: only dereferenced through at())
  const int trun = t_seg0 + row0 - 3;
  const bf16_t* vrun = sv.v + (size_t)b * T * ld + sv.col0 + h * GV + chan + (long long)trun * ld;
  auto at = [&](int tg) { return vrun + (long long)(tg - trun) * ld; };          // the thread's channels in row tg
  constexpr bool VTILE = NCW == 2;                                 // value rows come from the loaders' LDS tile, not from global loads
  const unsigned char* vtile = smem + Img<F8>::vt(NCW) + (vw * 16 + 4 * run) * 64 + cgp * 4;   // row t - 3 of the run, the thread's channel pair
  auto issue_loads = [&](int c, auto set_tag) {                    // v rows + beta + Tu of chunk c of the segment -> set S
    constexpr int S = decltype(set_tag)::value;
    if (c >= nt_seg) return;
    const int tc0 = t_seg0 + c * GC;
    if constexpr (VTILE) {
      return;                                                      // value rows, beta and Tu arrive by the loaders' LDS-DMA
    } else if (tc0 >= 3 && tc0 + GC <= T) {                        // interior chunk (wave-uniform): no clamping
      const bf16_t* p = vrun + (long long)c * GC * ld;
#pragma unroll
      for (int kk = VCONV ? 0 : 3; kk < 7; ++kk) ld_row(p + kk * ld, xr[S][kk]);
    } else {
#pragma unroll
      for (int kk = VCONV ? 0 : 3; kk < 7; ++kk) {
        int tg = tc0 + row0 - 3 + kk;
        tg = tg < 0 ? 0 : (tg > T - 1 ? T - 1 : tg);
        ld_row(at(tg), xr[S][kk]);
      }
    }
    const unsigned char* rec = ws_bh + (size_t)c * R::STRIDE;
    bt[S] = *(const u32x2*)(rec + R::BETA + row0 * 2);
    tu[S][0] = *(const u32x4*)(rec + R::TU + tri_blk(vw, 0) * 1024 + lane * 16);
    if (vw >= 2) tu[S][1] = *(const u32x4*)(rec + R::TU + tri_blk(vw, 1) * 1024 + lane * 16);
  };
  // the consumption point of a batch: every register of set S passes through an empty asm statement, so the compiler waits
  // for the whole batch HERE (one chunk after it was issued) and treats the values as landed from then on -- the mma of
  // the next phase must not wait again (its wait would also cover the batch issued in between)
  auto landed = [&](auto set_tag) {
    constexpr int S = decltype(set_tag)::value;
    if constexpr (VTILE) return;
#pragma unroll
    for (int kk = VCONV ? 0 : 3; kk < 7; ++kk)
#pragma unroll
      for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(xr[S][kk][p]));
    asm volatile("" : "+v"(bt[S]));
    asm volatile("" : "+v"(tu[S][0]));
    if (vw >= 2) asm volatile("" : "+v"(tu[S][1]));
  };
  // conv taps of the thread's channel pairs (pair p = channels 2p, 2p + 1 of the thread), kept packed (bf16): unpacked per use
  u32x4 wpk[NP];
  if constexpr (VCONV) {
    const u32x4* wp = (const u32x4*)(sv.w + ((size_t)h * GV + chan) * 4);
#pragma unroll
    for (int p = 0; p < NP; ++p) wpk[p] = wp[p];
  }
  // [conv + SiLU ->] bf16 -> * beta -> bf16, four tokens x NCW channels, staged as 8-byte half pieces (four consecutive times
  // of one column).  Same arithmetic and rounding points as ivl_short_conv_fwd / gdn_prologue_kernel followed by the beta
  // scaling of the round-2 pre-pass.
  unsigned int hist[3][NP] = {};                                   // history rows of the thread that holds time 0 (conv state)
  bool use_hist = false;
  auto conv_stage = [&](int c, auto set_tag) {
    constexpr int S = decltype(set_tag)::value;
    const f32x2 nl2e = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.f, 1.f};
    unsigned int xv[7][NP];                                        // the run's rows t-3 .. t+3
    u32x2 btv = bt[S];
    if constexpr (VTILE) {
      btv = *(const u32x2*)(smem + Img<F8>::betab(NCW) + (c & 1) * Img<F8>::BETA_BYTES + row0 * 2);
      const unsigned char* tp = vtile + (c & 1) * Img<F8>::VT_BYTES;
#pragma unroll
      for (int kk = VCONV ? 0 : 3; kk < 7; ++kk) xv[kk][0] = *(const unsigned int*)(tp + kk * 64);
      if constexpr (VCONV) {
        if (use_hist && c == 0) {
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) xv[kk][0] = hist[kk][0];
        }
      }
    } else {
#pragma unroll
      for (int kk = VCONV ? 0 : 3; kk < 7; ++kk)
#pragma unroll
        for (int p = 0; p < NP; ++p) xv[kk][p] = xr[S][kk][p];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      float res[4][2];                                             // [token][channel of the pair]
      f32x2 win[3], wf[4];
      if constexpr (VCONV) {
        const u32x4 w0 = wpk[p];
        wf[0] = f32x2{bflo(w0.x), bflo(w0.z)}; wf[1] = f32x2{bfhi(w0.x), bfhi(w0.z)};
        wf[2] = f32x2{bflo(w0.y), bflo(w0.w)}; wf[3] = f32x2{bfhi(w0.y), bfhi(w0.w)};
#pragma unroll
        for (int k = 0; k < 3; ++k) win[k] = f32x2{bflo(xv[k][p]), bfhi(xv[k][p])};
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x2 cur = f32x2{bflo(xv[k + 3][p]), bfhi(xv[k + 3][p])};
        f32x2 o;
        if constexpr (VCONV) {
          f32x2 a = wf[0] * win[0];
          a = __builtin_elementwise_fma(wf[1], win[1], a);
          a = __builtin_elementwise_fma(wf[2], win[2], a);
          a = __builtin_elementwise_fma(wf[3], cur, a);
          f32x2 e = a * nl2e;
          e = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + one;
          o = a * f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
          win[0] = win[1]; win[1] = win[2]; win[2] = cur;
          const unsigned int ob = pack2bf(o[0], o[1]);             // the conv output is a bf16 tensor in the reference
          o = f32x2{bflo(ob), bfhi(ob)};
        } else {
          o = cur;
        }
        const float btk = (k & 1) ? bfhi(k < 2 ? btv.x : btv.y) : bflo(k < 2 ? btv.x : btv.y);
        res[k][0] = o[0] * btk;
        res[k][1] = o[1] * btk;
      }
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const int col = cw + 2 * p + ch;                           // column inside the slab: pair col / 16, lane j = col % 16
        *(u32x2*)(stg_t + (col >> 4) * 2048 + (col & 15) * 16) = u32x2{pack2bf(res[0][ch], res[1][ch]), pack2bf(res[2][ch], res[3][ch])};
      }
    }
  };
  // u = bf16(Tu (beta v)), time tile vw x every pair, into the workgroup's u slab (single buffer: the state waves read u(c)
  // right behind the barrier that follows this phase, u(c + 1) is written a whole chunk later).  All B fragments first, then
  // all products, then the conversions: the wave shares its SIMD's matrix pipe with an output wave.
  auto mma_u = [&](int c, auto set_tag) {
    constexpr int S = decltype(set_tag)::value;
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 t0 = tu[S][0], t1 = tu[S][1];
    if constexpr (VTILE) {
      const unsigned char* tb = smem + Img<F8>::tub(NCW) + (c & 1) * Img<F8>::TU_BYTES + lane * 16;
      t0 = *(const u32x4*)(tb + tri_blk(vw, 0) * 1024);
      if (vw >= 2) t1 = *(const u32x4*)(tb + tri_blk(vw, 1) * 1024);
    }
#pragma unroll
    for (int p0 = 0; p0 < NCW; p0 += 2) {                          // two pairs at a time (register budget of the 16-wave workgroup)
      u32x4 bf[2][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        bf[p][0] = *(const u32x4*)(stg + (p0 + p) * 2048 + lane * 16);
        if (vw >= 2) bf[p][1] = *(const u32x4*)(stg + (p0 + p) * 2048 + 1024 + lane * 16);
      }
      f32x4 acc[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf(t0), mf(bf[p][0]), z, 0, 0, 0);
      if (vw >= 2) {
#pragma unroll
        for (int p = 0; p < 2; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf(t1), mf(bf[p][1]), acc[p], 0, 0, 0);
      }
#pragma unroll
      for (int p = 0; p < 2; ++p)
        *(u32x2*)(uslab + (p0 + p) * 2048) = u32x2{pack2bf(acc[p][0], acc[p][1]), pack2bf(acc[p][2], acc[p][3])};
    }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  IVL_T(tv_0);
  issue_loads(0, S0{});                  // chunks 0 and 1: both in flight together
  issue_loads(1, S1{});
  if constexpr (VCONV) {
    // carry-in / carry-out of the conv state: the thread that holds time 0 (first segment, V wave 0, run 0) reads the old state
    // of its channels (taps 1..3 = times -3, -2, -1) and writes the new one = the last four inputs of the sequence
    // ([old state, x] when T < 4) -- read before written by the same thread, so st_out may alias st_in
    if (t_seg0 == 0 && row0 == 0) {
      const size_t st_off = ((size_t)b * H * GV + (size_t)h * GV + chan) * 4;
      u32x4 st[NP];
      unsigned int tl[4][NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) st[p] = sv.st_in != nullptr ? *(const u32x4*)(sv.st_in + st_off + 8 * p) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int tg = T - 4 + j;
        tg = tg < 0 ? 0 : tg;
        ld_row(at(tg), tl[j]);
      }
      constexpr unsigned int HI = 0x07060302u, LO = 0x05040100u;   // (even.hi, odd.hi) | (even.lo, odd.lo) of a word pair
      use_hist = true;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        hist[0][p] = __builtin_amdgcn_perm(st[p].z, st[p].x, HI);  // tap 1 of the pair's two channels
        hist[1][p] = __builtin_amdgcn_perm(st[p].w, st[p].y, LO);  // tap 2
        hist[2][p] = __builtin_amdgcn_perm(st[p].w, st[p].y, HI);  // tap 3
        if constexpr (!VTILE) { xr[0][0][p] = hist[0][p]; xr[0][1][p] = hist[1][p]; xr[0][2][p] = hist[2][p]; }
      }
      if (sv.st_out != nullptr) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          unsigned int rw[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = T + j;                                   // ext[T + j], ext = [state(4), x(T)]
            rw[j] = (T >= 4 || e >= 4) ? tl[j][p] : (e == 1 ? hist[0][p] : (e == 2 ? hist[1][p] : hist[2][p]));
          }
          *(u32x4*)(sv.st_out + st_off + 8 * p) = u32x4{__builtin_amdgcn_perm(rw[1], rw[0], LO), __builtin_amdgcn_perm(rw[3], rw[2], LO),
                                                        __builtin_amdgcn_perm(rw[1], rw[0], HI), __builtin_amdgcn_perm(rw[3], rw[2], HI)};
        }
      }
    }
  }
  if constexpr (SYNC == 2) {
    if (vw == 0 && lane == 0) scan_abort_words<F8>(smem)[4] = 0u;
  }
  if constexpr (SYNC != 0) {
    lds_barrier();                       // PF: loader 0 has seen the flags (or its wait failed)
    if (scan_aborted<F8, false>(smem)) return;
  }
  lds_barrier();                         // PA: value tile 0 has landed (NCW = 2)
  landed(S0{});
  conv_stage(0, S0{});                   // beta v of chunk 0
  IVL_T(tv_1);
  lds_barrier();                         // P0
  mma_u(0, S0{});                        // u(0)
  IVL_T(tv_2);
  lds_barrier();                         // P
  IVL_T(tv_3);
  landed(S1{});                          // chunk 1
  if (nt_seg > 1) conv_stage(1, S1{});
  issue_loads(2, S0{});
  IVL_TVAR(tv_wT); IVL_TVAR(tv_wM); IVL_TVAR(tv_mma); IVL_TVAR(tv_conv); IVL_TVAR(tv_x1); IVL_TVAR(tv_x2); IVL_TVAR(tv_x3);
  // iteration ci: [T] u(ci + 1) from set (ci + 1) & 1 [M] beta v of chunk ci + 2 from set ci & 1, then chunk ci + 3 -> set (ci + 1) & 1
  // (whose Tu the mma of this iteration has just consumed)
  // Single-launch form: behind T(ci) the loader waves request beta (and, behind M(ci), Tu) of chunk ci + 3 from the record.
  // V wave 0 -- which issues no other vector-memory instruction in this form -- arrives at T(ci) only once that record is
  // published: the barrier carries the guarantee to the loaders (whose counted vmcnt bookkeeping a poll of their own would
  // break).  The flag word is requested one chunk before it is looked at (a device-scope load is a full memory round trip).
  unsigned int gate_val = 0;
  bool gate_dead = false;                                            // a wait has run out: the workgroup is stopping, no more polls
  auto gate_issue = [&](int c) {
    if (c < nt_seg) gate_val = __hip_atomic_load(sy.flags + bh * SYNC_HEAD_WORDS + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto gate_wait = [&](int c) {
    if (c >= nt_seg || gate_dead) return;
    unsigned int v = gate_val;
    const unsigned int want = c < sy.nsplit ? 2u : 1u;
    for (int spin = 0; v != want && spin < SYNC_SPIN_BOUND; ++spin) {       // bounded like scan_wait_records
      __builtin_amdgcn_s_sleep(4);
      v = __hip_atomic_load(sy.flags + bh * SYNC_HEAD_WORDS + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((spin & 31) == 31 && __hip_atomic_load(sy.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // the area has failed elsewhere
    }
    if (v != want) {                                                 // (wave-uniform) tell the workgroup through LDS, the host through the area
      gate_dead = true;
      if (lane == 0) {
        sync_fail(sy, SYNC_E_GATE, bh, c);
        scan_abort_words<F8>(smem)[4] = 1u;
      }
    }
  };
  if constexpr (SYNC != 0) {
    static_assert(VTILE, "the gate wave must have no vector-memory traffic of its own");
    if (SYNC == 2 && vw == 0) gate_issue(3);
  }
  auto body = [&](int ci, auto even_tag) {
    constexpr int E = decltype(even_tag)::value;                   // ci & 1
    using SA = std::integral_constant<int, E>;
    using SB = std::integral_constant<int, 1 - E>;
    if constexpr (SYNC != 0) {
      if (SYNC == 2 && vw == 0) { gate_wait(ci + 3); gate_issue(ci + 4); }
    }
    IVL_T(ta);
    lds_barrier();                       // T(ci): beta v of chunk ci + 1 is staged
    IVL_T(tb);
    if (ci + 1 < nt_seg) mma_u(ci + 1, SB{});    // u(ci + 1), with Tu(ci + 1)
    IVL_T(tc);
    lds_barrier();                       // M(ci)
    IVL_T(td);
    landed(SA{});                        // chunk ci + 2 (issued a whole chunk ago)
    IVL_T(tb2);
    if (ci + 2 < nt_seg) conv_stage(ci + 2, SA{});   // beta v of chunk ci + 2
    IVL_T(tb3);
    issue_loads(ci + 3, SB{});           // behind the conv: the loaders' H1 burst that follows M has left the vector-memory path by then
    IVL_T(te);
    IVL_TACC(tv_wT, tb, ta); IVL_TACC(tv_mma, tc, tb); IVL_TACC(tv_wM, td, tc); IVL_TACC(tv_conv, te, td);
    IVL_TACC(tv_x2, tb2, td); IVL_TACC(tv_x1, tb3, tb2); IVL_TACC(tv_x3, te, tb3);
  };
  for (int ci = 0; ci < nt_seg; ci += 2) {
    body(ci, S0{});
    if (ci + 1 < nt_seg) body(ci + 1, S1{});
  }
  lds_barrier();                         // F
#ifdef IVL_TRACE
  if (ivl_trace_buf != nullptr && lane == 0 && vw == 0 && trace_wg) {
    ivl_trace_buf[31] = tv_1 - tv_0; ivl_trace_buf[32] = tv_2 - tv_1; ivl_trace_buf[33] = tv_3 - tv_2;
    ivl_trace_buf[34] = tv_wT; ivl_trace_buf[35] = tv_mma; ivl_trace_buf[36] = tv_wM; ivl_trace_buf[37] = tv_conv;
    ivl_trace_buf[38] = tv_x1; ivl_trace_buf[39] = tv_x2; ivl_trace_buf[40] = tv_x3;
  }
  if (ivl_trace_buf != nullptr && lane == 0 && trace_wg) {
    ivl_trace_buf[104 + 2 * vw] = tv_wT; ivl_trace_buf[105 + 2 * vw] = tv_wM;
  }
#endif
}

// Workgroup = NCW state waves + NCW output waves + SCAN_NL (4) loader waves + NCW V waves; pair w owns state columns v0..v0+15.
//   state wave  : S (accumulators), per chunk  sb = bf16(S) -> LDS | T | v_new = u + Wn sb -> LDS | M | S = egl S + Kd^T v_new
//   output wave : per chunk                                         T | (q_hat sb)^T e^gamma    | M | + v_new^T Aqk^T, store o
//   V wave      : per chunk  (one chunk ahead)                      T | u = Tu (beta v) -> image | M | beta v of the chunk after
// A single wave issues at most one instruction per ~4-5 cycles, and a chunk needs ~290 of them per slab: split over two
// waves of the same SIMD the serial S -> v_new -> S chain carries 32 MFMAs + the conversions only, the other 22 MFMAs, the
// scaling and the stores run beside it.  sb / v_new cross through 6 KB of LDS per pair, ordered by the two barriers the
// loader protocol needs anyway.  NCW = 4: 64 columns per workgroup (operand image shared by four pairs: least L2 traffic);
// NCW = 2: 32 columns, twice the workgroups -- used while the grid would otherwise leave most of the chip idle.
// fp8 (e4m3) operand variant (F8): the four A-operand matrices arrive as 512-byte blocks, the state and v_new are
// converted to e4m3 for the products (v_mfma_f32_16x16x32_fp8_fp8: same lane layout, 8 bytes per fragment); accumulators,
// the carried state, u (and the product that makes it) and the output stay fp32 / bf16.
template <bool F8> struct FragT { typedef u32x4 type; };
template <> struct FragT<true> { typedef u32x2 type; };
template <bool F8>
__device__ __forceinline__ f32x4 mma16(typename FragT<F8>::type a, typename FragT<F8>::type b, f32x4 c) {
  if constexpr (F8) {
    long la, lb;
    __builtin_memcpy(&la, &a, 8);
    __builtin_memcpy(&lb, &b, 8);
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf(a), mf(b), c, 0, 0, 0);
  }
}
template <bool F8>
__device__ __forceinline__ typename FragT<F8>::type to_frag(f32x4 lo, f32x4 hi) {     // slots 0..3 <- lo, 4..7 <- hi
  if constexpr (F8) return u32x2{pack4_fp8(lo[0], lo[1], lo[2], lo[3]), pack4_fp8(hi[0], hi[1], hi[2], hi[3])};
  else return pack8(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
}

// wave -> role: NCW = 4: state 0-3, output 4-7, loaders 8-11, V 12-15 (one of each per SIMD);
//               NCW = 2: state 0-1, output 2-3, loaders 4-5, V 6-9, loaders 10-11 (waves go to the SIMDs in the cyclic order
//               0, 2, 1, 3: every SIMD hosts one V wave -- two of them on one SIMD made the younger one the last arrival
//               at every barrier)
enum { ROLE_STATE = 0, ROLE_OUT = 1, ROLE_LOAD = 2, ROLE_V = 3 };
template <int NCW>
__device__ __forceinline__ void scan_role(int w, int& role, int& idx) {
  if (w < NCW) { role = ROLE_STATE; idx = w; }
  else if (w < 2 * NCW) { role = ROLE_OUT; idx = w - NCW; }
  else if (NCW == 4) {
    if (w < 12) { role = ROLE_LOAD; idx = w - 8; }
    else { role = ROLE_V; idx = w - 12; }
  } else {
    const int q = (w - 4) >> 1, r = (w - 4) & 1;            // q = 0: loaders 0,1; 1: V 0,1; 2: V 2,3; 3: loaders 2,3
    role = (q == 1 || q == 2) ? ROLE_V : ROLE_LOAD;
    idx = (q == 0 || q == 1) ? r : 2 + r;
  }
}

// One scan workgroup: batch*head bx, column slab by.  SYNC (single-launch form): the records are written by pre-pass
// workgroups of the SAME launch; `sync` -> ScanSync tells the loader waves where to wait for them.
template <int NCW, bool F8, bool VCONV, int SYNC>
__device__ __forceinline__ void gdn_chunk_scan_body(
    unsigned char* smem, const int bx, const int by,
    const unsigned char* __restrict__ ws, bf16_t* __restrict__ o, const ScanV& sv,
    const void* h0, int h0_dtype, void* ht, int ht_dtype,
    int T, int H, int t_seg0, int nt_seg, float scale, const ScanSync sy) {
  using R = Rec<F8>;
  const bool trace_wg = bx == 0 && by == 0;
  (void)trace_wg;
  using frag_t = typename FragT<F8>::type;
  constexpr int IMG_BYTES = Img<F8>::BYTES, BLK = R::BLK;

  IVL_T(ts0);
#ifdef IVL_TRACE
  const long long rt0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  IVL_TVAR(t_bar); IVL_TVAR(t_bar2); IVL_TVAR(t_h1); IVL_TVAR(t_h2); IVL_TVAR(t_sb);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int j = lane & 15, g = lane >> 4;
  // grid = (B*H, 16/NCW): linear block id = bh + B*H*slab, so with B*H % 8 == 0 all column slabs of one head run on
  // the same XCD (id % 8) and share its L2 for the record they all read.
  const int bh = bx;
  const int b = bh / H, h = bh % H;
  const unsigned int lds0 = __builtin_amdgcn_readfirstlane((unsigned int)(size_t)smem);
  const unsigned int lane16 = lane * 16;                          // DMA piece offset
  const int lanef = lane * (BLK / 64);                            // fragment offset inside a block
  const unsigned char* ws_bh = ws + (size_t)bh * nt_seg * R::STRIDE;
  int role, ridx;
  scan_role<NCW>(wave_u, role, ridx);

  if (role == ROLE_LOAD) {                                      // ---- loader waves ----
    static_assert(SCAN_NL == 4, "loader dispatch below");
    ScanTouch tc;
    tc.vrow0 = (const unsigned char*)(sv.v + (size_t)b * T * sv.ld + sv.col0 + h * GV + by * (16 * NCW));
    tc.row_bytes = (unsigned int)sv.ld * 2u;
    tc.T = T; tc.t_seg0 = t_seg0;
    tc.dummy = lds0 + (unsigned int)Img<F8>::dummy(NCW);
    tc.vt = lds0 + (unsigned int)Img<F8>::vt(NCW);
    tc.tub = lds0 + (unsigned int)Img<F8>::tub(NCW);
    tc.betab = lds0 + (unsigned int)Img<F8>::betab(NCW);
    if (ridx == 0) scan_loader<0, NCW, F8, SYNC>(ws_bh, nt_seg, smem, lds0, lane16, tc, lane, sy, bh, trace_wg);
    else if (ridx == 1) scan_loader<1, NCW, F8, SYNC>(ws_bh, nt_seg, smem, lds0, lane16, tc, lane, sy, bh, trace_wg);
    else if (ridx == 2) scan_loader<2, NCW, F8, SYNC>(ws_bh, nt_seg, smem, lds0, lane16, tc, lane, sy, bh, trace_wg);
    else scan_loader<3, NCW, F8, SYNC>(ws_bh, nt_seg, smem, lds0, lane16, tc, lane, sy, bh, trace_wg);
    return;
  }
  if (role == ROLE_V) {                                         // ---- V waves ----
    scan_vwave<NCW, F8, VCONV, SYNC>(sv, ws_bh, smem, ridx, by, b, h, H, T, t_seg0, nt_seg, lane, trace_wg, sy, bh);
    return;
  }
  const int pair = ridx;
  const int v0 = (by * NCW + pair) * 16;                // first state column of this pair
  unsigned char* xsb = smem + Img<F8>::xch(NCW) + pair * Img<F8>::XCH_PAIR;   // sb: 4 fragment blocks, lane-linear
  unsigned char* xvn = xsb + 4 * BLK;                                         // v_new: 2 fragment blocks
  auto frag = [&](const unsigned char* im, int off, int idx) { return *(const frag_t*)(im + off + idx * BLK + lanef); };

  if (role == ROLE_OUT) {
    // =========================== output wave ===========================
    frag_t fq[16];
    float egv[4];
    // Once every wait of this workgroup on the head's flags lies behind it, the last of the head's scan workgroups to get here
    // clears them (and the count) for the next launch.  (The returning atomic is a memory round trip: where all records are
    // awaited at the start it sits here, behind PA -- every loader wave has seen the flags -- and costs nothing.)
    auto flags_done = [&]() {
      if (pair == 0 && lane == 0) {
        if (atomicAdd(sy.headdone + bh, 1u) == (unsigned int)(16 / NCW - 1)) {
          for (int c = 0; c < nt_seg; ++c) __hip_atomic_store(sy.flags + bh * SYNC_HEAD_WORDS + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(sy.headdone + bh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    };
    if constexpr (SYNC != 0) {
      lds_barrier();                     // PF
      if (scan_aborted<F8, false>(smem)) return;     // the wait for the records failed: no output from this workgroup, flags left as they are
    }
    lds_barrier();                       // PA
    if constexpr (SYNC != 0) {
      if (SYNC == 1) flags_done();
    }
    lds_barrier();                       // P0
    lds_barrier();                       // P: H1(0) has landed
#pragma unroll
    for (int i = 0; i < 16; ++i) fq[i] = frag(smem, R::QH, i);
#pragma unroll
    for (int m = 0; m < 4; ++m) egv[m] = *(const float*)(smem + R::EG + (16 * m + j) * 4);
    bf16_t* orow = o + (((size_t)b * T + t_seg0 + j) * H + h) * GV + v0 + 4 * g;
    const size_t ostep = (size_t)16 * H * GV;                   // 16 tokens further
    IVL_TVAR(ot_wT); IVL_TVAR(ot_A); IVL_TVAR(ot_wM); IVL_TVAR(ot_B);
    for (int ci = 0; ci < nt_seg; ++ci) {
      const unsigned char* img = smem + (ci & 1) * IMG_BYTES;
      const unsigned char* img_next = smem + ((ci + 1) & 1) * IMG_BYTES;
      const int tc0 = t_seg0 + ci * GC;
      IVL_T(o0);
      lds_barrier();                     // T(ci): sb(ci) published, H2(ci) landed
      IVL_T(o1);
      unsigned int gate_abort = 0u;      // long calls: the gate wave's word, read here, looked at in front of the stores
      if constexpr (SYNC == 2) gate_abort = scan_abort_words<F8>(smem)[4];
      frag_t sb[4], fa[6];
#pragma unroll
      for (int s = 0; s < 4; ++s) sb[s] = *(const frag_t*)(xsb + s * BLK + lanef);
#pragma unroll
      for (int i = 0; i < 6; ++i) fa[i] = frag(img, R::AQK, i);             // tri_blk order: (0,0) (1,0) (2,0) (2,1) (3,0) (3,1)
      // (q_hat S)^T: lane (g, j) register r <-> column v0 + 4g + r, time 16m + j
      f32x4 accO[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) accO[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          accO[m] = mma16<F8>(sb[s], fq[4 * m + s], accO[m]);
#pragma unroll
      for (int m = 0; m < 4; ++m) accO[m] *= egv[m];
      IVL_T(o2);
      lds_barrier();                     // M(ci): v_new(ci) published, H1(ci+1) landed
      IVL_T(o3);
      frag_t vn[2];
      vn[0] = *(const frag_t*)(xvn + lanef);
      vn[1] = *(const frag_t*)(xvn + BLK + lanef);
#pragma unroll
      for (int i = 0; i < 16; ++i) fq[i] = frag(img_next, R::QH, i);        // next chunk (stale but harmless after the last one)
#pragma unroll
      for (int m = 0; m < 4; ++m) egv[m] = *(const float*)(img_next + R::EG + (16 * m + j) * 4);
      // + v_new^T Aqk^T (the blocks (m < 2, s2 = 1) lie strictly above the diagonal)
      accO[0] = mma16<F8>(vn[0], fa[0], accO[0]);
      accO[1] = mma16<F8>(vn[0], fa[1], accO[1]);
      accO[2] = mma16<F8>(vn[0], fa[2], accO[2]);
      accO[3] = mma16<F8>(vn[0], fa[4], accO[3]);
      accO[2] = mma16<F8>(vn[1], fa[3], accO[2]);
      accO[3] = mma16<F8>(vn[1], fa[5], accO[3]);
      if constexpr (SYNC == 2) {
        if (__builtin_amdgcn_readfirstlane(gate_abort) != 0u) return;   // a later chunk's record never came: stop storing
      }
      // the step shape (single launch, SYNC == 1) writes its 2 MB of o through (`sc0 sc1`: they leave during the kernel instead of in
      // its end-of-kernel write-back: 17.2 -> 16.6 us in step); a long call's 33 MB written through back up into the output waves
      // (in-call 101.7 -> 118 us at T = 4096): plain stores there
      auto put = [&](bf16_t* dst, u32x2 w) __attribute__((always_inline)) {
        if constexpr (SYNC == 1) store_out8(dst, w);
        else *(u32x2*)dst = w;
      };
      if (tc0 + GC <= T) {               // full chunk (wave-uniform): four unconditional 8-byte row stores
#pragma unroll
        for (int m = 0; m < 4; ++m)
          put(orow + m * ostep, u32x2{pack2bf(accO[m][0] * scale, accO[m][1] * scale), pack2bf(accO[m][2] * scale, accO[m][3] * scale)});
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (tc0 + 16 * m + j < T)
            put(orow + m * ostep, u32x2{pack2bf(accO[m][0] * scale, accO[m][1] * scale), pack2bf(accO[m][2] * scale, accO[m][3] * scale)});
      }
      orow += 4 * ostep;
      IVL_T(o4);
      IVL_TACC(ot_wT, o1, o0); IVL_TACC(ot_A, o2, o1); IVL_TACC(ot_wM, o3, o2); IVL_TACC(ot_B, o4, o3);
    }
#ifdef IVL_TRACE
    if (ivl_trace_buf != nullptr && lane == 0 && pair == 0 && trace_wg) {
      ivl_trace_buf[100] = ot_wT; ivl_trace_buf[101] = ot_A; ivl_trace_buf[102] = ot_wM; ivl_trace_buf[103] = ot_B;
    }
    if (ivl_trace_buf != nullptr && lane == 0 && trace_wg) {
      ivl_trace_buf[112 + 2 * pair] = ot_wT; ivl_trace_buf[113 + 2 * pair] = ot_wM;
    }
#endif
    if constexpr (SYNC != 0) {
      if (SYNC == 2) flags_done();       // long calls: the last chunk's barriers have been passed
    }
    lds_barrier();                       // F
    return;
  }

  // =========================== state wave ===========================
  // state slab: tile t = rows 16t..16t+15, lane (g, j) register r <-> S[16t + 4g + r][v0 + j]
  // A bf16 initial state arrives in whole row pieces of the workgroup's columns (16-byte vectors, 32 NCW bytes per row) and is
  // transposed through LDS (the exchange area, free before the first chunk).  (An fp32 state keeps the element-wise path: its
  // tile does not fit the exchange area.)
  constexpr int TROW = 32 * NCW + 16;                              // bytes per tile row (+16: bank spread of the column reads)
  static_assert(GK * TROW <= NCW * Img<F8>::XCH_PAIR || F8, "state tile must fit the exchange area");
  constexpr bool TILE_OK = GK * TROW <= NCW * Img<F8>::XCH_PAIR;
  const int stid = pair * 64 + lane;                               // thread index among the state waves
  const size_t srow0 = (size_t)bh * GK * GV + (size_t)by * (16 * NCW);   // element offset of the workgroup's columns in row 0
  f32x4 S[8];
  bool tiled_in = false;
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
    } else if constexpr (TILE_OK) {
      // 128 rows x (2 NCW) 16-byte pieces, 4 per thread: global -> registers -> tile (read back behind P0)
      unsigned char* tile = smem + Img<F8>::xch(NCW);
      u32x4 pc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = stid + 64 * NCW * i, row = idx / (2 * NCW), pq = idx % (2 * NCW);
        pc[i] = *(const u32x4*)((const bf16_t*)h0 + srow0 + (size_t)row * GV + 8 * pq);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = stid + 64 * NCW * i, row = idx / (2 * NCW), pq = idx % (2 * NCW);
        *(u32x4*)(tile + row * TROW + 16 * pq) = pc[i];
      }
      tiled_in = true;
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
  // Wn fragments, u and e^gamma_L of the NEXT chunk are loop-carried: requested a phase ahead of their use
  frag_t fw[16];
  u32x2 uu[4];
  float egl;
  auto load_h1_frags = [&](const unsigned char* im) {
#pragma unroll
    for (int i = 0; i < 16; ++i) fw[i] = frag(im, R::WN, i);
#pragma unroll
    for (int m = 0; m < 4; ++m) uu[m] = *(const u32x2*)(smem + Img<F8>::uslab(NCW) + pair * 2048 + m * 512 + lane * 8);
    egl = *(const float*)(im + R::EGL);
  };
  if constexpr (SYNC != 0) {
    lds_barrier();                       // PF
    if (scan_aborted<F8, false>(smem)) return;
  }
  lds_barrier();                         // PA
  lds_barrier();                         // P0
  if constexpr (TILE_OK) {
    if (tiled_in) {                      // (workgroup-uniform) own slab out of the tile: column 16 pair + j, rows 16t + 4g + r
      const unsigned char* tp = smem + Img<F8>::xch(NCW) + (4 * g) * TROW + (16 * pair + j) * 2;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[t][r] = bf2f(*(const bf16_t*)(tp + (16 * t + r) * TROW));
    }
  }
  lds_barrier();                         // P: H1(0) has landed, u(0) is in place; the tile has been read (the exchange area is free)
  load_h1_frags(smem);

  for (int ci = 0; ci < nt_seg; ++ci) {
    const unsigned char* img = smem + (ci & 1) * IMG_BYTES;
    const unsigned char* img_next = smem + ((ci + 1) & 1) * IMG_BYTES;
    IVL_T(tc_0);
    // ---- B-operand fragments: the state (bf16); published for the output wave --------------------------------------
    frag_t sb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sb[s] = to_frag<F8>(S[2 * s], S[2 * s + 1]);
      *(frag_t*)(xsb + s * BLK + lanef) = sb[s];
    }
    // work that needs nothing from the other waves sits between the publish and the barrier: it covers the LDS-write drain
    f32x4 accV[4];                       // u (bf16) is the C input of v_new = u + Wn S
#pragma unroll
    for (int m = 0; m < 4; ++m) accV[m] = f32x4{bflo(uu[m].x), bfhi(uu[m].x), bflo(uu[m].y), bfhi(uu[m].y)};
#pragma unroll
    for (int t = 0; t < 8; ++t) S[t] *= egl;
    IVL_T(tc_a);
    lds_barrier();                       // T(ci): H2 of this chunk has landed
    IVL_T(tc_b);
    frag_t fk[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) fk[i] = frag(img, R::KDT, i);              // land under the 16 MFMAs below
    // ---- v_new = u + Wn S   (time tiles m, lane (g, j) register r <-> time 16m + 4g + r, column j) -------------------------
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        accV[m] = mma16<F8>(fw[4 * m + s], sb[s], accV[m]);
    frag_t vn[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      vn[s2] = to_frag<F8>(accV[2 * s2], accV[2 * s2 + 1]);
      *(frag_t*)(xvn + s2 * BLK + lanef) = vn[s2];
    }
    IVL_T(tc_c);
    lds_barrier();                       // M(ci): H1 of the next chunk has landed; every wave is done with H1 of this one
    IVL_T(tc_d);
    load_h1_frags(img_next);             // chunk ci + 1 (stale but harmless data after the last chunk)
    // ---- S = e^{gamma_L} S + Kd^T v_new ----------------------------------------------------------------------------
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int t = 0; t < 8; ++t)
        S[t] = mma16<F8>(fk[2 * t + s2], vn[s2], S[t]);
    IVL_T(tc_e);
    IVL_TACC(t_bar, tc_b, tc_a); IVL_TACC(t_bar2, tc_d, tc_c); IVL_TACC(t_h1, tc_c, tc_b); IVL_TACC(t_h2, tc_e, tc_d); IVL_TACC(t_sb, tc_a, tc_0);
  }
  IVL_T(ts1);

  // the final state leaves element-wise (an LDS-transposed store was measured: the workgroup barrier it needs in front of the
  // row stores -- every wave must be done with the images -- costs more than the 32 two-byte stores per lane save)
  bool keep_state = ht != nullptr;
  if constexpr (SYNC == 2) keep_state = keep_state && !scan_aborted<F8, true>(smem);   // a stopped workgroup stores no state
  if (keep_state) {
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
  lds_barrier();                         // F (matches the other roles' count)
  IVL_T(ts2);
  IVL_TOUT_WG(trace_wg, 16, ts0); IVL_TOUT_WG(trace_wg, 17, t_bar); IVL_TOUT_WG(trace_wg, 18, t_h1); IVL_TOUT_WG(trace_wg, 19, t_h2); IVL_TOUT_WG(trace_wg, 20, ts1 - ts0); IVL_TOUT_WG(trace_wg, 21, ts2 - ts1);
#ifdef IVL_TRACE
  IVL_TOUT_WG(trace_wg, 25, (long long)__builtin_amdgcn_s_memrealtime() - rt0);
  IVL_TOUT_WG(trace_wg, 41, rt0);
  IVL_TOUT_WG(trace_wg, 44, (long long)__builtin_amdgcn_s_memrealtime());
#endif
  IVL_TOUT_WG(trace_wg, 22, ts2); IVL_TOUT_WG(trace_wg, 23, t_bar2); IVL_TOUT_WG(trace_wg, 24, t_sb);
#ifdef IVL_TRACE
  if (ivl_trace_buf != nullptr && lane == 0 && trace_wg) {
    ivl_trace_buf[120 + 2 * pair] = t_bar; ivl_trace_buf[121 + 2 * pair] = t_bar2;
  }
#endif
}


template <int NCW, bool F8, bool VCONV>
__global__ __launch_bounds__(64 * (2 * NCW + SCAN_NL + SCAN_NV)) void gdn_chunk_scan_kernel(
    const unsigned char* __restrict__ ws, bf16_t* __restrict__ o, ScanV sv,
    const void* h0, int h0_dtype, void* ht, int ht_dtype,
    int T, int H, int t_seg0, int nt_seg, float scale) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  gdn_chunk_scan_body<NCW, F8, VCONV, 0>(smem, (int)blockIdx.x, (int)blockIdx.y, ws, o, sv, h0, h0_dtype, ht, ht_dtype, T, H, t_seg0,
                                             nt_seg, scale, ScanSync{});
}

// Single launch of the fused call at small grids (the benchmark's streaming step: 2 x 64 pre-pass + 128 scan workgroups on 256
// CUs).  The first blocks are pre-pass workgroups (waves 8-11 leave at once), the rest scan workgroups; both kinds need no more
// than one CU each.  The scan side waits for the pre-pass side (scan_wait_records) without a second launch -- one launch
// boundary (~4.5 us inside the step's graph) and the scan's own start-up (state tile, value tiles, conv of chunks 0 and 1)
// disappear behind the pre-pass.  The host launches this form only when the whole grid can be resident at once (occupancy
// query x CU count: the form's SPEED needs that; its liveness and safety do not, see scan_wait_records).
// With BH % 8 == 0 the pre-pass workgroup of head bh gets a block id = bh (mod 8): the same die (id % 8) as the head's scan
// workgroups, whose L2 then holds the record it is about to be asked for.
constexpr int SINGLE_THREADS = 64 * (2 * 2 + SCAN_NL + SCAN_NV);
constexpr int G_SYNC_BYTES = IVL_GDN_SYNC_BYTES;   // flags: 64 words per head (<= 32 heads) at 0; headdone, kread: 32 words each behind them; err: 4 words
constexpr int SYNC_MAX_HEADS = 32;
constexpr int SYNC_ERR_WORD = SYNC_MAX_HEADS * SYNC_HEAD_WORDS + 64;
static_assert(G_SYNC_BYTES >= 4 * (SYNC_ERR_WORD + 4), "sync area layout");
// MODE 1, SPLIT (room for twice the pre-pass workgroups): blocks [0, BH) are the q sides of chunk 0 (the k sides of chunk 0 wait
// for their `kread` at the very end: lower ids than their waiters), [BH, BH + nt BH) the k sides (the chain the scan waits
// for), then the q sides of chunks 1.., then the scan workgroups.
//
// MODE 2, long calls (unsplit pre-pass): `nprep` PERSISTENT pre-pass workgroups come first and walk the segment's (chunk, head)
// pairs in chunk-major order (pair w = id, id + nprep, ...: the records appear in the order the scan needs them), the 8 BH scan
// workgroups behind them consume the records as they are published (V wave 0 gates every chunk step, three chunks ahead): the
// pre-pass of a 4096-token call does not run in front of the 64 serial chunk steps but beside them.  Every workgroup of the
// launch gets the scan's 156 KB of LDS, i.e. one CU: the host sizes nprep = resident workgroups - 8 BH (at least 8 BH).
// A mode is a template instance of its own: the step-shape kernel (MODE 1) carries none of the long-call code.
// MODE: 0 = small grid, whole pre-pass per chunk; 1 = small grid, split pre-pass; 2 = long call
constexpr int LONG_THREADS = 1024;                 // MODE 2: a pre-pass workgroup = TWO 512-thread bodies (two chunk-head pairs at once)
constexpr int P_BODY_STRIDE = (P_BYTES + 1023) / 1024 * 1024;      // LDS of the second body starts here
static_assert(2 * P_BODY_STRIDE <= scan_lds_bytes(2, false) && 2 * P_BODY_STRIDE <= 160 * 1024, "two pre-pass bodies fit the launch's LDS");
template <bool F8, int MODE>
__global__ __launch_bounds__(MODE == 2 ? LONG_THREADS : SINGLE_THREADS) void gdn_chunk_single_kernel(
    PrepFused pf, unsigned char* __restrict__ ws, bf16_t* __restrict__ o, ScanV sv, const void* h0, int h0_dtype, void* ht,
    int ht_dtype, int T, int H, int BH, int t_seg0, int nt_seg, int nprep, float scale, ScanSync sy) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr bool SPLIT = MODE == 1, LONG = MODE == 2;
  const int nside = nt_seg * BH;                       // (chunk, head) pairs of the segment; nprep = pre-pass workgroups of the launch
  int id = (int)blockIdx.x;
  // MODE 2: the first 2 nsplit BH blocks are STARTERS -- the split pre-pass (a k side and a q side, 512 threads each) of the
  // chunks the scan needs first (sy.nsplit = 5: measured 3 / 4 / 5 / 6 -> 90 / 87.8 / 86.8 / 87.0 us at T = 4096, round 3: 97):
  // they publish after ~7 us where a round of the two-body persistent workgroups takes ~16, and their CUs go to the blocks that
  // were not resident at the start (the grid is larger than the chip by these blocks; the scan workgroups, last in id order,
  // need their CUs only when the first records exist).  ids: q sides of chunk 0 | k sides | q sides of chunks 1.. (as in MODE 1).
  const int nstart = LONG ? 2 * sy.nsplit * BH : 0;
  if (LONG && id < nstart) {
    if (threadIdx.x >= 512) return;
    const int nks = sy.nsplit * BH;
    bool qside = false;
    if (id < BH) { qside = true; }
    else if (id < BH + nks) { id -= BH; }
    else { qside = true; id -= nks; }
    const int ci = id / BH, bh = id % BH;              // chunk-major
    unsigned int* done = sy.flags + bh * SYNC_HEAD_WORDS + ci;
    if (!qside)
      gdn_chunk_prepare_body<F8, true, true, 1>(smem, ci, bh, nullptr, nullptr, nullptr, nullptr, pf, ws, T, H, t_seg0, nt_seg, 1, done, sy.kread + bh, &sy);
    else
      gdn_chunk_prepare_body<F8, true, true, 2>(smem, ci, bh, nullptr, nullptr, nullptr, nullptr, pf, ws, T, H, t_seg0, nt_seg, 1, done, sy.kread + bh, &sy);
    return;
  }
  id -= nstart;
  if (id < nprep) {
    if constexpr (LONG) {
      // Two bodies per workgroup, threads [0, 512) and [512, 1024), each with its own 70 KB of LDS and its own (chunk, head)
      // pair: the launch's LDS size (the scan's 156 KB) admits one workgroup per CU, and ONE body per CU is latency-bound
      // (measured: the scan waited 1.2k of its 2.9k cycles per chunk for the records; two bodies per CU are what the two-launch
      // pre-pass runs at).  Both halves execute the same barrier sequence (same code, equal trip counts: the host keeps the
      // pair count even), so the bodies' workgroup barriers simply span both.
      const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 9);       // (wave-uniform: kept in an SGPR)
      unsigned char* smem_body = smem + half * P_BODY_STRIDE;
      // the front-end arguments are re-read from the kernel-argument segment in every round (scalar loads through a pointer the
      // compiler cannot see through): held in SGPRs across the whole body they exceed the scalar register file
      static_assert(__builtin_offsetof(PrepFused, proj) == 0, "pf is the first kernel argument");
#if defined(__HIP_DEVICE_COMPILE__)
      typedef __attribute__((address_space(4))) const unsigned int kernarg_word;
      kernarg_word* kp = (kernarg_word*)__builtin_amdgcn_kernarg_segment_ptr();
#else
      const unsigned int* kp = (const unsigned int*)&pf;
#endif
      for (int w = sy.nsplit * BH + 2 * id + half; w < nside; w += 2 * nprep) {   // (the publish at the end of the body is a workgroup barrier: LDS is free again)
        asm volatile("" : "+s"(kp));
        PrepFused pfl;
        {
          static_assert(sizeof(PrepFused) % 4 == 0, "copied in dwords");
          unsigned int words[sizeof(PrepFused) / 4];
#pragma unroll
          for (unsigned int i = 0; i < sizeof(PrepFused) / 4; ++i) words[i] = kp[i];
          __builtin_memcpy(&pfl, words, sizeof(PrepFused));
        }
        const int ci = w / BH, bh = w % BH;
        gdn_chunk_prepare_body<F8, true, true, 0, true>(smem_body, ci, bh, nullptr, nullptr, nullptr, nullptr, pfl, ws, T, H, t_seg0, nt_seg, 1,
                                                  sy.flags + bh * SYNC_HEAD_WORDS + ci, nullptr, &sy, half * 512);
      }
    } else {
      if (threadIdx.x >= 512) return;
      bool qside = false;
      if constexpr (SPLIT) {                           // ids: q sides of chunk 0 | k sides | q sides of chunks 1..
        if (id < BH) { qside = true; }
        else if (id < BH + nside) { id -= BH; }
        else { qside = true; id -= nside; }            // -> [BH, nside): chunk >= 1 in the numbering below
      }
      int bh, ci;
      if ((BH & 7) == 0) { bh = (id & 7) + 8 * ((id >> 3) / nt_seg); ci = (id >> 3) % nt_seg; }
      else { bh = id / nt_seg; ci = id % nt_seg; }
      if (SPLIT && qside) {                            // q sides are numbered chunk-major: id = ci BH + bh (chunk 0 first)
        ci = id / BH; bh = id % BH;
      }
      unsigned int* done = sy.flags + bh * SYNC_HEAD_WORDS + ci;
      if constexpr (!SPLIT)
        gdn_chunk_prepare_body<F8, true, true, 0>(smem, ci, bh, nullptr, nullptr, nullptr, nullptr, pf, ws, T, H, t_seg0, nt_seg, 1, done, nullptr, &sy);
      else if (!qside)
        gdn_chunk_prepare_body<F8, true, true, 1>(smem, ci, bh, nullptr, nullptr, nullptr, nullptr, pf, ws, T, H, t_seg0, nt_seg, 1, done, sy.kread + bh, &sy);
      else
        gdn_chunk_prepare_body<F8, true, true, 2>(smem, ci, bh, nullptr, nullptr, nullptr, nullptr, pf, ws, T, H, t_seg0, nt_seg, 1, done, sy.kread + bh, &sy);
    }
  } else {
    if (LONG && threadIdx.x >= SINGLE_THREADS) return;     // (the long-call launch has 1024-thread workgroups: a scan workgroup uses 12 waves)
    id -= nprep;
    // (step shape with the long calls' progressive wait -- chunks 0-2 at the start, chunk 3 gated, chunk-major ids -- measured:
    // 16.4 -> 17.0 us; the gate costs more than the dispatch skew between the chunks gives)
    gdn_chunk_scan_body<2, F8, true, LONG ? 2 : 1>(smem, id % BH, id / BH, ws, o, sv, h0, h0_dtype, ht, ht_dtype, T, H, t_seg0, nt_seg,
                                                   scale, sy);
  }
}

#ifdef IVL_TRACE
int g_scan_ncw = 0;                        // developer knob (trace build only): force 2 or 4 compute waves per scan workgroup
int g_gdn_single = 1;                      // developer knob (trace build only): 0 = two launches always, 2 = never split the pre-pass, 3 = no overlapped long calls
#endif

}  // namespace ivl

using namespace ivl;

static inline int seg_chunks(int NT) { return NT < G_SEG_CHUNKS ? NT : G_SEG_CHUNKS; }

template <int NCW, bool F8, bool VCONV>
static void scan_set_attr() {
  (void)hipFuncSetAttribute((const void*)gdn_chunk_scan_kernel<NCW, F8, VCONV>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            scan_lds_bytes(NCW, F8));
}
static int g_cu_count[64];
static int g_resident[64][2];                       // [device][F8]: workgroups of the single-launch kernels that can be resident at once
// [device]: SYNC_STATUS_SLOTS x two words (code, where) of pinned host memory the kernels report a failed wait in.  A sync area
// reports into ITS OWN slot, so one area's failure refuses further launches on that area, not on every stream and graph of the
// device.  Slots are handed out by registration (with slots assigned by address hash, ivl_gdn_sync_reset of an area B
// cleared the slot of a still-failed area A that hashed to the same slot, and the next call on A was launched and returned IVL_OK
// with incomplete outputs): an area address keeps its slot while it is among the 64 most recently used ones of its device; when
// all slots are taken, a new address takes over the least recently used slot whose status is HEALTHY (a failed area never loses
// its slot; an area that lost a healthy slot simply registers again at its next call).  Only if all 64 slots hold failed areas
// does an address fall back to the hash (shared slot: refused together with its owner).
constexpr int SYNC_STATUS_SLOTS = 64;
static unsigned int* g_host_status[64];
static unsigned int* g_host_status_dev[64];         // ... as the device addresses them
static unsigned long long g_slot_area[64][SYNC_STATUS_SLOTS];      // [device][slot]: the area address that owns the slot (0: free)
static unsigned long long g_slot_used[64][SYNC_STATUS_SLOTS];      // ... and when it was last looked up (a per-process counter)
static unsigned long long g_slot_clock = 0;
static std::mutex g_slot_mutex;
static int device_index();
static inline int status_slot(const void* sync) {
  const unsigned long long a = (unsigned long long)(size_t)sync;
  const int dev = device_index();
  unsigned long long* tab = g_slot_area[dev];
  unsigned long long* used = g_slot_used[dev];
  const unsigned int* hs = g_host_status[dev];
  std::lock_guard<std::mutex> lock(g_slot_mutex);
  const unsigned long long now = ++g_slot_clock;
  int free_slot = -1, lru = -1;
  for (int s = 0; s < SYNC_STATUS_SLOTS; ++s) {
    if (tab[s] == a) {
      used[s] = now;
      return s;
    }
    if (tab[s] == 0ull) {
      if (free_slot < 0) free_slot = s;
    } else if ((hs == nullptr || __atomic_load_n(hs + 2 * s, __ATOMIC_RELAXED) == 0u) && (lru < 0 || used[s] < used[lru])) {
      lru = s;
    }
  }
  const int take = free_slot >= 0 ? free_slot : lru;
  if (take >= 0) {
    tab[take] = a;
    used[take] = now;
    return take;
  }
  return (int)((((unsigned int)(a >> 8)) * 0x9E3779B1u) >> 26);       // every slot holds a failed area: 0 .. 63 by address hash
}
static int device_index() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev & 63;
}
static int device_cu_count() { return g_cu_count[device_index()]; }
// Workgroups of gdn_chunk_single_kernel<F8, .> that the device holds at once: occupancy query x CU count, the smallest over the
// three modes (768 threads and ~156 KB of LDS: one per CU).  ivl_gdn_resident_blocks overrides it (0 = never a single launch:
// the switch back to the two-launch form; a small number stands for a small or partitioned device).
static int g_resident_override = -1;
static int resident_blocks(bool f8) {
  const int ov = __atomic_load_n(&g_resident_override, __ATOMIC_RELAXED);
  if (ov >= 0) return ov;
  return g_resident[device_index()][f8 ? 1 : 0];
}
template <bool F8, int MODE>
static int single_occupancy(int lds, int cus) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)gdn_chunk_single_kernel<F8, MODE>, MODE == 2 ? LONG_THREADS : SINGLE_THREADS,
                                                   (size_t)lds) != hipSuccess) nb = 0;
  return nb * cus;
}
// dynamic-LDS opt-in, occupancy and the host status word, once per device (hipFuncSetAttribute acts on the current device)
static void gdn_chunk_init_device() {
  static std::once_flag once[64];
  const int dev = device_index();
  std::call_once(once[dev], [dev] {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    g_cu_count[dev] = cus;
    const hipFuncAttribute attr = hipFuncAttributeMaxDynamicSharedMemorySize;
    (void)hipFuncSetAttribute((const void*)gdn_chunk_prepare_kernel<false, false>, attr, P_BYTES);
    (void)hipFuncSetAttribute((const void*)gdn_chunk_prepare_kernel<true, false>, attr, P_BYTES);
    (void)hipFuncSetAttribute((const void*)gdn_chunk_prepare_kernel<false, true>, attr, P_BYTES);
    (void)hipFuncSetAttribute((const void*)gdn_chunk_prepare_kernel<true, true>, attr, P_BYTES);
    scan_set_attr<4, false, false>(); scan_set_attr<2, false, false>(); scan_set_attr<4, true, false>(); scan_set_attr<2, true, false>();
    scan_set_attr<4, false, true>(); scan_set_attr<2, false, true>(); scan_set_attr<4, true, true>(); scan_set_attr<2, true, true>();
    const int lds16 = scan_lds_bytes(2, false);
    (void)hipFuncSetAttribute((const void*)gdn_chunk_single_kernel<false, 0>, attr, lds16);
    (void)hipFuncSetAttribute((const void*)gdn_chunk_single_kernel<false, 1>, attr, lds16);
    (void)hipFuncSetAttribute((const void*)gdn_chunk_single_kernel<false, 2>, attr, lds16);
    const int lds8 = scan_lds_bytes(2, true) > P_BYTES ? scan_lds_bytes(2, true) : P_BYTES;
    const int lds8_long = lds8 > 2 * P_BODY_STRIDE ? lds8 : 2 * P_BODY_STRIDE;      // the long-call form runs TWO pre-pass bodies per workgroup
    (void)hipFuncSetAttribute((const void*)gdn_chunk_single_kernel<true, 0>, attr, lds8);
    (void)hipFuncSetAttribute((const void*)gdn_chunk_single_kernel<true, 1>, attr, lds8);
    (void)hipFuncSetAttribute((const void*)gdn_chunk_single_kernel<true, 2>, attr, lds8_long);
    auto min3 = [](int a, int b, int c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); };
    g_resident[dev][0] = min3(single_occupancy<false, 0>(lds16, cus), single_occupancy<false, 1>(lds16, cus), single_occupancy<false, 2>(lds16, cus));
    g_resident[dev][1] = min3(single_occupancy<true, 0>(lds8, cus), single_occupancy<true, 1>(lds8, cus), single_occupancy<true, 2>(lds8_long, cus));
    // the status slots: 512 bytes of pinned, device-mapped host memory for the life of the process (the one thing the library
    // allocates; without it a failed wait is still recorded in the sync area and found by ivl_gdn_sync_status)
    void* hp = nullptr;
    void* dp = nullptr;
    if (hipHostMalloc(&hp, SYNC_STATUS_SLOTS * 8, hipHostMallocMapped) == hipSuccess && hp != nullptr) {
      __builtin_memset(hp, 0, SYNC_STATUS_SLOTS * 8);
      if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
        g_host_status[dev] = (unsigned int*)hp;
        g_host_status_dev[dev] = (unsigned int*)dp;
      } else {
        (void)hipHostFree(hp);
      }
    }
    (void)hipGetLastError();
  });
}
static const char* sync_code_name(unsigned int code) {
  switch (code & 0xffu) {
    case SYNC_E_START: return "scan workgroup: the records it waits for at its start were not published in time";
    case SYNC_E_GATE: return "scan workgroup (long call): a later chunk's record was not published in time";
    case SYNC_E_KREAD: return "pre-pass k side: the q side did not read the old conv state in time";
    default: return "unknown code";
  }
}
// the host-visible status of the current device: 0 = healthy.  sync != NULL: of that area's slot; NULL: the first failed slot
static unsigned int host_status(const void* sync, unsigned int* where) {
  const unsigned int* hs = g_host_status[device_index()];
  if (hs == nullptr) return 0u;
  const int s0 = sync != nullptr ? status_slot(sync) : 0, s1 = sync != nullptr ? s0 + 1 : SYNC_STATUS_SLOTS;
  for (int s = s0; s < s1; ++s) {
    const unsigned int code = __atomic_load_n(hs + 2 * s, __ATOMIC_RELAXED);
    if (code != 0u) {
      if (where != nullptr) *where = __atomic_load_n(hs + 2 * s + 1, __ATOMIC_RELAXED);
      return code;
    }
  }
  return 0u;
}

extern "C" size_t ivl_gdn_chunk_workspace_bytes(int B, int T, int H, int K, int V) {
  if (B <= 0 || T <= 0 || H <= 0 || K != GK || V != GV) return 0;
  const int NT = (T + GC - 1) / GC;
  size_t bytes = (size_t)B * H * seg_chunks(NT) * Rec<false>::STRIDE;          // the fp8 records are smaller
  if (NT > G_SEG_CHUNKS) bytes += (size_t)B * H * GK * GV * sizeof(float);   // fp32 state carried between segments
  return bytes;
}

template <int NCW, bool F8, bool VCONV>
static void scan_launch(int B, int H, hipStream_t st, const unsigned char* wsb, bf16_t* o, const ScanV& sv, const void* hin, int hin_dt,
                        void* hout, int hout_dt, int T, int t_seg0, int nseg, float scale) {
  hipLaunchKernelGGL((gdn_chunk_scan_kernel<NCW, F8, VCONV>), dim3(B * H, 16 / NCW), dim3(64 * (2 * NCW + SCAN_NL + SCAN_NV)), scan_lds_bytes(NCW, F8),
                     st, wsb, o, sv, hin, hin_dt, hout, hout_dt, T, H, t_seg0, nseg, scale);
}

template <bool F8>
static int gdn_chunk_launch(const void* q, const void* k, const void* v, const float* g, const void* beta, const PrepFused* pf, void* o,
                            const void* h0, int h0_dtype, void* ht, int ht_dtype, int B, int T, int H, float scale,
                            int use_qk_l2norm, unsigned char* wsb, unsigned int* sync, hipStream_t st) {
  const int NT = (T + GC - 1) / GC;
  const int segc = seg_chunks(NT);
  float* carry = NT > G_SEG_CHUNKS ? (float*)(wsb + (size_t)B * H * segc * Rec<false>::STRIDE) : nullptr;
  int ncw = B * H * 4 <= 128 ? 2 : 4;          // 32-column workgroups while 64-column ones would leave half the CUs idle
#ifdef IVL_TRACE
  if (g_scan_ncw) ncw = g_scan_ncw;
#endif
  // the value side of the scan: the raw projection + conv taps + conv state (fused front end) or the convolved v tensor
  ScanV sv;
  if (pf != nullptr) {
    sv.v = pf->proj; sv.ld = pf->ld; sv.col0 = pf->col_v; sv.w = pf->w[2]; sv.st_in = pf->st_in[2]; sv.st_out = pf->st_out[2];
  } else {
    sv.v = (const bf16_t*)v; sv.ld = (long long)H * GV; sv.col0 = 0; sv.w = nullptr; sv.st_in = nullptr; sv.st_out = nullptr;
  }
  const int BH = B * H;
  // Single-launch forms: the caller handed over a healthy sync area and the grid can be resident at once (the forms' speed needs
  // that; IVL_GDN_RESIDENT_BLOCKS = 0 switches them off).  NT <= 63: one poll lane per chunk plus the error-word lane.
  const int resident = sync != nullptr ? resident_blocks(F8) : 0;
  const bool can_sync = pf != nullptr && sync != nullptr && ncw == 2 && BH <= SYNC_MAX_HEADS;
  bool single = can_sync && NT <= 63 && NT * BH + (16 / 2) * BH <= resident;
  // long calls: one launch per segment, persistent pre-pass workgroups beside the scan workgroups -- at least as many of them
  // (an even number of (chunk, head) pairs behind the starters of every segment: a persistent pre-pass workgroup runs two bodies
  // side by side; with an odd B*H that needs segments of an odd number of chunks > 3, which 64-chunk segments are not)
  bool overlap = can_sync && !single && 2 * 8 * BH <= resident && BH % 2 == 0;
#ifdef IVL_TRACE
  single = single && g_gdn_single != 0;
  overlap = overlap && g_gdn_single != 0 && g_gdn_single != 3;
#endif
  ScanSync sy;
  if (can_sync) {
    sy.flags = sync; sy.headdone = sync + SYNC_MAX_HEADS * SYNC_HEAD_WORDS; sy.kread = sy.headdone + 32; sy.err = sync + SYNC_ERR_WORD;
    sy.host_err = g_host_status_dev[device_index()] != nullptr ? g_host_status_dev[device_index()] + 2 * status_slot(sync) : nullptr;
    sy.BH = BH; sy.nsplit = 0;
  }
  const int lds1 = scan_lds_bytes(2, F8) > P_BYTES ? scan_lds_bytes(2, F8) : P_BYTES;
  if (single) {
    bool split = 2 * NT * BH + 8 * BH <= resident;
#ifdef IVL_TRACE
    split = split && g_gdn_single != 2;
#endif
    sy.nsplit = split ? SYNC_HEAD_WORDS : 0;
    if (split)
      hipLaunchKernelGGL((gdn_chunk_single_kernel<F8, 1>), dim3(2 * NT * BH + 8 * BH), dim3(SINGLE_THREADS), lds1, st, *pf, wsb,
                         (bf16_t*)o, sv, h0, h0_dtype, ht, ht_dtype, T, H, BH, 0, NT, 2 * NT * BH, scale, sy);
    else
      hipLaunchKernelGGL((gdn_chunk_single_kernel<F8, 0>), dim3(NT * BH + 8 * BH), dim3(SINGLE_THREADS), lds1, st, *pf, wsb,
                         (bf16_t*)o, sv, h0, h0_dtype, ht, ht_dtype, T, H, BH, 0, NT, NT * BH, scale, sy);
    return check_launch("ivl_gdn_chunk_fused_fwd(single launch)");
  }
  for (int c0 = 0; c0 < NT; c0 += segc) {
    const int nseg = (NT - c0) < segc ? (NT - c0) : segc;
    const bool first = c0 == 0, last = c0 + nseg >= NT;
    if (overlap) {
      const void* hin = first ? h0 : (const void*)carry;
      const int hin_dt = first ? h0_dtype : IVL_F32;
      void* hout = last ? ht : (void*)carry;
      const int hout_dt = last ? ht_dtype : IVL_F32;
      constexpr int NSTART = 5;                                    // (3 / 4 / 5 / 6 measured: 90.0 / 87.8 / 86.8 / 87.0 us at T = 4096)
      sy.nsplit = nseg < NSTART ? nseg : NSTART;                   // starter chunks: at least what the scan's loaders request before the first chunk step
      int nprep = resident - 8 * BH;                               // persistent pre-pass workgroups: what the chip holds beside the scan
      if (nprep > (nseg - sy.nsplit) * BH / 2) nprep = (nseg - sy.nsplit) * BH / 2;   // (two chunk-head pairs per workgroup and round)
      // two pre-pass bodies per workgroup: 2 x 70 KB whatever the scan's own LDS need is (the fp8 scan images are half the bf16 ones:
      // until round 5 an fp8 long call ran its second body outside the launch's LDS allocation)
      const int lds2 = lds1 > 2 * P_BODY_STRIDE ? lds1 : 2 * P_BODY_STRIDE;
      hipLaunchKernelGGL((gdn_chunk_single_kernel<F8, 2>), dim3(2 * sy.nsplit * BH + nprep + 8 * BH), dim3(LONG_THREADS), lds2, st, *pf, wsb,
                         (bf16_t*)o, sv, hin, hin_dt, hout, hout_dt, T, H, BH, c0 * GC, nseg, nprep, scale, sy);
      int rc = check_launch("ivl_gdn_chunk_fused_fwd(overlapped launch)");
      if (rc != IVL_OK) return rc;
      continue;
    }
    if (pf != nullptr)
      hipLaunchKernelGGL((gdn_chunk_prepare_kernel<F8, true>), dim3(nseg, B * H), dim3(512), P_BYTES, st, (const bf16_t*)nullptr,
                         (const bf16_t*)nullptr, (const float*)nullptr, (const bf16_t*)nullptr, *pf, wsb, T, H, c0 * GC, nseg,
                         use_qk_l2norm);
    else
      hipLaunchKernelGGL((gdn_chunk_prepare_kernel<F8, false>), dim3(nseg, B * H), dim3(512), P_BYTES, st, (const bf16_t*)q,
                         (const bf16_t*)k, g, (const bf16_t*)beta, PrepFused{}, wsb, T, H, c0 * GC, nseg, use_qk_l2norm);
    int rc = check_launch("ivl_gdn_chunk_fwd(prepare)");
    if (rc != IVL_OK) return rc;
    const void* hin = first ? h0 : (const void*)carry;
    const int hin_dt = first ? h0_dtype : IVL_F32;
    void* hout = last ? ht : (void*)carry;
    const int hout_dt = last ? ht_dtype : IVL_F32;
    if (pf != nullptr) {
      if (ncw == 2) scan_launch<2, F8, true>(B, H, st, wsb, (bf16_t*)o, sv, hin, hin_dt, hout, hout_dt, T, c0 * GC, nseg, scale);
      else scan_launch<4, F8, true>(B, H, st, wsb, (bf16_t*)o, sv, hin, hin_dt, hout, hout_dt, T, c0 * GC, nseg, scale);
    } else {
      if (ncw == 2) scan_launch<2, F8, false>(B, H, st, wsb, (bf16_t*)o, sv, hin, hin_dt, hout, hout_dt, T, c0 * GC, nseg, scale);
      else scan_launch<4, F8, false>(B, H, st, wsb, (bf16_t*)o, sv, hin, hin_dt, hout, hout_dt, T, c0 * GC, nseg, scale);
    }
    rc = check_launch("ivl_gdn_chunk_fwd(scan)");
    if (rc != IVL_OK) return rc;
  }
  return IVL_OK;
}

extern "C" int ivl_gdn_chunk_fwd(const void* q, const void* k, const void* v, const float* g, const void* beta,
                                 void* o, const void* h0, int h0_dtype, void* ht, int ht_dtype,
                                 int B, int T, int H, int K, int V, float scale, int use_qk_l2norm, int mma_dtype,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  IVL_REQUIRE(q && k && v && g && beta && o, IVL_ERR_INVALID_ARG, "ivl_gdn_chunk_fwd: NULL pointer");
  IVL_REQUIRE(B > 0 && T > 0 && H > 0, IVL_ERR_INVALID_ARG, "ivl_gdn_chunk_fwd: B,T,H must be positive (%d,%d,%d)", B, T, H);
  IVL_REQUIRE(K == GK && V == GV, IVL_ERR_UNSUPPORTED, "ivl_gdn_chunk_fwd: built for K=128,V=256 (got %d,%d)", K, V);
  IVL_REQUIRE((h0 == nullptr || h0_dtype == IVL_F32 || h0_dtype == IVL_BF16) &&
              (ht == nullptr || ht_dtype == IVL_F32 || ht_dtype == IVL_BF16),
              IVL_ERR_INVALID_ARG, "ivl_gdn_chunk_fwd: state dtype must be IVL_F32 or IVL_BF16");
  IVL_REQUIRE(mma_dtype == IVL_BF16 || mma_dtype == IVL_FP8_E4M3, IVL_ERR_INVALID_ARG,
              "ivl_gdn_chunk_fwd: mma_dtype must be IVL_BF16 or IVL_FP8_E4M3 (got %d)", mma_dtype);
  const size_t need = ivl_gdn_chunk_workspace_bytes(B, T, H, K, V);
  IVL_REQUIRE(workspace != nullptr && workspace_bytes >= need, IVL_ERR_WORKSPACE,
              "ivl_gdn_chunk_fwd: workspace %zu bytes < required %zu", workspace_bytes, need);
  gdn_chunk_init_device();
  if (mma_dtype == IVL_FP8_E4M3)
    return gdn_chunk_launch<true>(q, k, v, g, beta, nullptr, o, h0, h0_dtype, ht, ht_dtype, B, T, H, scale, use_qk_l2norm,
                                  (unsigned char*)workspace, nullptr, (hipStream_t)stream);
  return gdn_chunk_launch<false>(q, k, v, g, beta, nullptr, o, h0, h0_dtype, ht, ht_dtype, B, T, H, scale, use_qk_l2norm,
                                 (unsigned char*)workspace, nullptr, (hipStream_t)stream);
}

extern "C" int ivl_gdn_chunk_fused_fwd(const void* proj, int64_t ld, int col_q, int col_k, int col_v, int col_a, int col_b,
                                       const void* wq, const void* wk, const void* wv, const void* sq_in, const void* sk_in,
                                       const void* sv_in, void* sq_out, void* sk_out, void* sv_out, const float* A_log,
                                       const float* dt_bias, void* o, const void* h0, int h0_dtype, void* ht, int ht_dtype, int B,
                                       int T, int H, int K, int V, int conv_width, float scale, int mma_dtype, void* workspace,
                                       size_t workspace_bytes, void* sync, void* stream) {
  IVL_REQUIRE(proj && wq && wk && wv && A_log && dt_bias && o, IVL_ERR_INVALID_ARG, "ivl_gdn_chunk_fused_fwd: NULL pointer");
  IVL_REQUIRE(B > 0 && T > 0 && H > 0, IVL_ERR_INVALID_ARG, "ivl_gdn_chunk_fused_fwd: B,T,H must be positive (%d,%d,%d)", B, T, H);
  IVL_REQUIRE(K == GK && V == GV && conv_width == 4, IVL_ERR_UNSUPPORTED,
              "ivl_gdn_chunk_fused_fwd: built for K=128, V=256, conv width 4 (got %d,%d,%d)", K, V, conv_width);
  IVL_REQUIRE((long long)T * ld < (1LL << 31), IVL_ERR_UNSUPPORTED,
              "ivl_gdn_chunk_fused_fwd: T * ld = %lld elements exceeds the 32-bit row addressing of the pre-pass (split the call)",
              (long long)T * ld);
  IVL_REQUIRE(ld % 8 == 0 && col_q % 8 == 0 && col_k % 8 == 0 && col_v % 8 == 0 && col_q >= 0 && col_k >= 0 && col_v >= 0 &&
                  col_a >= 0 && col_b >= 0 && col_a + H <= ld && col_b + H <= ld,
              IVL_ERR_INVALID_ARG, "ivl_gdn_chunk_fused_fwd: projection columns must be 16-byte aligned and inside the row (ld=%lld)",
              (long long)ld);
  IVL_REQUIRE((h0 == nullptr || h0_dtype == IVL_F32 || h0_dtype == IVL_BF16) &&
              (ht == nullptr || ht_dtype == IVL_F32 || ht_dtype == IVL_BF16),
              IVL_ERR_INVALID_ARG, "ivl_gdn_chunk_fused_fwd: state dtype must be IVL_F32 or IVL_BF16");
  IVL_REQUIRE(mma_dtype == IVL_BF16 || mma_dtype == IVL_FP8_E4M3, IVL_ERR_INVALID_ARG,
              "ivl_gdn_chunk_fused_fwd: mma_dtype must be IVL_BF16 or IVL_FP8_E4M3 (got %d)", mma_dtype);
  const size_t need = ivl_gdn_chunk_workspace_bytes(B, T, H, K, V);
  IVL_REQUIRE(workspace != nullptr && workspace_bytes >= need, IVL_ERR_WORKSPACE,
              "ivl_gdn_chunk_fused_fwd: workspace %zu bytes < required %zu", workspace_bytes, need);
  IVL_REQUIRE(sync == nullptr || ((size_t)sync & 15) == 0, IVL_ERR_INVALID_ARG, "ivl_gdn_chunk_fused_fwd: sync area must be 16-byte aligned");
  gdn_chunk_init_device();
  if (sync != nullptr) {
    unsigned int where = 0;
    const unsigned int code = host_status(sync, &where);
    IVL_REQUIRE(code == 0u, IVL_ERR_SYNC,
                "ivl_gdn_chunk_fused_fwd: an earlier single-launch call on this sync area failed (code %u: %s; head %u, chunk %u): its "
                "outputs and states are incomplete. ivl_gdn_sync_reset() re-arms the sync area",
                code, sync_code_name(code), where >> 16, where & 0xffffu);
  }
  PrepFused pf;
  pf.proj = (const bf16_t*)proj; pf.ld = ld;
  pf.col_q = col_q; pf.col_k = col_k; pf.col_v = col_v; pf.col_a = col_a; pf.col_b = col_b;
  pf.w[0] = (const bf16_t*)wq; pf.w[1] = (const bf16_t*)wk; pf.w[2] = (const bf16_t*)wv;
  pf.st_in[0] = (const bf16_t*)sq_in; pf.st_in[1] = (const bf16_t*)sk_in; pf.st_in[2] = (const bf16_t*)sv_in;
  pf.st_out[0] = (bf16_t*)sq_out; pf.st_out[1] = (bf16_t*)sk_out; pf.st_out[2] = (bf16_t*)sv_out;
  pf.A_log = A_log; pf.dt_bias = dt_bias;
  if (mma_dtype == IVL_FP8_E4M3)
    return gdn_chunk_launch<true>(nullptr, nullptr, nullptr, nullptr, nullptr, &pf, o, h0, h0_dtype, ht, ht_dtype, B, T, H, scale, 1,
                                  (unsigned char*)workspace, (unsigned int*)sync, (hipStream_t)stream);
  return gdn_chunk_launch<false>(nullptr, nullptr, nullptr, nullptr, nullptr, &pf, o, h0, h0_dtype, ht, ht_dtype, B, T, H, scale, 1,
                                 (unsigned char*)workspace, (unsigned int*)sync, (hipStream_t)stream);
}

// Status of the single-launch forms on the current device: IVL_OK, or IVL_ERR_SYNC when a wait inside a launch ran out (the
// kernels report it through a pinned host word: no stream work, no synchronisation -- what has been reported so far).
// `sync` != NULL additionally reads the area's own error word from the device (a blocking copy behind `stream`: never during
// capture), which also covers a device whose host word could not be allocated.
extern "C" int ivl_gdn_sync_status(const void* sync, void* stream) {
  gdn_chunk_init_device();
  unsigned int where = 0;
  unsigned int code = host_status(sync, &where);
  if (code == 0u && sync != nullptr) {
    unsigned int w[2] = {0u, 0u};
    hipError_t e = hipMemcpyAsync(w, (const unsigned int*)sync + SYNC_ERR_WORD, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    IVL_REQUIRE(e == hipSuccess, IVL_ERR_LAUNCH, "ivl_gdn_sync_status: %s", hipGetErrorString(e));
    code = w[0]; where = w[1];
  }
  IVL_REQUIRE(code == 0u, IVL_ERR_SYNC, "ivl_gdn_sync_status: a wait inside a single-launch call ran out (code %u: %s; head %u, chunk %u)",
              code, sync_code_name(code), where >> 16, where & 0xffffu);
  return IVL_OK;
}

// Re-arm: zero the area (behind `stream`) and the area's host status slot.  Also the way to initialise a fresh area.
extern "C" int ivl_gdn_sync_reset(void* sync, void* stream) {
  IVL_REQUIRE(sync != nullptr && ((size_t)sync & 15) == 0, IVL_ERR_INVALID_ARG, "ivl_gdn_sync_reset: sync area must be a 16-byte aligned device pointer");
  gdn_chunk_init_device();
  const hipError_t e = hipMemsetAsync(sync, 0, G_SYNC_BYTES, (hipStream_t)stream);
  IVL_REQUIRE(e == hipSuccess, IVL_ERR_LAUNCH, "ivl_gdn_sync_reset: %s", hipGetErrorString(e));
  unsigned int* hs = g_host_status[device_index()];
  if (hs != nullptr) {
    hs += 2 * status_slot(sync);
    __atomic_store_n(hs + 1, 0u, __ATOMIC_RELAXED);
    __atomic_store_n(hs, 0u, __ATOMIC_RELAXED);
  }
  return IVL_OK;
}

// How many workgroups of the single-launch kernels the library takes to be resident at once on the current device (what gates
// the single-launch forms).  override >= 0 replaces the occupancy-derived number process-wide (0: always two launches),
// override < 0 restores it, IVL_GDN_RESIDENT_QUERY changes nothing; returns the number in force for bf16 operands.
extern "C" int ivl_gdn_resident_blocks(int override_blocks) {
  gdn_chunk_init_device();
  if (override_blocks != IVL_GDN_RESIDENT_QUERY)
    __atomic_store_n(&g_resident_override, override_blocks < 0 ? -1 : override_blocks, __ATOMIC_RELAXED);
  return resident_blocks(false);
}
