/*
 * ivl_hip.h -- C ABI of libivl_hip.so: MI355X (gfx950 / CDNA4) kernels for the
 * InfiniteVL hybrid-attention hot path (Gated DeltaNet + sliding-window attention, and the
 * vision tower's window attention).
 *
 * This is the drop-in boundary.  Each entry point replaces one
 * operator the reference reaches through a third-party CUDA/Triton package; the
 * reference call site it serves is cited as `std:<line>` =
 * infinitevl/infinitevl_standard/modeling_infinitevl.py, `strm:` = the same file of
 * infinitevl/infinitevl_streaming/, and `fla:` =
 * src/llamafactory/model/fla/ (vendored snapshot of flash-linear-attention).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller;
 *     the library never allocates, frees, synchronises or branches on device data on the
 *     host, so every call is hipGraph-capturable (SURVEY.md section 8b "Threading / streams").
 *     (One exception, ownership: the first GDN chunk call on a device makes ONE hipHostMalloc of 512 bytes of pinned,
 *     device-mapped host memory -- 64 status slots of two words for ivl_gdn_sync_status -- owned by the library and kept
 *     for the life of the process; if it fails the library runs without it.  ivl_gdn_sync_status(sync != NULL) is the
 *     one blocking call.)
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - layouts are the reference's time-major ones: q,k [B,T,H,K], v,o [B,T,H,V],
 *     g,beta [B,T,H], recurrent state [B,H,K,V]; bf16 activations, fp32 g.
 *   - return value: 0 = ok, negative = error code below; never throws, never exits.
 *     ivl_last_error() returns a thread-local human-readable message for the last failure.
 */
#ifndef IVL_HIP_H
#define IVL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVL_ABI_VERSION 11

/* The library is built with -fvisibility=hidden: the entry points declared here are its ONLY exported symbols. */
#define IVL_API __attribute__((visibility("default")))

/* element type codes for the arguments that accept more than one */
#define IVL_BF16 0
#define IVL_F32 2
#define IVL_FP8_E4M3 3   /* OCP e4m3fn (gfx950); accepted only as `mma_dtype`: operand format of the MFMA products */

/* error codes */
#define IVL_OK 0
#define IVL_ERR_INVALID_ARG (-1)   /* NULL pointer / non-positive size / bad dtype code        */
#define IVL_ERR_UNSUPPORTED (-2)   /* shape outside what the kernels are built for             */
#define IVL_ERR_WORKSPACE (-3)     /* workspace too small (see the *_workspace_bytes helpers)  */
#define IVL_ERR_LAUNCH (-4)        /* hipLaunch / hipGetLastError failure                      */
#define IVL_ERR_SYNC (-5)          /* a wait inside a single-launch GDN call ran out (ivl_gdn_sync_status / _reset) */

IVL_API int ivl_abi_version(void);
IVL_API const char* ivl_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Gated DeltaNet, token-recurrent form.
 * Replaces fla.ops.gated_delta_rule.fused_recurrent_gated_delta_rule
 *   (call site std:1309-1320; kernel fla:ops/gated_delta_rule/fused_recurrent.py:21-112).
 * Per token: S *= exp(g); d = beta*(v - S^T k); S += k d^T; o = S^T (q*scale); l2norm(q), l2norm(k)
 * first when use_qk_l2norm != 0 (eps 1e-6, result rounded to bf16 like fla's l2norm_fwd).
 * h0 (may be NULL = zeros) and ht (may be NULL = not stored) are [B,H,K,V] in h0_dtype/ht_dtype
 * (IVL_F32 or IVL_BF16); ht may alias h0 (in-place state update: each element is read once, then
 * written once by the same thread).  Requires K == 128, V % 64 == 0.
 * ------------------------------------------------------------------------------------------- */
IVL_API int ivl_gdn_recurrent_fwd(const void* q, const void* k, const void* v, const float* g, const void* beta,
                          void* o, const void* h0, int h0_dtype, void* ht, int ht_dtype,
                          int B, int T, int H, int K, int V, float scale, int use_qk_l2norm, void* stream);
/* The same rule on IEEE-half activations (q, k, v, beta, o fp16; g fp32; the state in its own dtype): fla's operators take fp16 as
 * well as bf16 (fla:ops/gated_delta_rule/chunk.py:352 refuses fp32 only).  The package serves BOTH GDN operators with it when
 * handed fp16 tensors -- fp32 arithmetic on the fp16 inputs, token by token, whatever T: the function the chunkwise form
 * evaluates, without that form's intermediate roundings (the MFMA kernels of the chunk path are bf16 / e4m3 only). */
IVL_API int ivl_gdn_recurrent_f16_fwd(const void* q, const void* k, const void* v, const float* g, const void* beta,
                              void* o, const void* h0, int h0_dtype, void* ht, int ht_dtype,
                              int B, int T, int H, int K, int V, float scale, int use_qk_l2norm, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gated DeltaNet, chunkwise form (chunk = 64 tokens).
 * Replaces fla.ops.gated_delta_rule.chunk_gated_delta_rule (call site std:1297-1308;
 *   fla:ops/gated_delta_rule/chunk.py:18-71 = l2norm + cumsum + WY transform + state scan + output,
 *   kernels K1-K6 of SURVEY.md section 2.1) with two launches: a chunk-parallel pre-pass and a fused
 *   state-scan + output pass (the per-chunk state snapshots never reach HBM).
 * `workspace` holds the pre-pass results; size from ivl_gdn_chunk_workspace_bytes.
 * Requires K == 128, V == 256 (the InfiniteVL head shape), any T >= 1 (zero-padded last chunk).
 * mma_dtype = IVL_BF16: the reference's precision (bf16 operands at the reference's rounding points, fp32 accumulation).
 * mma_dtype = IVL_FP8_E4M3 (BASELINE.json configs[4]; the reference has no such path): the four products of the serial
 *   pass (w S, q S, k^T v_new, A v_new) take e4m3 operands -- w e^gamma, q_hat, k_hat e^{gl-gamma}, A, the state and v_new
 *   are rounded to e4m3 (clamped to +-448) -- with fp32 accumulators, fp32 carried state and bf16 u / o: half the LDS and
 *   L2 operand traffic of the scan.  Tolerance: tests/test_gpu_parity.py (fp8 section).
 * ------------------------------------------------------------------------------------------- */
IVL_API size_t ivl_gdn_chunk_workspace_bytes(int B, int T, int H, int K, int V);
IVL_API int ivl_gdn_chunk_fwd(const void* q, const void* k, const void* v, const float* g, const void* beta,
                      void* o, const void* h0, int h0_dtype, void* ht, int ht_dtype,
                      int B, int T, int H, int K, int V, float scale, int use_qk_l2norm, int mma_dtype,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Chunked gated delta rule WITH its front end, from the mixer's single projection buffer: the three causal short
 * convolutions (+SiLU, carry-in from / carry-out to the conv states) and the gate math run inside the chunk-parallel
 * pre-pass, so q / k / v / g / beta never reach HBM and ivl_gdn_prologue_fwd's launch disappears.
 * Replaces std:1253-1283 (q/k/v_conv1d), std:1293-1294 (beta, g) and std:1299-1323 (chunk_gated_delta_rule with
 *   use_qk_l2norm_in_kernel=True) of Qwen3NextGatedDeltaNet.forward in one call; bit-identical to ivl_gdn_prologue_fwd
 *   followed by ivl_gdn_chunk_fwd (same arithmetic, same bf16 rounding points).
 * proj bf16 [B*T, ld]; col_q / col_k / col_v = first column of head 0 of the q / k / v blocks ([H*K], [H*K], [H*V]),
 * col_a / col_b = first column of the a / b gate inputs ([H]); conv taps wq / wk / wv bf16 [D, 1, 4]; conv states
 * [B, D, 4] bf16 (in: NULL = zero history; out: NULL = not wanted; out may alias in).  The rest as ivl_gdn_chunk_fwd
 * (q/k l2norm always on, as the reference calls it).
 * `sync` (optional, NULL = always two launches): IVL_GDN_SYNC_BYTES of device memory, 16-byte aligned, that the CALLER
 *   zeroed (hipMemset or ivl_gdn_sync_reset) before its first use and that nothing but this library has written since; ONE
 *   AREA PER STREAM of concurrent calls.  With it the call runs as ONE launch per 4096-token segment whenever the grid can
 *   be resident at once (occupancy query x CU count; ivl_gdn_resident_blocks overrides it, 0 = always two
 *   launches) and B*H <= 32:
 *     - B*H*(T/64 + 8) workgroups fit (the 256-token streaming step): pre-pass and scan workgroups side by side, the scan
 *       waiting on flags in this area; with room for twice the pre-pass workgroups the pre-pass is split in a k and a q side;
 *     - longer calls, when the device holds 2 * 8*B*H workgroups: persistent pre-pass workgroups in front of the scan
 *       workgroups, the scan consuming the records chunk by chunk as they are published.
 *   The launch clears its flags again (all-zero between launches: replayable from a hipGraph).  Results are bit-identical to
 *   the two-launch form.  Contract of the in-launch waits (csrc/gdn_chunk.hip, scan_wait_records): a workgroup only waits for
 *   workgroups with lower block ids; every wait is bounded; a wait that runs out raises a sticky error word in the area and in
 *   a host-visible status word, and the waiting workgroup stores NO output / state computed from records it has not seen.
 *   After that this entry point returns IVL_ERR_SYNC for every call with THAT sync area (and launches nothing) until
 *   ivl_gdn_sync_reset of the area; calls with other areas -- other streams, other graphs -- go on (an area reports into
 *   its own host status slot: the 64 most recently used area addresses of a device own a slot each, and a failed area never loses
 *   its slot); workgroups of launches already queued (a hipGraph) stop at once when they find
 *   the area failed.
 * ------------------------------------------------------------------------------------------- */
#define IVL_GDN_SYNC_BYTES 16384
IVL_API int ivl_gdn_chunk_fused_fwd(const void* proj, int64_t ld, int col_q, int col_k, int col_v, int col_a, int col_b,
                            const void* wq, const void* wk, const void* wv, const void* sq_in, const void* sk_in,
                            const void* sv_in, void* sq_out, void* sk_out, void* sv_out, const float* A_log,
                            const float* dt_bias, void* o, const void* h0, int h0_dtype, void* ht, int ht_dtype, int B,
                            int T, int H, int K, int V, int conv_width, float scale, int mma_dtype, void* workspace,
                            size_t workspace_bytes, void* sync, void* stream);
/* Status of the single-launch forms on the CURRENT device: IVL_OK or IVL_ERR_SYNC (ivl_last_error(): which wait, head, chunk).
 * sync == NULL: what the kernels of ANY area have reported so far through the host-visible status slots -- no stream work,
 *   callable at any time, also during capture (a failure shows up once the failing kernel has run: check after a
 *   synchronisation point).
 * sync != NULL: that area's slot, and additionally the area's own error word copied back behind `stream` (blocking; not capturable).
 * ivl_gdn_sync_reset zeroes the area behind `stream` and clears the area's status slot (also the way to initialise an area). */
IVL_API int ivl_gdn_sync_status(const void* sync, void* stream);
IVL_API int ivl_gdn_sync_reset(void* sync, void* stream);
/* The number of single-launch workgroups the library takes to be resident at once on the current device (occupancy query x CU
 * count: 256 on a whole MI355X).  override_blocks >= 0 replaces it process-wide -- 0 = always the two-launch form, a small
 * number = a partitioned / shared device; < 0 restores the query; IVL_GDN_RESIDENT_QUERY changes nothing (a pure read).  Returns
 * the number in force. */
#define IVL_GDN_RESIDENT_QUERY (-2147483647 - 1)
IVL_API int ivl_gdn_resident_blocks(int override_blocks);

/* ---------------------------------------------------------------------------------------------
 * Gate math: beta = sigmoid(b) (bf16), g = -exp(A_log) * softplus(a + dt_bias) (fp32).
 * Replaces the torch glue at std:1293-1294.  a,b bf16 [rows,H] (a_proj / b_proj outputs);
 * A_log, dt_bias fp32 [H].
 * ------------------------------------------------------------------------------------------- */
IVL_API int ivl_gdn_gate_fwd(const void* a, const void* b, const float* A_log, const float* dt_bias,
                     float* g, void* beta, int rows, int H, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Causal depthwise short convolution (+ optional SiLU) with state carry-in.
 * Replaces fla.modules.ShortConvolution.forward / .step (call sites std:1263-1280;
 *   fla:modules/convolution.py:195-293; CUDA ext causal_conv1d_fn / causal_conv1d_update).
 * x,y bf16 [B,T,D]; weight bf16 [D,W] (W == 4); state [B,D,W] bf16 = last W raw inputs, newest
 * last.  state_in == NULL means zero history; state_out == NULL means "do not store";
 * state_out may alias state_in.  T == 1 is the decode step.  D % 8 == 0.
 * ------------------------------------------------------------------------------------------- */
IVL_API int ivl_short_conv_fwd(const void* x, const void* weight, const void* state_in, void* y, void* state_out,
                       int B, int T, int D, int W, int apply_silu, void* stream);
/* The same with fla's `bias=True` option (fla:modules/convolution.py:128-160; InfiniteVL uses conv_bias = False): bias bf16 [D]
 * (NULL = none); y = act(bias + sum_j w[.,j] x[t - W + 1 + j]), the accumulator starting from the bias as in causal_conv1d. */
IVL_API int ivl_short_conv_bias_fwd(const void* x, const void* weight, const void* bias, const void* state_in, void* y, void* state_out,
                            int B, int T, int D, int W, int apply_silu, void* stream);

/* ---------------------------------------------------------------------------------------------
 * y = rmsnorm(x) * weight * gate * sigmoid(gate), rows of N == 256, statistics in fp32.
 * Replaces fla.modules.FusedRMSNormGated.forward (call site std:1338;
 *   fla:modules/fused_norm_gate.py:27-95).  x, gate, y bf16 [rows,N]; weight bf16 [N].
 * gate == NULL: y = rmsnorm(x) * weight with fla's rounding (x * rstd * w in fp32, one rounding at the store) =
 *   fla.modules.RMSNorm.forward (fla:modules/layernorm.py:103-143), the mixer's output norm when use_gate=False (std:1213).
 * ------------------------------------------------------------------------------------------- */
IVL_API int ivl_rmsnorm_swish_gate_fwd(const void* x, const void* gate, const void* weight, void* y,
                               int rows, int N, float eps, void* stream);
/* The same with fla's residual= / prenorm= / residual_in_fp32= options (fla:modules/fused_norm_gate.py:54-60, 98-155; not used by
 * InfiniteVL): the row is x + residual in fp32 (residual [rows,N] IVL_BF16 or IVL_F32; NULL = none), written to residual_out
 * ([rows,N] IVL_BF16 or IVL_F32; NULL = not wanted); statistics and output are those of the unrounded fp32 row. */
IVL_API int ivl_rmsnorm_swish_gate_res_fwd(const void* x, const void* gate, const void* weight, const void* residual, int residual_dtype,
                                   void* residual_out, int residual_out_dtype, void* y, int rows, int N, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multimodal rotary embedding applied in place to q [B,T,Hq,d] and k [B,T,Hkv,d] (bf16,
 * contiguous time-major, i.e. the projection outputs before the reference's transpose).
 * Replaces apply_multimodal_rotary_pos_emb (std:949-984, call site std:1057-1064).
 * cos,sin bf16 [3,B,T,d] as produced by InfiniteVLRotaryEmbedding (std:918-930); sections s0,s1,s2
 * (sum == d/2) pick the t/h/w table per channel block.  Products and the sum are each rounded to
 * bf16 like the reference's eager bf16 arithmetic (bit-identical result).
 * ------------------------------------------------------------------------------------------- */
IVL_API int ivl_mrope_fwd(void* q, void* k, const void* cos, const void* sin,
                  int B, int T, int Hq, int Hkv, int d, int s0, int s1, int s2, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sliding-window attention over (ring-buffer cache ++ new tokens), GQA, d == 128.
 * Replaces ALL_ATTENTION_FUNCTIONS["flash_attention_2"] (call site std:1092-1108) together with
 * the torch.cat cache maintenance of StaticSlidingWindowLayerPrealloc.update (std:126-173).
 *
 * Query row i of the call (absolute position pos+i) sees call-local keys
 *     lo(i) = max(0, n_prev + i - window + 1) .. hi(i) = n_prev + i          (inclusive),
 * where n_prev = min(cache_capacity, pos) cached keys precede the T new ones.
 * Cached token with absolute position p lives in ring slot p % cache_capacity.
 *
 * All strides are in ELEMENTS.  k_new/v_new hold `T_new` >= T keys; when T_new > T the first
 * T_new - T of them are additional already-seen keys in chronological order (this is how the
 * operator-level call with a concatenated full_k/full_v is expressed: k_cache = NULL,
 * T_new = S).  `pos` = tokens seen before this call; when pos_dev != NULL the kernels read the value
 * from device memory instead (one hipGraph then serves every step).
 * Workspace (split-KV partials): ivl_swa_workspace_bytes.
 * ------------------------------------------------------------------------------------------- */
typedef struct ivl_swa_args {
  const void* q;            /* bf16, element (b,t,h,:) at q + b*q_sb + t*q_st + h*q_sh            */
  const void* k_new;        /* bf16, element (b,j,hk,:) at k_new + b*kn_sb + j*kn_st + hk*kn_sh  */
  const void* v_new;        /* same strides as k_new                                              */
  const void* k_cache;      /* bf16 ring [B,Hkv,cache_capacity,d] contiguous, or NULL            */
  const void* v_cache;
  void* o;                  /* bf16 [B,T,Hq,d] contiguous                                          */
  int64_t q_sb, q_st, q_sh;
  int64_t kn_sb, kn_st, kn_sh;
  int B, T, T_new, Hq, Hkv, d;
  int cache_capacity;       /* C (= window-1 in the reference's cache); 0 when k_cache == NULL     */
  int window;               /* W; <= 0 means plain causal                                          */
  int64_t pos;              /* tokens seen before this call (ignored when pos_dev != NULL)         */
  const int64_t* pos_dev;   /* optional device scalar                                              */
  float scaling;
  void* workspace;
  size_t workspace_bytes;
  const void* rope_cos;     /* optional fused M-RoPE (std:949-984): cos / sin tables bf16 [3,B,T,d] as produced by  */
  const void* rope_sin;     /* InfiniteVLRotaryEmbedding; q and k_new are then the UN-rotated projections and are rotated while   */
  int rope_s0, rope_s1;     /* they are loaded (bit-identical to ivl_mrope_fwd).  Needs T_new == T; sections s0|s1|rest, multiples */
                            /* of 8.  NULL: q / k_new are already rotated.                                                         */
  int mma_dtype;            /* IVL_BF16 (the reference's precision) or IVL_FP8_E4M3: the single-token decode step
                               (T * Hq/Hkv <= 64 packed rows) rounds q, K, V and the probabilities to e4m3 for the two
                               products (BASELINE.json configs[4]); longer calls always run in bf16                  */
  int append_new;           /* != 0: after the attention, also append the call's T tokens to the ring (what ivl_swa_cache_append
                               does, same rope arguments) - inside the split-KV combine launch when there is one, so that a
                               layer costs one launch less.  Needs a ring cache and T_new == T.                      */
  int64_t pos_min;          /* a lower bound of the position the CALLER guarantees (0: none).  The position itself may live in
                               device memory (pos_dev), but a long call over a FULL ring (pos_min >= cache_capacity, window ==
                               cache_capacity + 1, T a multiple of 256, >= 256 workgroups of 256 rows, workspace of
                               ivl_swa_ring256_workspace_bytes) takes the 256-row form: the pre-pass lays the ring out in
                               chronological order in front of the call's keys (and appends), the attention kernel has no
                               ring / band arithmetic in its steady state.  Never set it for a call recorded into a graph that
                               may be replayed from an earlier position.                                                  */
} ivl_swa_args;

IVL_API size_t ivl_swa_workspace_bytes(int B, int T, int Hq, int d);
/* Workspace of the 256-row form of a long call over a full ring (see ivl_swa_args.pos_min); 0: the shape does not qualify. */
IVL_API size_t ivl_swa_ring256_workspace_bytes(int B, int T, int Hq, int Hkv, int d, int cache_capacity);
IVL_API int ivl_swa_fwd(const ivl_swa_args* args, void* stream);

/* Append the T new tokens to the ring (slot (pos+t) % C) -- after ivl_swa_fwd of the same call.
 * Replaces the tail copy-back of std:146-172.  k_new,v_new bf16 with the strides given.  With rope_cos / rope_sin
 * (same meaning as in ivl_swa_args) the keys are rotated on the way into the ring. */
IVL_API int ivl_swa_cache_append(const void* k_new, const void* v_new, int64_t kn_sb, int64_t kn_st, int64_t kn_sh,
                         void* k_cache, void* v_cache, int B, int T, int Hkv, int d, int cache_capacity,
                         int64_t pos, const int64_t* pos_dev, const void* rope_cos, const void* rope_sin,
                         int rope_s0, int rope_s1, void* stream);

/* 3-D rotary tables cos / sin bf16 [rows, 2*half_dim] from position ids (int64 [rows], rows = 3*B*T in (axis, batch,
 * token) order) and the inverse frequencies fp32 [half_dim]: cos(pos * inv_freq) in fp32, the frequencies repeated over
 * both halves of the channels, times attention_scaling, rounded to bf16.
 * Replaces InfiniteVLRotaryEmbedding.forward (std:896-930: cast, K=1 matmul, cat, cos, sin, scaling, cast) with one launch. */
IVL_API int ivl_rope_tables_fwd(const int64_t* position_ids, const float* inv_freq, void* cos_out, void* sin_out, int rows,
                        int half_dim, float attention_scaling, void* stream);

/* Vision-tower window attention: NON-causal softmax attention inside each segment
 * [cu_seqlens[s], cu_seqlens[s+1]) of one packed patch sequence, with the vision rotary embedding folded into the Q / K
 * loads.  Replaces apply_rotary_pos_emb_vision (strm:657-671) + the per-window attention_interface loop / the
 * flash-attention varlen call of InfiniteVLVisionAttention.forward (strm:752-796 = std:623-664).
 *   q, k, v : bf16, token t / head h at element offset t * x_st + h * x_sh (the three slices of the fused qkv projection
 *             output [S, 3, H, d] are passed without a copy); o : bf16 with its own strides ([S, H, d] contiguous for proj)
 *   cu_seqlens : device int32 [n_seg + 1], ascending, cu_seqlens[0] = 0 (strm:1076-1077); never read on the host
 *   max_seqlen : an upper bound of the segment lengths (the grid is sized from it; longer segments are NOT processed)
 *   d          : head_dim, 64 / 80 / 128 built (80 = the 3B vision tower: 1280 / 16)
 *   rope_cos, rope_sin : fp32 [S, d] (emb.cos(), emb.sin() of strm:1073-1074) or both NULL when q, k are already
 *             rotated; the rotation is done in fp32 with separately rounded products and sum, rounded to bf16 once:
 *             bit-identical to the reference's eager arithmetic.
 *   S          : number of tokens (patches) in the packed sequence
 *   workspace  : ivl_vision_attn_workspace_bytes(S, H, d, max_seqlen) bytes (0 when max_seqlen <= 64) or NULL.  With rope
 *             tables and segments longer than one 64-row query tile, the keys are rotated ONCE into it by a pre-pass;
 *             without it the rotation is redone in every query tile's loads (same results, slower).
 * Scores and the softmax statistics are fp32, the probabilities are rounded to bf16 for the PV product, fp32 accumulation. */
IVL_API size_t ivl_vision_attn_workspace_bytes(int S, int H, int d, int max_seqlen);
IVL_API int ivl_vision_attn_fwd(const void* q, const void* k, const void* v, void* o,
                        int64_t q_st, int64_t q_sh, int64_t k_st, int64_t k_sh, int64_t v_st, int64_t v_sh,
                        int64_t o_st, int64_t o_sh, const int32_t* cu_seqlens, int n_seg, int max_seqlen,
                        int S, int H, int d, float scaling, const float* rope_cos, const float* rope_sin,
                        void* workspace, size_t workspace_bytes, void* stream);

/* *counter += delta on the device (graph-replayable position bookkeeping). */
IVL_API int ivl_counter_add(int64_t* counter, int64_t delta, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused prologue / epilogue entry points: the same arithmetic as the
 * single-purpose entry points above, but reading the column blocks of ONE fused projection output
 * (row stride `ld` elements) so that a Gated DeltaNet layer needs one projection GEMM and one
 * prologue launch instead of six GEMMs + 3 conv + gate launches (call sites std:1261-1294).
 * ------------------------------------------------------------------------------------------- */

/* 3 short convs (+SiLU, carry-in, state in/out as in ivl_short_conv_fwd) + gate math (ivl_gdn_gate_fwd).
 * proj bf16 [B*T, ld]; q|k|v|a|b start at columns col_*; outputs q [B,T,Dq], k [B,T,Dk], v [B,T,Dv] bf16
 * contiguous, g fp32 [B,T,H], beta bf16 [B,T,H]. */
IVL_API int ivl_gdn_prologue_fwd(const void* proj, int64_t ld, int col_q, int col_k, int col_v, int col_a, int col_b,
                         const void* w_q, const void* w_k, const void* w_v,
                         const void* sq_in, const void* sk_in, const void* sv_in,
                         void* sq_out, void* sk_out, void* sv_out,
                         const float* A_log, const float* dt_bias,
                         void* q, void* k, void* v, float* g, void* beta,
                         int B, int T, int H, int Dq, int Dk, int Dv, int W, int apply_silu, void* stream);

/* ivl_rmsnorm_swish_gate_fwd with the gate read in place: gate element (token, head, c) at
 * gate + token*gate_ld + head*N + c; x,y [rows = tokens*H, N] contiguous. */
IVL_API int ivl_rmsnorm_swish_gate_strided_fwd(const void* x, const void* gate, int64_t gate_ld, int H,
                                       const void* weight, void* y, int rows, int N, float eps, void* stream);

/* ivl_mrope_fwd on column blocks of a fused qkv projection: q element (b,t,h,c) at
 * q + (b*T+t)*q_ld + h*d + c, k likewise with k_ld. */
IVL_API int ivl_mrope_strided_fwd(void* q, void* k, int64_t q_ld, int64_t k_ld, const void* cos, const void* sin,
                          int B, int T, int Hq, int Hkv, int d, int s0, int s1, int s2, void* stream);

/* Decoder-layer norm (std:1400-1420 / Qwen2RMSNorm) fused with the residual add:
 *   h = bf16(x + residual) -> h_out   (skipped when residual == NULL: h = x)
 *   y = bf16(weight * bf16(h * rsqrt(mean(h^2) + eps)));  x,residual,y,h_out bf16 [rows,N], N%8==0, N<=8192. */
IVL_API int ivl_add_rmsnorm_fwd(const void* x, const void* residual, const void* weight, void* y, void* h_out,
                        int rows, int N, float eps, void* stream);

/* SwiGLU gate over a fused gate|up projection: y[r,i] = bf16(bf16(silu(gu[r,i])) * gu[r,I+i]) (std:945). */
IVL_API int ivl_silu_mul_fwd(const void* gate_up, void* y, int64_t rows, int I, void* stream);

/* Gated DeltaNet mixer core for ONE new token per sequence (GatedDeltaNet.forward, std:1215-1347, q_len == 1):
 * = ivl_gdn_prologue_fwd + ivl_gdn_recurrent_fwd(use_qk_l2norm=1) + ivl_rmsnorm_swish_gate_strided_fwd in one launch,
 * same rounding points.  proj [B, ld] bf16 is the fused projection row (columns col_q|col_k: H*128 each, col_v|col_g:
 * H*256 each, col_a|col_b: H each); conv weights [D,4] bf16; conv states [B,D,4] bf16 and the recurrent state
 * [B,H,128,256] (IVL_F32 / IVL_BF16) are updated IN PLACE; y [B, H*256] bf16 is the gated-norm output (o_proj input). */
IVL_API int ivl_gdn_decode_step_fwd(const void* proj, int64_t ld, int col_q, int col_k, int col_v, int col_g, int col_a,
                            int col_b, const void* conv_wq, const void* conv_wk, const void* conv_wv,
                            void* conv_state_q, void* conv_state_k, void* conv_state_v, const float* A_log,
                            const float* dt_bias, const void* norm_weight, float eps, void* state, int state_dtype,
                            void* y, int B, int H, int K, int V, float scale, void* stream);

/* The same step on 4 x as many workgroups (round 5): a workgroup = (sequence, head, quarter of the 256 value columns); the delta
 * rule is column-local, so the quarters exchange nothing.  What spans a head moves into the NEXT launch
 * (ivl_gdn_out_linear_small_m_fwd, the o_proj weight stream): the gated RMSNorm, and the shift of the q / k conv states (every
 * quarter reads them).  Here: conv states of q and k are READ ONLY, the v conv state and the recurrent state are updated in
 * place, o_raw [B, H*256] bf16 receives the delta rule's UN-NORMALISED output (the bf16 rounding point of the recurrent kernel).
 * ivl_gdn_decode_split_fwd + ivl_gdn_out_linear_small_m_fwd == ivl_gdn_decode_step_fwd + ivl_linear_small_m_fwd, bit for bit
 * (outputs, recurrent state, all three conv states).  Replaces std:1241-1342 at q_len == 1. */
IVL_API int ivl_gdn_decode_split_fwd(const void* proj, int64_t ld, int col_q, int col_k, int col_v, int col_a, int col_b,
                             const void* conv_wq, const void* conv_wk, const void* conv_wv, const void* conv_state_q,
                             const void* conv_state_k, void* conv_state_v, const float* A_log, const float* dt_bias,
                             void* state, int state_dtype, void* o_raw, int B, int H, int K, int V, float scale, void* stream);
/* y[M,N] = bf16(o_norm(o_raw, gate)[M, H*256] W[N, H*256]^T + bias): the GDN output projection of a decode step (std:1336-1342)
 * with FusedRMSNormGated (fla:modules/fused_norm_gate.py:778-796; 256-wide heads, eps) in the prologue of the weight stream.
 * gate = first gate element of row 0 of the fused projection (row stride gate_ld elements); norm_weight bf16 [256].  Also shifts
 * the conv states [M, H*128, 4] of q and k by the step's raw projection values (proj + col_q / col_k, row stride proj_ld): the
 * second half of ivl_gdn_decode_split_fwd's state update.  M <= 4, H <= 16. */
IVL_API int ivl_gdn_out_linear_small_m_fwd(const void* o_raw, const void* gate, int64_t gate_ld, const void* norm_weight, float eps,
                                   int H, const void* proj, int64_t proj_ld, int col_q, int col_k, void* conv_state_q,
                                   void* conv_state_k, const void* w, const void* bias, void* y, int M, int N, int K, void* stream);

/* nn.Linear for the single-token decode step (M <= 4 rows): y[M,N] = bf16(x[M,K] W[N,K]^T + bias[N]).
 * Replaces the q/k/v/o, GDN in/out, MLP and tied lm_head projections (std:1047-1054, 1215-1240, 945, 2091-2092)
 * when q_len == 1: a pure weight stream bounded by HBM.  x,W,bias,y bf16, row-major contiguous; fp32 accumulation;
 * bias may be NULL.  K % 8 == 0.  IVL_ERR_UNSUPPORTED for M > 4 (callers use a GEMM there). */
IVL_API int ivl_linear_small_m_fwd(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, void* stream);

/* SwiGLU MLP head of a decode step (std:945: act_fn(gate_proj(x)) * up_proj(x)), fused gate|up weight [2I,K]:
 *   y[M,I] = bf16( bf16(silu(bf16(x Wg^T))) * bf16(x Wu^T) ),  Wg = w_gate_up[:I], Wu = w_gate_up[I:]
 * = ivl_linear_small_m_fwd on the fused weight followed by ivl_silu_mul_fwd, bit for bit.  bias [2I] or NULL. */
IVL_API int ivl_linear_swiglu_small_m_fwd(const void* x, const void* w_gate_up, const void* bias, void* y, int M, int I, int K,
                                  void* stream);

/* (Residual add +) RMSNorm in the prologue of the decode step's projection: ivl_add_rmsnorm_fwd followed by
 * ivl_linear_small_m_fwd (glu == 0) or ivl_linear_swiglu_small_m_fwd (glu != 0, N = I), bit for bit, in ONE launch --
 * the decoder layer's input_layernorm -> q|k|v / GDN in-projection, post_attention_layernorm -> gate|up, and the final
 * norm -> lm_head (std:1350-1429, 1573, 2091-2092) at q_len == 1.  (The package routes the 72 per-layer norms of a decode token through
 * it; the final norm in front of lm_head stays a launch of its own.)
 *   h = bf16(x + residual) (residual NULL: h = x; else written to h_out [M,K]);  xn = bf16(norm_weight * bf16(h * rstd(h)));
 *   y = linear(xn).  x, residual, h_out bf16 [M,K]; norm_weight bf16 [K]; 512 <= K <= 4096; the rest as the two entry points above. */
IVL_API int ivl_norm_linear_small_m_fwd(const void* x, const void* residual, const void* norm_weight, float eps, void* h_out,
                                const void* w, const void* bias, void* y, int M, int N, int K, int glu, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IVL_HIP_H */
