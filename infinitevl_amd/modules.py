"""Drop-in mixer modules: same class names, constructor arguments, parameter names/shapes,
forward signatures and return values as the reference's `GatedDeltaNet` (std:1116-1347) and
`InfiniteVLSelfAttention` (std:987-1113), so they slot into `InfiniteVLDecoderLayer`
(std:1361-1364) and load the reference checkpoint unchanged.  The arithmetic between the
projections runs in the gfx950 kernels (infinitevl_amd.ops); the projections themselves are stock
rocBLAS/hipBLASLt GEMMs through torch (out of scope).

`std:` = infinitevl/infinitevl_standard/modeling_infinitevl.py of the reference.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .cache import StaticLinearLayerPrealloc, StaticSlidingWindowLayerPrealloc


class InfiniteVLRotaryEmbedding(nn.Module):
    """3-D (t,h,w) rotary tables cos/sin [3,B,T,head_dim] in the activation dtype (std:896-930,
    'default' rope init)."""

    def __init__(self, config, device=None):
        super().__init__()
        self.config = config
        head_dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        theta = float(getattr(config, "rope_theta", 1e6))
        # kept as a plain fp32 tensor (not a buffer): `module.to(bfloat16)` must not round the frequencies
        self._inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
        self.attention_scaling = 1.0

    @property
    def inv_freq(self) -> torch.Tensor:
        return self._inv_freq

    @torch.no_grad()
    def forward(self, x: torch.Tensor, position_ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._inv_freq.device != x.device:
            self._inv_freq = self._inv_freq.to(x.device)
        if x.is_cuda and x.dtype == torch.bfloat16:
            # one launch instead of the eager chain (cast, outer product, cat, cos, sin, scalings, casts)
            return ops.rope_tables(position_ids, self._inv_freq, self.attention_scaling)
        # outer product position x frequency in fp32 (std:920-925 does it as a K=1 matmul)
        freqs = position_ids[..., None].float() * self._inv_freq
        emb = torch.cat((freqs, freqs), dim=-1)
        return (emb.cos() * self.attention_scaling).to(x.dtype), (emb.sin() * self.attention_scaling).to(x.dtype)


class InfiniteVLSelfAttention(nn.Module):
    """Sliding-window GQA attention mixer (std:987-1113)."""

    def __init__(self, config, layer_idx: Optional[int] = None):
        super().__init__()
        self.config = config
        self.layer_idx = layer_idx
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.is_causal = True
        self.attention_dropout = getattr(config, "attention_dropout", 0.0)
        self.rope_scaling = config.rope_scaling
        self.scaling = self.head_dim ** -0.5
        if (self.head_dim * self.num_heads) != self.hidden_size:
            raise ValueError(
                f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size}"
                f" and `num_heads`: {self.num_heads}).")
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=True)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=True)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=True)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=False)
        self.sliding_window = (config.sliding_window
                               if config.layer_types[self.layer_idx] == "sliding_attention" else None)
        self.rotary_emb = InfiniteVLRotaryEmbedding(config=config)
        self._fused_w = self._fused_b = None
        self.mma_dtype = None           # None = bf16 operands; "fp8_e4m3": single-token decode products on e4m3 (configs[4])

    @torch.no_grad()
    def fuse_(self) -> "InfiniteVLSelfAttention":
        """Inference-time: store q|k|v projection weights/biases in ONE tensor (one GEMM instead of three)
        and re-point the original parameters at views of it -- names, shapes and state_dict are unchanged
        and no memory is duplicated."""
        ws = [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight]
        bs = [self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]
        self._fused_w = torch.cat([w.data for w in ws], dim=0).contiguous()
        self._fused_b = torch.cat([b.data for b in bs], dim=0).contiguous()
        r = 0
        for w, b in zip(ws, bs):
            n = w.shape[0]
            w.data = self._fused_w[r:r + n]
            b.data = self._fused_b[r:r + n]
            r += n
        return self

    def _fused_ok(self, x: torch.Tensor) -> bool:
        return (self._fused_w is not None and x.dtype == torch.bfloat16 and self._fused_w.dtype == torch.bfloat16
                and self.q_proj.weight.data_ptr() == self._fused_w.data_ptr() and self.head_dim % 16 == 0)

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.LongTensor] = None, past_key_values=None,
                output_attentions: bool = False, use_cache: bool = False,
                cache_position: Optional[torch.LongTensor] = None,
                position_embeddings: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, **kwargs):
        if isinstance(hidden_states, ops.PreNorm) and not self._fused_ok(hidden_states):
            hidden_states = hidden_states.materialize()        # (a not-yet-launched input norm: only the fused q|k|v GEMV runs it)
        bsz, q_len, _ = hidden_states.size()
        cos, sin = position_embeddings
        # projections stay time-major [B,T,H,d]: the kernels take strides, no transpose/copy (std:1047-1054)
        layer = past_key_values.layers[self.layer_idx] if past_key_values is not None else None
        sec = self.rope_scaling["mrope_section"]
        native = layer is None or isinstance(layer, StaticSlidingWindowLayerPrealloc)
        # M-RoPE is fused into the attention / append kernels (q and k are rotated while they are loaded: no separate
        # launch, bit-identical) whenever our own cache (or none) is used; the fp8 decode step and the foreign-cache
        # protocol take rotated tensors
        fuse_rope = (native and self.head_dim == 128 and sec[0] % 8 == 0 and sec[1] % 8 == 0 and cos.shape[0] == 3
                     and not (self.mma_dtype is not None and q_len * self.num_key_value_groups <= 64))
        if self._fused_ok(hidden_states):
            nq, nkv = self.num_heads * self.head_dim, self.num_key_value_heads * self.head_dim
            qkv = ops.linear(hidden_states, self._fused_w, self._fused_b)                     # [B,T,nq+2nkv]
            q = qkv[..., :nq].unflatten(-1, (self.num_heads, self.head_dim))
            k = qkv[..., nq:nq + nkv].unflatten(-1, (self.num_key_value_heads, self.head_dim))
            v = qkv[..., nq + nkv:].unflatten(-1, (self.num_key_value_heads, self.head_dim))
            if not fuse_rope:
                ops.apply_mrope_strided_inplace(q, k, cos, sin, sec)
        else:
            q = self.q_proj(hidden_states).view(bsz, q_len, self.num_heads, self.head_dim)
            k = self.k_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim)
            v = self.v_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim)
            if not fuse_rope:
                ops.apply_mrope_inplace(q, k, cos, sin, sec)                                  # std:1057-1064
        rope = (cos, sin, sec) if fuse_rope else None

        if isinstance(layer, StaticSlidingWindowLayerPrealloc):
            attn = layer.attend(q, k, v, self.scaling, self.sliding_window, mma_dtype=self.mma_dtype, rope=rope)   # std:1067-1108 fused
        elif layer is None:
            attn = ops.swa_forward(q, k, v, window=self.sliding_window, scaling=self.scaling, rope=rope)
        else:  # foreign cache object: go through the reference protocol (cat of cached + new)
            fk, fv = past_key_values.update(layer_idx=self.layer_idx, key_states=k.transpose(1, 2),
                                            value_states=v.transpose(1, 2), conv_state=None, recurrent_state=None,
                                            cache_kwargs={"sin": sin, "cos": cos, "cache_position": cache_position})
            attn, _ = ops.swa_attention_interface(self, q.transpose(1, 2), fk, fv, None, scaling=self.scaling,
                                                  sliding_window=self.sliding_window)
        attn = attn.reshape(bsz, q_len, -1)
        return ops.linear(attn, self.o_proj.weight, self.o_proj.bias), None


class GatedDeltaNet(nn.Module):
    """Gated DeltaNet mixer (std:1116-1347): q/k/v short convs, gates, delta-rule kernel, gated norm."""

    def __init__(self, config, layer_idx: int):
        super().__init__()
        self.mode = config.mode
        self.hidden_size = config.hidden_size
        self.expand_v = config.expand_v
        self.norm_eps = config.norm_eps
        self.use_gate = config.use_gate
        self.use_short_conv = config.use_short_conv
        self.conv_size = config.conv_size
        self.conv_bias = config.conv_bias
        self.num_heads = config.num_linear_heads
        self.num_key_value_heads = config.num_linear_key_value_heads
        self.head_dim = getattr(config, "linear_head_dim", config.hidden_size // config.num_attention_heads)
        self.key_dim = int(self.num_key_value_heads * self.head_dim)
        self.value_dim = int(self.key_dim * self.expand_v)
        self.head_k_dim = self.head_dim
        self.head_v_dim = int(self.head_dim * self.expand_v)
        self.layer_idx = layer_idx
        if not math.isclose(self.key_dim * self.expand_v, self.value_dim, rel_tol=1e-5):
            raise ValueError(f"expand_v={self.expand_v} does not produce an integer value when multiplied by "
                             f"key_dim={self.key_dim}.")
        if not math.isclose(self.head_dim * self.expand_v, self.head_v_dim, rel_tol=1e-5):
            raise ValueError(f"expand_v={self.expand_v} does not produce an integer value when multiplied by "
                             f"head_dim={self.head_dim}.")
        assert self.mode in ["chunk", "fused_recurrent"], f"Not suppoerted mode `{self.mode}`."
        if not self.use_short_conv:
            raise UserWarning("ShortConvolution is crucial to the performance. Do not turn it off.")
        if self.num_key_value_heads != self.num_heads:
            raise NotImplementedError(
                f"num_linear_key_value_heads ({self.num_key_value_heads}) != num_linear_heads ({self.num_heads}): the "
                f"kernels index q, k and v with one head count (InfiniteVL ships 16 / 16)")

        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=False)
        self.k_proj = nn.Linear(self.hidden_size, self.key_dim, bias=False)
        self.v_proj = nn.Linear(self.hidden_size, self.value_dim, bias=False)
        self.a_proj = nn.Linear(self.hidden_size, self.num_heads, bias=False)
        self.b_proj = nn.Linear(self.hidden_size, self.num_heads, bias=False)
        A = torch.empty(self.num_heads, dtype=torch.float32).uniform_(0, 16)
        self.A_log = nn.Parameter(torch.log(A))
        self.A_log._no_weight_decay = True
        dt_min, dt_max, dt_init_floor = 0.001, 0.1, 1e-4
        dt = torch.exp(torch.rand(self.num_heads) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = torch.clamp(dt, min=dt_init_floor)
        self.dt_bias = nn.Parameter(dt + torch.log(-torch.expm1(-dt)))
        self.dt_bias._no_weight_decay = True
        self.q_conv1d = ops.ShortConvolution(self.num_heads * self.head_dim, self.conv_size, activation="silu")
        self.k_conv1d = ops.ShortConvolution(self.key_dim, self.conv_size, activation="silu")
        self.v_conv1d = ops.ShortConvolution(self.value_dim, self.conv_size, activation="silu")
        if self.use_gate:                                                              # std:1209-1213
            self.g_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_v_dim, bias=False)
            self.o_norm = ops.FusedRMSNormGated(self.head_v_dim, eps=self.norm_eps)
        else:
            self.o_norm = ops.RMSNorm(self.head_v_dim, eps=self.norm_eps)
        self.o_proj = nn.Linear(self.num_heads * self.head_v_dim, self.hidden_size, bias=False)
        self._fused_w = None
        self._fused_cols = None
        self._gate32 = None
        self.mma_dtype = None           # None = the reference's bf16 operands; "fp8_e4m3" = BASELINE.json configs[4]

    @torch.no_grad()
    def fuse_(self) -> "GatedDeltaNet":
        """Inference-time: q|k|v|g|a|b projection weights in ONE tensor (one GEMM instead of six; rows padded
        to a multiple of 8) with the original parameters re-pointed at views of it (names/shapes/state_dict
        unchanged, no duplicated memory), plus fp32 copies of A_log / dt_bias for the gate math."""
        if not self.use_gate:                        # the fused path reads its gate from the projection buffer
            return self
        ws = [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.g_proj.weight,
              self.a_proj.weight, self.b_proj.weight]
        total = sum(w.shape[0] for w in ws)
        ld = (total + 7) // 8 * 8
        fused = torch.zeros(ld, self.hidden_size, dtype=ws[0].dtype, device=ws[0].device)
        cols, r = [], 0
        for w in ws:
            n = w.shape[0]
            fused[r:r + n].copy_(w.data)
            w.data = fused[r:r + n]
            cols.append(r)
            r += n
        self._fused_w = fused
        self._fused_cols = cols                      # q, k, v, g, a, b
        self._gate32 = None
        return self

    def _gate_params32(self):
        """fp32 copies of A_log / dt_bias for the gate math (std:1294 upcasts them per call); cached and
        refreshed whenever the parameters are modified in place (e.g. load_state_dict after fuse_)."""
        key = (self.A_log._version, self.dt_bias._version, self.A_log.data_ptr(), self.dt_bias.data_ptr())
        if self._gate32 is None or self._gate32[0] != key:
            self._gate32 = (key, self.A_log.detach().float().contiguous(), self.dt_bias.detach().float().contiguous())
        return self._gate32[1], self._gate32[2]

    def _fused_ok(self, x: torch.Tensor, layer) -> bool:
        return (self._fused_w is not None and x.dtype == torch.bfloat16 and self._fused_w.dtype == torch.bfloat16
                and self.q_proj.weight.data_ptr() == self._fused_w.data_ptr()
                and (layer is None or (isinstance(layer, StaticLinearLayerPrealloc) and layer.dtype == torch.bfloat16))
                and self.q_conv1d.weight.dtype == torch.bfloat16)

    def _forward_fused(self, hidden_states: torch.Tensor, past_key_values, layer, cache_position, mode: str):
        """One projection GEMM -> one prologue launch (3 convs + gate math) -> delta-rule kernel (state in place)
        -> gated norm reading its gate from the projection buffer -> o_proj."""
        B, T, _ = hidden_states.shape
        H, K, V = self.num_heads, self.head_dim, self.head_v_dim
        Dq, Dk, Dv = H * K, self.key_dim, self.value_dim
        cq, ck, cv, cg, ca, cb = self._fused_cols
        proj = ops.linear(hidden_states, self._fused_w)                               # [B,T,ld]
        ld = proj.shape[-1]
        prev = (None, None, None)
        h0 = None
        if layer is not None:                                                          # std:1241-1251
            prev, h0 = past_key_values.update(layer_idx=self.layer_idx, key_states=None, value_states=None,
                                              conv_state=None, recurrent_state=None,
                                              cache_kwargs={"op": "get", "cache_position": cache_position})
            outs = (layer.conv_state_q, layer.conv_state_k, layer.conv_state_v)
        else:
            outs = (None, None, None)
        A32, dt32 = self._gate_params32()
        w = self.o_norm.weight
        w = w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)
        if (T == 1 and layer is not None and prev[0] is not None and h0 is not None and Dk == Dq
                and h0.data_ptr() == layer.recurrent_state.data_ptr() and K == 128 and V == 256):
            ow, ob = self.o_proj.weight, self.o_proj.bias
            if (ops._SPLIT_DECODE and B <= 4 and H <= 16 and ow.dtype == torch.bfloat16 and ow.is_contiguous() and cg % 4 == 0
                    and ld % 4 == 0 and (ob is None or (ob.dtype == torch.bfloat16 and ob.is_contiguous()))):
                # decode step on 64 workgroups (sequence x head x column quarter); the gated norm and the q / k conv-state shift ride
                # in the o_proj launch: the same two launches, 4 x the memory parallelism for the 2 MB of state (bit-identical)
                o_raw = ops.gdn_decode_split(proj, (cq, ck, cv, ca, cb),
                                             (self.q_conv1d.weight, self.k_conv1d.weight, self.v_conv1d.weight), outs, A32, dt32,
                                             layer.recurrent_state, H, K, V, K ** -0.5)
                y = ops.gdn_out_linear(o_raw, proj, cg, cq, ck, w, self.norm_eps, outs[0], outs[1], ow, ob, H)
                past_key_values.update(layer_idx=self.layer_idx, key_states=None, value_states=None, conv_state=outs,
                                       recurrent_state=layer.recurrent_state,
                                       cache_kwargs={"op": "set", "delta_len": T, "cache_position": cache_position})
                return y, None
            # decode step: convs + gates + delta rule + gated norm in ONE launch, cache tensors updated in place
            o = ops.gdn_decode_step(proj, (cq, ck, cv, cg, ca, cb),
                                    (self.q_conv1d.weight, self.k_conv1d.weight, self.v_conv1d.weight), outs, A32, dt32,
                                    w, self.norm_eps, layer.recurrent_state, H, K, V, K ** -0.5)
            past_key_values.update(layer_idx=self.layer_idx, key_states=None, value_states=None, conv_state=outs,
                                   recurrent_state=layer.recurrent_state,
                                   cache_kwargs={"op": "set", "delta_len": T, "cache_position": cache_position})
            return ops.linear(o, self.o_proj.weight, self.o_proj.bias), None
        convs = (self.q_conv1d.weight, self.k_conv1d.weight, self.v_conv1d.weight)
        if mode == "chunk" and Dk == Dq and K == 128 and V == 256 and convs[0].shape[-1] == 4:
            # convs + gates inside the chunk kernel's pre-pass: q / k / v / g / beta never reach HBM, one launch less
            o = ops.gdn_chunk_fused(proj, (cq, ck, cv, ca, cb), convs, prev, outs, A32, dt32, H, K, V,
                                    initial_state=h0, final_state_out=layer.recurrent_state if layer is not None else None,
                                    mma_dtype=self.mma_dtype)
        else:
            q, k, v, g, beta = ops.gdn_prologue(proj, (cq, ck, cv, ca, cb), convs, prev, outs, A32, dt32, H, Dq, Dk, Dv)
            fn = ops.chunk_gated_delta_rule if mode == "chunk" else ops.fused_recurrent_gated_delta_rule
            extra = {"mma_dtype": self.mma_dtype} if mode == "chunk" else {}
            o, _ = fn(q=q.view(B, T, H, K), k=k.view(B, T, self.num_key_value_heads, K), v=v.view(B, T, H, V), g=g,
                      beta=beta, initial_state=h0, use_qk_l2norm_in_kernel=True,
                      final_state_out=layer.recurrent_state if layer is not None else None, **extra)
        if layer is not None:                                                          # std:1325-1333
            past_key_values.update(layer_idx=self.layer_idx, key_states=None, value_states=None, conv_state=outs,
                                   recurrent_state=layer.recurrent_state,
                                   cache_kwargs={"op": "set", "delta_len": T, "cache_position": cache_position})
        o = ops.rmsnorm_swish_gate_strided(o, proj[..., cg:], ld, w, self.norm_eps)
        return ops.linear(o.reshape(B, T, -1), self.o_proj.weight, self.o_proj.bias), None

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                past_key_values=None, cache_position: Optional[torch.LongTensor] = None, **kwargs):
        # the reference nulls the mask (std:1223): padded batches are not supported by the path
        batch_size, q_len, _ = hidden_states.shape
        mode = "fused_recurrent" if q_len <= 64 else self.mode                        # std:1230
        if kwargs.get("cu_seqlens", None) is not None:
            raise NotImplementedError("variable-length inputs are not used by InfiniteVL (std:1223)")

        layer = past_key_values.layers[self.layer_idx] if past_key_values is not None else None
        if self._fused_ok(hidden_states, layer):
            return self._forward_fused(hidden_states, past_key_values, layer, cache_position, mode)
        if isinstance(hidden_states, ops.PreNorm):
            hidden_states = hidden_states.materialize()
        use_cache = past_key_values is not None
        native = isinstance(layer, StaticLinearLayerPrealloc)
        prev_conv, recurrent_state = (None, None, None), None
        if use_cache:                                                                  # std:1241-1251
            prev_conv, recurrent_state = past_key_values.update(
                layer_idx=self.layer_idx, key_states=None, value_states=None, conv_state=None, recurrent_state=None,
                cache_kwargs={"op": "get", "cache_position": cache_position})
        # With our own cache class the kernels write conv / recurrent state straight into the pre-allocated
        # tensors (in the cache dtype, i.e. the reference's fp32 -> cache-dtype rounding of std:335).
        q_lin, k_lin, v_lin = self.q_proj(hidden_states), self.k_proj(hidden_states), self.v_proj(hidden_states)
        if native:
            q, sq = _conv_into(self.q_conv1d, q_lin, prev_conv[0], layer.conv_state_q)
            k, sk = _conv_into(self.k_conv1d, k_lin, prev_conv[1], layer.conv_state_k)
            v, sv = _conv_into(self.v_conv1d, v_lin, prev_conv[2], layer.conv_state_v)
        else:
            q, sq = self.q_conv1d(q_lin, cache=prev_conv[0], output_final_state=use_cache)
            k, sk = self.k_conv1d(k_lin, cache=prev_conv[1], output_final_state=use_cache)
            v, sv = self.v_conv1d(v_lin, cache=prev_conv[2], output_final_state=use_cache)
        q = q.view(batch_size, q_len, self.num_heads, self.head_dim)
        k = k.view(batch_size, q_len, self.num_key_value_heads, self.head_k_dim)
        v = v.view(batch_size, q_len, self.num_key_value_heads, self.head_v_dim)

        g, beta = ops.gdn_gate(self.a_proj(hidden_states), self.b_proj(hidden_states), self.A_log, self.dt_bias)

        fn = ops.chunk_gated_delta_rule if mode == "chunk" else ops.fused_recurrent_gated_delta_rule
        extra = {"mma_dtype": self.mma_dtype} if mode == "chunk" else {}
        o, next_state = fn(q=q, k=k, v=v, g=g, beta=beta, initial_state=recurrent_state,
                           output_final_state=use_cache and not native, use_qk_l2norm_in_kernel=True,
                           final_state_out=layer.recurrent_state if native else None, **extra)

        if use_cache:                                                                  # std:1325-1333
            past_key_values.update(
                layer_idx=self.layer_idx, key_states=None, value_states=None, conv_state=(sq, sk, sv),
                recurrent_state=next_state,
                cache_kwargs={"op": "set", "delta_len": q_len, "cache_position": cache_position})

        if self.use_gate:                                                              # std:1336-1341
            g_gate = self.g_proj(hidden_states).view(batch_size, q_len, self.num_heads, self.head_v_dim)
            o = self.o_norm(o, g_gate)
        else:
            o = self.o_norm(o)
        o = self.o_proj(o.reshape(batch_size, q_len, -1))
        return o, None


class InfiniteVLVisionAttention(nn.Module):
    """Vision-tower attention block (strm:713-801 = std:583-668): fused `qkv` projection (bias), vision rotary embedding,
    NON-causal attention inside each window / frame segment of `cu_seqlens`, `proj`.  Parameter names and shapes match the
    checkpoint (`qkv` [3*dim, dim] + bias, `proj` [dim, dim] + bias).  The rotary embedding and the per-segment attention
    run in ONE launch (ops.vision_window_attention) on the strided q / k / v slices of the qkv output - no per-window
    Python loop, no `.tolist()` host sync (strm:775-778), so the tower is graph-capturable for any `cu_seqlens` bounded by
    `max_seqlen`.

    forward(hidden_states [S, dim], cu_seqlens int32 [n+1], rotary_pos_emb=None, position_embeddings=(cos, sin) [S, head_dim],
    max_seqlen=None) -> [S, dim].  `max_seqlen` (kwarg; the reference computes it on the device, strm:754) is an upper bound
    of the segment lengths; when omitted it is read from `cu_seqlens` with a host sync, like the reference's own fallback."""

    def __init__(self, config) -> None:
        super().__init__()
        self.dim = config.hidden_size
        self.num_heads = config.num_heads
        self.head_dim = self.dim // self.num_heads
        self.num_key_value_groups = 1
        self.qkv = nn.Linear(self.dim, self.dim * 3, bias=True)
        self.proj = nn.Linear(self.dim, self.dim)
        self.scaling = self.head_dim ** -0.5
        self.config = config
        self.attention_dropout = 0.0
        self.is_causal = False

    def forward(self, hidden_states: torch.Tensor, cu_seqlens: torch.Tensor, rotary_pos_emb: Optional[torch.Tensor] = None,
                position_embeddings: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, **kwargs) -> torch.Tensor:
        seq_length = hidden_states.shape[0]
        qkv = self.qkv(hidden_states).view(seq_length, 3, self.num_heads, self.head_dim)
        max_seqlen = kwargs.get("max_seqlen", None)
        if max_seqlen is None:
            win = kwargs.get("win_lengths_list", None)                      # the streaming variant's precomputed list (strm:773)
            max_seqlen = max(win) if win else int((cu_seqlens[1:] - cu_seqlens[:-1]).max())
        if position_embeddings is None:
            # older call sites hand over the raw frequencies only (HF Qwen2.5-VL: emb = cat(freqs, freqs) -> cos / sin);
            # with neither the reference fails on unpacking None (strm:742) - so do we, with a message
            if rotary_pos_emb is None:
                raise TypeError("InfiniteVLVisionAttention.forward needs position_embeddings=(cos, sin) or rotary_pos_emb")
            emb = torch.cat((rotary_pos_emb, rotary_pos_emb), dim=-1).float()
            position_embeddings = (emb.cos(), emb.sin())
        attn = ops.vision_window_attention(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu_seqlens, int(max_seqlen), self.scaling,
                                           rope=position_embeddings)
        return self.proj(attn.view(seq_length, -1))


def _conv_into(mod: "ops.ShortConvolution", x: torch.Tensor, prev: Optional[torch.Tensor], dst: torch.Tensor):
    """Run the short conv reading history from `prev` (None = zero history, the reference's first call,
    std:298-300) and writing the new state into the pre-allocated cache tensor `dst` (may alias prev)."""
    from . import _lib
    from .ops import _p, _stream
    if dst.dtype != torch.bfloat16:
        # non-bf16 cache: run on a bf16 copy of the history; cache.update(op="set") copies the result back
        tmp = prev.to(torch.bfloat16) if prev is not None else None
        return mod(x, cache=tmp, output_final_state=True)
    B, T, D = x.shape
    W = mod.kernel_size[0]
    x = x.contiguous()
    y = torch.empty_like(x)
    w = mod.weight if mod.weight.dtype == torch.bfloat16 else mod.weight.to(torch.bfloat16)
    _lib.check(_lib.load().ivl_short_conv_fwd(_p(x), _p(w.contiguous()), _p(prev), _p(y), _p(dst), B, T, D, W,
                                              int(mod.activation is not None), _stream(x)))
    return y, dst
