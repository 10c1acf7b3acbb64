"""Oracle (CPU) for the layer level of the hot path: the Gated DeltaNet mixer,
the sliding-window mixer, the decoder layer around them, and a minimal text
stack + greedy loop used by the harness fixtures.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  This is also the
"reference CPU eager path" that bench.py times as `cpu_baseline` (BASELINE.md
section 3): the reference itself has no CPU path (FA2 hard-coded at std:1028, Triton
ops at std:52-54).
`std:` = /root/reference/infinitevl/infinitevl_standard/modeling_infinitevl.py.

Weights are passed as plain dicts keyed by the reference's parameter names
(`q_proj.weight`, `q_conv1d.weight`, `A_log`, `o_norm.weight`, ...).
`act_dtype` (e.g. torch.bfloat16) emulates a reduced-precision model by rounding
every module output; `None` keeps fp32 throughout.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import gdn as _gdn
from . import swa as _swa
from .cache import LinearCounters, SwaCounters


@dataclass
class OracleConfig:
    hidden_size: int = 2048
    intermediate_size: int = 11008
    num_attention_heads: int = 16
    num_key_value_heads: int = 2
    num_linear_heads: int = 16
    linear_head_dim: int = 128
    expand_v: float = 2.0
    conv_size: int = 4
    sliding_window: int = 8192
    rope_theta: float = 1e6
    mrope_section: List[int] = field(default_factory=lambda: [16, 24, 24])
    rms_norm_eps: float = 1e-6
    norm_eps: float = 1e-5
    layer_types: List[str] = field(default_factory=list)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def head_v_dim(self) -> int:
        return int(self.linear_head_dim * self.expand_v)


class OracleLayerState:
    """Per-layer cache content (tensor + integer side) mirroring std:66-364."""

    def __init__(self, kind: str, cfg: OracleConfig, cache_dtype: Optional[torch.dtype]):
        self.kind = kind
        self.cache_dtype = cache_dtype
        if kind == "sliding_attention":
            self.counters = SwaCounters(cfg.sliding_window)
            self.k: Optional[torch.Tensor] = None       # [B,Hkv,size,d]
            self.v: Optional[torch.Tensor] = None
        else:
            self.counters = LinearCounters()
            self.conv = (None, None, None)
            self.recurrent: Optional[torch.Tensor] = None

    def clone(self) -> "OracleLayerState":
        import copy
        new = copy.copy(self)
        new.counters = copy.copy(self.counters)
        for name in ("k", "v", "recurrent"):
            t = getattr(self, name, None)
            if t is not None:
                setattr(new, name, t.clone())
        if self.kind != "sliding_attention":
            new.conv = tuple(None if c is None else c.clone() for c in self.conv)
        return new


def new_cache(cfg: OracleConfig, cache_dtype: Optional[torch.dtype] = None) -> List[OracleLayerState]:
    return [OracleLayerState(t, cfg, cache_dtype) for t in cfg.layer_types]


def clone_cache(cache: List[OracleLayerState]) -> List[OracleLayerState]:
    return [s.clone() for s in cache]


def _rd(x: torch.Tensor, dt: Optional[torch.dtype]) -> torch.Tensor:
    return x if dt is None else x.to(dt).float()


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float, act_dtype=None) -> torch.Tensor:
    """Qwen2RMSNorm: fp32 statistics, cast to the activation dtype, then * weight."""
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return _rd(w.float() * _rd(y, act_dtype), act_dtype)


def _linear(x, p, name, act_dtype):
    y = F.linear(x, p[name + ".weight"].float(), p.get(name + ".bias", None))
    return _rd(y, act_dtype)


def gdn_layer(p: Dict[str, torch.Tensor], x: torch.Tensor, cfg: OracleConfig,
              state: Optional[OracleLayerState], act_dtype=None, kernel_rounding=None) -> torch.Tensor:
    """GatedDeltaNet.forward, std:1215-1347."""
    B, T, _ = x.shape
    H, K, V = cfg.num_linear_heads, cfg.linear_head_dim, cfg.head_v_dim
    use_cache = state is not None
    prev_conv, prev_rec = (None, None, None), None
    if use_cache and state.counters.get():                      # std:1241-1251, 298-300
        prev_conv, prev_rec = state.conv, state.recurrent

    q_lin, k_lin, v_lin = (_linear(x, p, n, act_dtype) for n in ("q_proj", "k_proj", "v_proj"))
    q, sq = _gdn.short_conv(q_lin, p["q_conv1d.weight"], prev_conv[0])       # std:1261-1281
    k, sk = _gdn.short_conv(k_lin, p["k_conv1d.weight"], prev_conv[1])
    v, sv = _gdn.short_conv(v_lin, p["v_conv1d.weight"], prev_conv[2])
    q, k, v = _rd(q, act_dtype), _rd(k, act_dtype), _rd(v, act_dtype)
    q, k, v = q.view(B, T, H, K), k.view(B, T, H, K), v.view(B, T, H, V)

    a = _linear(x, p, "a_proj", act_dtype)
    b = _linear(x, p, "b_proj", act_dtype)
    g, beta = _gdn.gate_math(a, b, p["A_log"], p["dt_bias"].float())         # std:1293-1294
    beta = _rd(beta, act_dtype)

    if T <= 64:                                                              # std:1230
        o, S = _gdn.gdn_recurrent(q, k, v, g, beta, initial_state=prev_rec,
                                  qk_round_dtype=kernel_rounding)
    else:
        o, S = _gdn.gdn_chunk(q, k, v, g, beta, initial_state=prev_rec, rounding=kernel_rounding)
    o = _rd(o, act_dtype)

    if use_cache:                                                            # std:1325-1333, 335
        cd = state.cache_dtype
        state.conv = tuple(_rd(s, cd) for s in (sq, sk, sv))
        state.recurrent = _rd(S, cd)
        state.counters.set(T)

    gate = _linear(x, p, "g_proj", act_dtype).view(B, T, H, V)
    o = _rd(_gdn.rmsnorm_swish_gate(o, gate, p["o_norm.weight"], cfg.norm_eps), act_dtype)   # std:1336-1338
    return _linear(o.reshape(B, T, H * V), p, "o_proj", act_dtype)


def swa_layer(p: Dict[str, torch.Tensor], x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
              cfg: OracleConfig, state: Optional[OracleLayerState], act_dtype=None) -> torch.Tensor:
    """InfiniteVLSelfAttention.forward (std:1032-1113) with the window defined by
    oracle.swa.window_bounds (what FA2 computes)."""
    B, T, _ = x.shape
    Hq, Hkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    q = _linear(x, p, "q_proj", act_dtype).view(B, T, Hq, d).transpose(1, 2)
    k = _linear(x, p, "k_proj", act_dtype).view(B, T, Hkv, d).transpose(1, 2)
    v = _linear(x, p, "v_proj", act_dtype).view(B, T, Hkv, d).transpose(1, 2)
    q, k = _swa.apply_mrope(q, k, _rd(cos, act_dtype), _rd(sin, act_dtype), cfg.mrope_section)
    q, k = _rd(q, act_dtype), _rd(k, act_dtype)

    n_prev = 0
    if state is not None:                                                    # std:126-173
        cd = state.cache_dtype
        n_prev = state.counters.size
        if state.k is not None:
            full_k = torch.cat([state.k, k], dim=2)
            full_v = torch.cat([state.v, v], dim=2)
        else:
            full_k, full_v = k, v
        state.counters.update(T)
        keep = state.counters.size
        state.k = _rd(full_k[:, :, full_k.shape[2] - keep:], cd)
        state.v = _rd(full_v[:, :, full_v.shape[2] - keep:], cd)
    else:
        full_k, full_v = k, v
    o = _swa.swa_attention(q, full_k, full_v, n_prev, cfg.sliding_window, d ** -0.5,
                           p_round_dtype=act_dtype)
    o = _rd(o, act_dtype)
    return _linear(o.reshape(B, T, Hq * d), p, "o_proj", act_dtype)


def mlp(p, x, act_dtype=None):
    gate = _linear(x, p, "gate_proj", act_dtype)
    up = _linear(x, p, "up_proj", act_dtype)
    return _linear(_rd(F.silu(gate) * up, act_dtype), p, "down_proj", act_dtype)


def _sub(p: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    n = len(prefix)
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix)}


def decoder_layer(p, x, cos, sin, cfg: OracleConfig, layer_idx: int, state, act_dtype=None,
                  kernel_rounding=None):
    """InfiniteVLDecoderLayer.forward, std:1372-1429."""
    h = rms_norm(x, p["input_layernorm.weight"], cfg.rms_norm_eps, act_dtype)
    pa = _sub(p, "self_attn.")
    if cfg.layer_types[layer_idx] == "linear_attention":
        h = gdn_layer(pa, h, cfg, state, act_dtype, kernel_rounding)
    else:
        h = swa_layer(pa, h, cos, sin, cfg, state, act_dtype)
    x = _rd(x + h, act_dtype)
    h = rms_norm(x, p["post_attention_layernorm.weight"], cfg.rms_norm_eps, act_dtype)
    return _rd(x + mlp(_sub(p, "mlp."), h, act_dtype), act_dtype)


def text_stack(params: Dict[str, torch.Tensor], inputs_embeds: torch.Tensor, position_ids: torch.Tensor,
               cfg: OracleConfig, cache: Optional[List[OracleLayerState]], act_dtype=None,
               kernel_rounding=None) -> torch.Tensor:
    """36-layer loop + final norm (the part of InfiniteVLTextModel.forward the path
    lives in, std:1549-1575).  position_ids [3,B,T].  Returns hidden [B,T,D]."""
    cos, sin = _swa.rotary_cos_sin(position_ids, cfg.head_dim, cfg.rope_theta)
    x = inputs_embeds.float()
    for i in range(len(cfg.layer_types)):
        st = cache[i] if cache is not None else None
        x = decoder_layer(_sub(params, f"layers.{i}."), x, cos, sin, cfg, i, st, act_dtype, kernel_rounding)
    return rms_norm(x, params["norm.weight"], cfg.rms_norm_eps, act_dtype)


def random_params(cfg: OracleConfig, seed: int = 0, std: float = 0.02, vocab: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init weights with the reference's parameter names and shapes
    (std:1019-1022, 1161-1213, 939-941).  fp32."""
    gen = torch.Generator().manual_seed(seed)
    D, I = cfg.hidden_size, cfg.intermediate_size
    Hq, Hkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    H, K, V, W = cfg.num_linear_heads, cfg.linear_head_dim, cfg.head_v_dim, cfg.conv_size

    def rn(*shape, s=std):
        return torch.randn(*shape, generator=gen) * s

    p: Dict[str, torch.Tensor] = {}
    for i, lt in enumerate(cfg.layer_types):
        pre = f"layers.{i}."
        p[pre + "input_layernorm.weight"] = 1.0 + rn(D, s=0.1)
        p[pre + "post_attention_layernorm.weight"] = 1.0 + rn(D, s=0.1)
        p[pre + "mlp.gate_proj.weight"] = rn(I, D)
        p[pre + "mlp.up_proj.weight"] = rn(I, D)
        p[pre + "mlp.down_proj.weight"] = rn(D, I)
        a = pre + "self_attn."
        if lt == "linear_attention":
            p[a + "q_proj.weight"] = rn(H * K, D)
            p[a + "k_proj.weight"] = rn(H * K, D)
            p[a + "v_proj.weight"] = rn(H * V, D)
            p[a + "a_proj.weight"] = rn(H, D)
            p[a + "b_proj.weight"] = rn(H, D)
            p[a + "g_proj.weight"] = rn(H * V, D)
            p[a + "o_proj.weight"] = rn(D, H * V)
            p[a + "A_log"] = torch.log(torch.empty(H).uniform_(0.5, 16, generator=gen))
            dt = torch.exp(torch.rand(H, generator=gen) * (torch.log(torch.tensor(0.1)) - torch.log(torch.tensor(0.001)))
                           + torch.log(torch.tensor(0.001))).clamp(min=1e-4)
            p[a + "dt_bias"] = dt + torch.log(-torch.expm1(-dt))
            p[a + "q_conv1d.weight"] = rn(H * K, 1, W, s=0.3)
            p[a + "k_conv1d.weight"] = rn(H * K, 1, W, s=0.3)
            p[a + "v_conv1d.weight"] = rn(H * V, 1, W, s=0.3)
            p[a + "o_norm.weight"] = 1.0 + rn(V, s=0.1)
        else:
            p[a + "q_proj.weight"] = rn(Hq * d, D)
            p[a + "q_proj.bias"] = rn(Hq * d, s=0.1)
            p[a + "k_proj.weight"] = rn(Hkv * d, D)
            p[a + "k_proj.bias"] = rn(Hkv * d, s=0.1)
            p[a + "v_proj.weight"] = rn(Hkv * d, D)
            p[a + "v_proj.bias"] = rn(Hkv * d, s=0.1)
            p[a + "o_proj.weight"] = rn(D, Hq * d)
    p["norm.weight"] = 1.0 + rn(D, s=0.1)
    if vocab:
        p["embed_tokens.weight"] = rn(vocab, D)
    return p
