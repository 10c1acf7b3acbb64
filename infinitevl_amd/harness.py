"""Harness around the hot path: the build's own counterpart of the
reference's decoder layer / text-model loop (std:1350-1429, 1549-1575), `allocate_inference_cache`
(std:2316-2322) and the CUDA-graph streaming demo (inference_examples/demo_streaming_inference.py:
262-269 static buffers, 473-489 capture/replay, 399-422 greedy loop, 111-160 cache clone).

Everything that is not the Gated DeltaNet / SWA mixer (RMSNorm, SwiGLU MLP, embeddings, lm_head) is
stock PyTorch-ROCm (rocBLAS/hipBLASLt GEMMs): callers of the path, kept as-is.

`std:` = infinitevl/infinitevl_standard/modeling_infinitevl.py of the reference.
"""
from __future__ import annotations

from dataclasses import dataclass, field, fields
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .cache import StaticCachePrealloc

import os as _os

from .modules import GatedDeltaNet, InfiniteVLRotaryEmbedding, InfiniteVLSelfAttention


@dataclass
class InfiniteVLTextConfig:
    """The fields of the reference's InfiniteVLTextConfig that the hot path reads
    (configuration_infinitevl.py:208-284); defaults = the shipped InfiniteVL-3B config.json."""
    vocab_size: int = 151936
    hidden_size: int = 2048
    intermediate_size: int = 11008
    num_hidden_layers: int = 36
    num_attention_heads: int = 16
    num_key_value_heads: int = 2
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    norm_eps: float = 1e-5
    rope_theta: float = 1e6
    rope_scaling: Dict = field(default_factory=lambda: {"type": "default", "rope_type": "default",
                                                         "mrope_section": [16, 24, 24]})
    sliding_window: int = 8192
    use_sliding_window: bool = True
    layer_types: Optional[List[str]] = None
    attention_dropout: float = 0.0
    max_position_embeddings: int = 128000
    tie_word_embeddings: bool = True
    expand_v: float = 2
    mode: str = "chunk"
    use_gate: bool = True
    use_short_conv: bool = True
    conv_size: int = 4
    conv_bias: bool = False
    num_linear_heads: int = 16
    num_linear_key_value_heads: int = 16
    linear_head_dim: int = 128

    def __post_init__(self):
        if self.layer_types is None:            # configuration_infinitevl.py:278-284: every 4th layer is SWA
            self.layer_types = ["sliding_attention" if i % 4 == 0 else "linear_attention"
                                for i in range(self.num_hidden_layers)]

    @classmethod
    def from_hf_config(cls, cfg) -> "InfiniteVLTextConfig":
        """From the reference's config.json (path or dict): text fields live at the top level or under `text_config`
        (configuration_infinitevl.py:287-330); unknown keys (vision_config, token ids, ...) are ignored."""
        if isinstance(cfg, (str, bytes, _os.PathLike)):
            import json
            with open(cfg) as f:
                cfg = json.load(f)
        src = dict(cfg)
        src.update(cfg.get("text_config") or {})
        names = {f_.name for f_ in fields(cls)}
        return cls(**{k: v for k, v in src.items() if k in names and v is not None})


class InfiniteVLRMSNorm(nn.Module):
    """Qwen2RMSNorm (std:50): fp32 statistics, cast back, times weight."""

    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda and x.dtype == torch.bfloat16:          # one launch instead of seven elementwise kernels
            return ops.add_rmsnorm(x, None, self.weight, self.variance_epsilon)[0]
        dt = x.dtype
        xf = x.to(torch.float32)
        xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.variance_epsilon)
        return self.weight * xf.to(dt)

    def add_and_norm(self, x: torch.Tensor, residual: torch.Tensor):
        """(x + residual, norm(x + residual)) in one launch."""
        if x.is_cuda and x.dtype == torch.bfloat16:
            y, h = ops.add_rmsnorm(x, residual, self.weight, self.variance_epsilon)
            return h, y
        h = residual + x
        return h, self.forward(h)


class InfiniteVLTextMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.gate_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.up_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)

        self._fused_w = None

    @torch.no_grad()
    def fuse_(self) -> "InfiniteVLTextMLP":
        """gate|up projection weights in one tensor (one GEMM), parameters re-pointed at views of it."""
        g, u = self.gate_proj.weight, self.up_proj.weight
        self._fused_w = torch.cat([g.data, u.data], dim=0).contiguous()
        n = g.shape[0]
        g.data, u.data = self._fused_w[:n], self._fused_w[n:]
        return self

    def forward(self, x):
        if (self._fused_w is not None and x.is_cuda and x.dtype == torch.bfloat16
                and self.gate_proj.weight.data_ptr() == self._fused_w.data_ptr()):
            return ops.linear(ops.linear_swiglu(x, self._fused_w), self.down_proj.weight)
        if isinstance(x, ops.PreNorm):
            x = x.materialize()
        return self.down_proj(F.silu(self.gate_proj(x)) * self.up_proj(x))


class InfiniteVLDecoderLayer(nn.Module):
    """std:1350-1429: RMSNorm -> mixer -> +res -> RMSNorm -> SwiGLU -> +res."""

    def __init__(self, config, layer_idx: int):
        super().__init__()
        self.layer_type = config.layer_types[layer_idx]
        if self.layer_type == "linear_attention":
            self.self_attn = GatedDeltaNet(config, layer_idx)
        else:
            self.self_attn = InfiniteVLSelfAttention(config, layer_idx)
        self.mlp = InfiniteVLTextMLP(config)
        self.input_layernorm = InfiniteVLRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = InfiniteVLRMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None,
                output_attentions=False, use_cache=False, cache_position=None, position_embeddings=None, **kwargs):
        residual = hidden_states
        h = self.input_layernorm(hidden_states)
        h, _ = self.self_attn(hidden_states=h, attention_mask=attention_mask, position_ids=position_ids,
                              past_key_values=past_key_values, output_attentions=output_attentions,
                              use_cache=use_cache, cache_position=cache_position,
                              position_embeddings=position_embeddings, **kwargs)
        residual, h = self.post_attention_layernorm.add_and_norm(h, residual)   # residual + h, and its norm
        return (residual + self.mlp(h),)


class InfiniteVLTextStack(nn.Module):
    """Embedding + N decoder layers + final norm + tied lm_head: the text side of the model as the
    streaming demo drives it (inputs_embeds already contain the ViT features)."""

    def __init__(self, config: InfiniteVLTextConfig):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([InfiniteVLDecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = InfiniteVLRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.rotary_emb = InfiniteVLRotaryEmbedding(config)

    @torch.no_grad()
    def init_weights_(self, seed: int = 0, std: float = 0.02) -> "InfiniteVLTextStack":
        """Random init in the spirit of the reference's initializer_range=0.02 (there is no checkpoint
        offline).  Works in place on whatever device/dtype the parameters live on."""
        dev = next(self.parameters()).device
        gen = torch.Generator(device=dev).manual_seed(seed)
        import math
        for name, p_ in self.named_parameters():
            if name.endswith("A_log"):              # same distributions as the constructor (std:1177-1190), but drawn
                a = torch.rand(p_.shape, generator=gen, device=dev, dtype=torch.float32) * 16   # from THIS generator so
                p_.copy_(torch.log(a.clamp_min(1e-3)))                                      # replicas agree
                continue
            if name.endswith("dt_bias"):
                u = torch.rand(p_.shape, generator=gen, device=dev, dtype=torch.float32)
                dt = torch.exp(u * (math.log(0.1) - math.log(0.001)) + math.log(0.001)).clamp_min(1e-4)
                p_.copy_(dt + torch.log(-torch.expm1(-dt)))
                continue
            if "layernorm" in name or name.endswith("norm.weight"):
                p_.fill_(1.0)
            elif "conv1d" in name:
                p_.normal_(0.0, 0.3, generator=gen)
            else:
                p_.normal_(0.0, std, generator=gen)
        return self

    @torch.no_grad()
    def fuse_(self) -> "InfiniteVLTextStack":
        """Inference-time weight fusion (call after loading / casting the weights): q|k|v(|g|a|b) and gate|up
        projections become single GEMMs; parameter names, shapes and the state_dict stay as in the reference."""
        for layer in self.layers:
            layer.self_attn.fuse_()
            layer.mlp.fuse_()
        return self

    def set_mma_dtype(self, mma_dtype) -> "InfiniteVLTextStack":
        """Operand format of the mixers' MFMA products: None / "bf16" = the reference's precision; "fp8_e4m3" =
        BASELINE.json configs[4] (e4m3 operands in the Gated DeltaNet chunk scan and the SWA decode step; fp32
        accumulation and state).  Graphs captured before the switch keep the format they were captured with."""
        ops.mma_code(mma_dtype)                      # validates
        for layer in self.layers:
            layer.self_attn.mma_dtype = mma_dtype
        return self

    def allocate_inference_cache(self, batch_size: int = 1, dtype: Optional[torch.dtype] = None,
                                 zero_init: bool = False) -> StaticCachePrealloc:
        p_ = next(self.parameters())
        return StaticCachePrealloc(config=self.config, batch_size=batch_size, device=p_.device,
                                   dtype=dtype or p_.dtype, zero_init=zero_init)

    def forward(self, input_ids: Optional[torch.Tensor] = None, inputs_embeds: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.Tensor] = None, past_key_values: Optional[StaticCachePrealloc] = None,
                cache_position: Optional[torch.Tensor] = None, logits_to_keep: int = 1,
                layer_hooks=None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """layer_hooks = (before_mixer(i), after_mixer(i)) or None: called around the cache-touching part of
        decoder layer i (used by dist.sequence_parallel_prefill to receive / forward that layer's state)."""
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(input_ids)
        B, T, _ = inputs_embeds.shape
        if position_ids is None:
            start = past_key_values.get_seq_length() if past_key_values is not None else 0
            position_ids = torch.arange(start, start + T, device=inputs_embeds.device)[None, None, :].expand(3, B, T)
        position_embeddings = self.rotary_emb(inputs_embeds, position_ids)           # std:1549
        # The residual add that ends a decoder layer (std:1422) is fused into the NEXT layer's input RMSNorm
        # (one add+norm launch instead of add, norm): `resid` is the residual stream, `pend` the not-yet-added
        # MLP output of the previous layer.
        # Decode steps (<= 4 rows): the norms are not launched here -- a PreNorm travels into the mixer / MLP and runs in the
        # prologue of their first weight-stream kernel (ops.PreNorm).
        small = ops._PRENORM and inputs_embeds.is_cuda and inputs_embeds.dtype == torch.bfloat16 and B * T <= 4

        def norm(mod, x, residual):
            """(new residual stream, normalised input or its PreNorm)"""
            if small:
                pn = ops.PreNorm(x, residual, mod.weight, mod.variance_epsilon)
                return pn.h, pn
            if residual is None:
                return x, mod(x)
            return mod.add_and_norm(x, residual)

        resid, pend = inputs_embeds, None
        for layer in self.layers:                                                    # std:1555-1571
            if pend is None:
                resid, y = norm(layer.input_layernorm, resid, None)
            else:
                resid, y = norm(layer.input_layernorm, pend, resid)
            if layer_hooks is not None:
                layer_hooks[0](layer.self_attn.layer_idx)
            attn, _ = layer.self_attn(hidden_states=y, position_ids=position_ids, past_key_values=past_key_values,
                                      use_cache=past_key_values is not None, cache_position=cache_position,
                                      position_embeddings=position_embeddings)
            if layer_hooks is not None:
                layer_hooks[1](layer.self_attn.layer_idx)
            if isinstance(y, ops.PreNorm) and not y.done:
                y.materialize()                          # (a mixer path that never asked for its input: keep the residual valid)
            resid, y = norm(layer.post_attention_layernorm, attn, resid)
            pend = layer.mlp(y)
            if isinstance(y, ops.PreNorm) and not y.done:
                y.materialize()
        if pend is None:
            h = self.norm(resid)
        else:
            _, h = self.norm.add_and_norm(pend, resid)
        logits = None
        if logits_to_keep:
            logits = ops.linear(h[:, -logits_to_keep:, :], self.embed_tokens.weight)  # tied lm_head (std:2091-2092)
        return h, logits


_TEXT_PREFIXES = ("model.language_model.", "language_model.model.", "language_model.", "model.")


def text_state_dict_from_reference(tensors) -> Tuple[Dict[str, torch.Tensor], List[str]]:
    """Map the reference checkpoint's tensor names onto InfiniteVLTextStack's (std:1595-1618, 1975-1987): the text
    decoder lives under `model.language_model.` (current layout) or `model.` (Qwen2.5-VL legacy layout, remapped by
    `_checkpoint_conversion_mapping`); the vision tower (`visual.` / `model.visual.`) and the tied `lm_head.weight`
    are not part of the path.  Returns (state_dict for the stack, skipped names)."""
    out, skipped = {}, []
    for name, t in tensors.items():
        if name.startswith(("visual.", "model.visual.")) or name == "lm_head.weight":
            skipped.append(name)
            continue
        for pre in _TEXT_PREFIXES:
            if name.startswith(pre):
                out[name[len(pre):]] = t
                break
        else:
            skipped.append(name)
    return out, skipped


@torch.no_grad()
def load_reference_checkpoint(stack: "InfiniteVLTextStack", path: str) -> List[str]:
    """Load the text decoder of a reference checkpoint directory (`*.safetensors`, sharded or not) or file into
    `stack` (call BEFORE fuse_()).  Every text parameter must be present with the right shape; returns the names
    that were skipped (vision tower, tied lm_head)."""
    import glob
    from safetensors import safe_open
    files = sorted(glob.glob(_os.path.join(path, "*.safetensors"))) if _os.path.isdir(path) else [path]
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    tensors = {}
    for fpath in files:
        with safe_open(fpath, framework="pt", device="cpu") as f:
            for k in f.keys():
                tensors[k] = f.get_tensor(k)
    sd, skipped = text_state_dict_from_reference(tensors)
    own = stack.state_dict()
    missing = [k for k in own if k not in sd and "inv_freq" not in k]
    unexpected = [k for k in sd if k not in own]
    if missing or unexpected:
        raise KeyError(f"checkpoint does not match the text stack: missing {missing[:5]} unexpected {unexpected[:5]}")
    for k, v in sd.items():
        if tuple(v.shape) != tuple(own[k].shape):
            raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} != {tuple(own[k].shape)}")
        own[k].copy_(v.to(own[k].dtype))
    return skipped


def clone_inference_cache(cache: StaticCachePrealloc) -> StaticCachePrealloc:
    """demo:111-160."""
    return cache.clone()


class GraphedStep:
    """One hipGraph-captured forward of fixed length T over static input buffers
    (demo:262-269, 473-489).  The cache carries device-resident counters, so the same graph is valid
    for every step; positions advance inside the graph."""

    def __init__(self, model: InfiniteVLTextStack, cache: StaticCachePrealloc, batch_size: int, T: int,
                 logits_to_keep: int = 1, warmup: int = 2):
        p_ = next(model.parameters())
        self.model, self.cache, self.T, self.B = model, cache, T, batch_size
        self.inputs_embeds = torch.zeros(batch_size, T, model.config.hidden_size, dtype=p_.dtype, device=p_.device)
        start = cache.get_seq_length()
        self.position_ids = (torch.arange(start, start + T, device=p_.device, dtype=torch.int64)[None, None, :]
                             .expand(3, batch_size, T).contiguous())
        self.logits_to_keep = logits_to_keep
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.hidden = self.logits = None
        self._warmup = warmup
        self._sync: Optional[torch.Tensor] = None

    def _run(self):
        h, lg = self.model(inputs_embeds=self.inputs_embeds, position_ids=self.position_ids,
                           past_key_values=self.cache, logits_to_keep=self.logits_to_keep)
        self.position_ids.add_(self.T)
        return h, lg

    def capture(self) -> None:
        """Warm up on a side stream with a throw-away clone of the cache state (demo:285-315), then
        capture.  The live cache is restored afterwards, so capture has no side effects on it."""
        self.cache.ensure_started()          # a replayed graph always reads the cache tensors
        saved = self.cache.clone()
        saved_pos = self.position_ids.clone()
        # the graph owns the flag words of its single-launch GDN calls (ops.gdn_sync_scope): it may be replayed beside eager
        # calls or other graphs without sharing them
        if self._sync is None:
            self._sync = ops.new_gdn_sync_area(self.inputs_embeds.device)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with ops.gdn_sync_scope(self._sync):
            with torch.cuda.stream(s), torch.no_grad():
                for _ in range(self._warmup):
                    self._run()
            torch.cuda.current_stream().wait_stream(s)
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self.hidden, self.logits = self._run()
        self.cache.copy_from(saved)
        self.position_ids.copy_(saved_pos)

    def reset(self) -> None:
        """Start a new sequence on the same graph: cache counters and state tensors back to empty, positions to 0."""
        self.cache.reset()
        self.cache.ensure_started()
        self.position_ids.copy_(torch.arange(self.T, device=self.position_ids.device)[None, None, :]
                                .expand(3, self.B, self.T))

    def step(self, inputs_embeds: Optional[torch.Tensor] = None):
        if self.graph is None:
            self.capture()
        self.cache.ensure_started()          # a replayed graph always reads the cache tensors (no-op once started)
        if inputs_embeds is not None:
            self.inputs_embeds.copy_(inputs_embeds)
        self.graph.replay()
        self.cache.advance(self.T)
        ops.gdn_sync_check(self.inputs_embeds.device)    # free (a host word): a failed in-launch wait of an EARLIER replay raises here
        return self.hidden, self.logits


class GraphedDecode:
    """hipGraph-captured greedy single-token step: embed(token) -> stack -> argmax -> token buffer
    (the demo's eager loop, demo:399-422, made replayable: token and position stay on the device)."""

    def __init__(self, model: InfiniteVLTextStack, cache: StaticCachePrealloc, batch_size: int, warmup: int = 2):
        p_ = next(model.parameters())
        self.model, self.cache, self.B = model, cache, batch_size
        self.token = torch.zeros(batch_size, 1, dtype=torch.int64, device=p_.device)
        start = cache.get_seq_length()
        self.position_ids = torch.full((3, batch_size, 1), start, dtype=torch.int64, device=p_.device)
        self.logits = None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._warmup = warmup

    def _run(self):
        _, lg = self.model(input_ids=self.token, position_ids=self.position_ids, past_key_values=self.cache,
                           logits_to_keep=1)
        self.token.copy_(lg[:, -1].argmax(-1, keepdim=True))
        self.position_ids.add_(1)
        return lg

    def capture(self) -> None:
        self.cache.ensure_started()
        saved, saved_pos, saved_tok = self.cache.clone(), self.position_ids.clone(), self.token.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(self._warmup):
                self._run()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.logits = self._run()
        self.cache.copy_from(saved)
        self.position_ids.copy_(saved_pos)
        self.token.copy_(saved_tok)

    def step(self):
        """One decode step; the new token is left in self.token (device)."""
        if self.graph is None:
            self.capture()
        self.cache.ensure_started()
        self.graph.replay()
        self.cache.advance(1)
        return self.token


@torch.no_grad()
def greedy_decode(model: InfiniteVLTextStack, cache: StaticCachePrealloc, first_token: torch.Tensor, steps: int,
                  start_pos: Optional[int] = None) -> torch.Tensor:
    """Eager greedy loop of single-token forwards (demo:399-422)."""
    B = first_token.shape[0]
    pos = cache.get_seq_length() if start_pos is None else start_pos
    tok = first_token.view(B, 1)
    out = []
    for _ in range(steps):
        pid = torch.full((3, B, 1), pos, device=tok.device, dtype=torch.int64)
        _, logits = model(input_ids=tok, position_ids=pid, past_key_values=cache)
        tok = logits[:, -1].argmax(-1, keepdim=True)
        out.append(tok)
        pos += 1
    return torch.cat(out, dim=1)
