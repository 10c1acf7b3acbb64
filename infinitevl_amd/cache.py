"""Static inference cache with the reference's class names, constructor arguments, methods and
attribute names (std:66-443; SURVEY.md section 8b "Cache contract"), re-designed for MI355X:

  * the sliding-window K/V store is a RING BUFFER (token with absolute position p lives in slot
    p % capacity) that the attention kernel reads in place -- no torch.cat of (cache, new), no
    tail copy-back (std:142-171 moves up to 2*(W-1)*Hkv*d elements per layer per call);
  * the position counter also lives ON THE DEVICE (`_pos_dev`), so a hipGraph captured once is valid
    for every later step (fixes the frozen-geometry replay of the reference);
  * the Gated DeltaNet state tensors are written by the kernels directly (no copy_ round trips).

Host-side integers (`size`, `cumulative_length`, `seq_len`, `start`) keep the reference semantics
bit-exactly; after a hipGraph replay (which skips the Python bookkeeping) call `advance(T)`.

`std:` = infinitevl/infinitevl_standard/modeling_infinitevl.py of the reference.
"""
from __future__ import annotations

import copy
from typing import Any, Optional, Tuple

import torch

try:  # inherit from HF's cache classes when transformers is present so `generate()` accepts the cache
    from transformers.cache_utils import Cache as _HFCache
    from transformers.cache_utils import CacheLayerMixin as _HFLayer
except Exception:  # pragma: no cover - transformers is optional for the kernels themselves
    _HFCache, _HFLayer = object, object

from . import ops


def _get_decoder_cfg(config):
    if hasattr(config, "get_text_config"):
        try:
            return config.get_text_config(decoder=True)
        except TypeError:
            return config.get_text_config()
    return config


class StaticSlidingWindowLayerPrealloc(_HFLayer):
    """SWA layer cache: keeps the last W-1 keys/values (std:66-227)."""
    is_sliding = True

    def __init__(self, *, config, batch_size: int, device="cpu", dtype: torch.dtype = torch.float32,
                 zero_init: bool = False):
        if _HFLayer is not object:
            super().__init__()
        cfg = _get_decoder_cfg(config)
        num_kv_heads = int(getattr(cfg, "num_key_value_heads", getattr(cfg, "num_attention_heads")))
        head_dim = getattr(cfg, "head_dim", None) or int(cfg.hidden_size) // int(cfg.num_attention_heads)
        W = (getattr(cfg, "sliding_window", None) or getattr(cfg, "attention_chunk_size", None)
             or int(getattr(cfg, "max_position_embeddings")))
        if W is None or int(W) <= 0:
            raise ValueError("SWA requires valid sliding_window / attention_chunk_size / max_position_embeddings")
        W = int(W)
        self.sliding_window = W
        self.capacity = max(W - 1, 0)
        self.is_initialized = True
        self.dtype, self.device = dtype, device
        self.batch_size = int(batch_size)
        self.num_kv_heads, self.head_dim = num_kv_heads, int(head_dim)
        self.size = 0
        self.cumulative_length = 0
        shape = (self.batch_size, self.num_kv_heads, self.capacity, self.head_dim)
        alloc = torch.zeros if zero_init else torch.empty
        if self.capacity > 0:
            self._buf_keys = alloc(shape, dtype=dtype, device=device)       # RING storage
            self._buf_values = alloc(shape, dtype=dtype, device=device)
        else:
            self._buf_keys = self._buf_values = None
        self._pos_dev = torch.zeros(1, dtype=torch.int64, device=device)     # == cumulative_length, on device
        # Inside a StaticCachePrealloc all sliding layers share ONE device counter (they always hold the same value) and
        # only the last of them advances it, once per forward: one counter launch per step instead of one per layer.
        self._advances_counter = True

    # ---- chronological views (reference attribute names; materialised on demand) -------------
    def _chronological(self, buf: torch.Tensor) -> torch.Tensor:
        if self.capacity == 0 or self.size == 0:
            return buf[:, :, :0, :] if buf is not None else torch.empty(
                (self.batch_size, self.num_kv_heads, 0, self.head_dim), dtype=self.dtype, device=self.device)
        start = (self.cumulative_length - self.size) % self.capacity
        idx = (torch.arange(self.size, device=buf.device) + start) % self.capacity
        return buf.index_select(2, idx)

    @property
    def keys(self) -> torch.Tensor:
        return self._chronological(self._buf_keys)

    @keys.setter
    def keys(self, _value) -> None:     # HF's CacheLayerMixin.__init__ assigns None; the view is derived
        pass

    @property
    def values(self) -> torch.Tensor:
        return self._chronological(self._buf_values)

    @values.setter
    def values(self, _value) -> None:
        pass

    # ---- fast path used by InfiniteVLSelfAttention -----------------------------------------------
    def _prev_cache(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Read-only chronological view of the cached keys / values (std:122-124)."""
        return self.keys, self.values

    def attend(self, q: torch.Tensor, k_new: torch.Tensor, v_new: torch.Tensor, scaling: float,
               window: Optional[int], mma_dtype=None, rope=None) -> torch.Tensor:
        """q [B,T,Hq,d], k_new/v_new [B,T,Hkv,d] (time-major; post-RoPE, or the raw projections together with
        rope=(cos, sin, mrope_section): the kernels then rotate q / k while loading them).  Attention over
        (ring ++ new) with the band of SURVEY.md section 8a S2, then append + advance.  Returns [B,T,Hq,d]."""
        B, T, Hkv, D = k_new.shape
        if B != self.batch_size:
            raise ValueError(f"SWA pre-allocated batch_size={self.batch_size}, but got B={B}")
        if Hkv != self.num_kv_heads or D != self.head_dim:
            raise ValueError(f"SWA head dim mismatch: got H={Hkv},D={D}, expect H={self.num_kv_heads},D={self.head_dim}")
        # attention over (ring ++ new), then the append: one call (the append rides in the split-KV combine launch)
        o = ops.swa_forward(q, k_new, v_new, window=window, scaling=scaling,
                            k_cache=self._buf_keys, v_cache=self._buf_values, pos_dev=self._pos_dev, mma_dtype=mma_dtype,
                            rope=rope, append=self.capacity > 0, pos_min=self.cumulative_length)
        if self._advances_counter:
            ops.counter_add(self._pos_dev, T)
        self.advance(T)
        return o

    def advance(self, T: int) -> None:
        """Host-side counters for a call of T tokens (std:147, 171-172)."""
        self.size = int(min(self.capacity, self.size + T))
        self.cumulative_length += int(T)

    # ---- reference-compatible slow path -------------------------------------------------------------
    def update(self, key_states: torch.Tensor, value_states: torch.Tensor, conv_state=None, recurrent_state=None,
               cache_kwargs: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Returns cat(cached, new) like std:126-173 (allocates; kept for API compatibility)."""
        assert key_states.shape == value_states.shape, "K/V shapes must match"
        B, H, Tq, D = key_states.shape
        if B != self.batch_size:
            raise ValueError(f"SWA pre-allocated batch_size={self.batch_size}, but got B={B}")
        if H != self.num_kv_heads or D != self.head_dim:
            raise ValueError(f"SWA head dim mismatch: got H={H},D={D}, expect H={self.num_kv_heads},D={self.head_dim}")
        full_k = torch.cat([self.keys, key_states], dim=-2)
        full_v = torch.cat([self.values, value_states], dim=-2)
        if self.capacity > 0:
            ops.swa_cache_append(key_states.transpose(1, 2), value_states.transpose(1, 2),
                                 self._buf_keys, self._buf_values, pos_dev=self._pos_dev)
        if self._advances_counter:
            ops.counter_add(self._pos_dev, Tq)
        self.advance(Tq)
        return full_k, full_v

    def get_mask_sizes(self, cache_position: torch.Tensor) -> Tuple[int, int]:
        q_len = int(cache_position.shape[0])                                # std:175-184
        pre_cum = max(int(self.cumulative_length) - q_len, 0)
        kv_offset = max(pre_cum - self.sliding_window + 1, 0)
        if pre_cum >= self.sliding_window:
            kv_len = (self.sliding_window - 1) + q_len
        else:
            kv_len = pre_cum + q_len
        return kv_len, kv_offset

    def get_seq_length(self, *a, **k) -> int:
        return int(self.cumulative_length)

    def get_max_cache_shape(self) -> int:
        return int(self.sliding_window)

    def get_max_length(self) -> int:          # transformers >= 5 abstract method
        return int(self.sliding_window)

    def crop(self, max_length: int) -> None:
        """std:192-213: keeps the LAST new_size cached tokens, moved to the front, and restarts the counters at
        new_size.  Only allowed while the window is not full, where ring slot == absolute position."""
        if self.get_seq_length() >= self.sliding_window:                    # std:192-194
            raise ValueError("Cropping is forbidden after filling SWA window (to avoid state loss)")
        new_size = max(0, self.size - abs(max_length)) if max_length < 0 else min(self.size, max_length)
        if self.capacity > 0 and 0 < new_size < self.size:
            lo = self.size - new_size
            self._buf_keys[:, :, :new_size, :].copy_(self._buf_keys[:, :, lo:self.size, :].clone())
            self._buf_values[:, :, :new_size, :].copy_(self._buf_values[:, :, lo:self.size, :].clone())
        self.size = int(new_size)
        self.cumulative_length = int(new_size)
        self._pos_dev.fill_(int(new_size))

    def batch_repeat_interleave(self, repeats: int) -> None:
        if repeats != 1:
            raise RuntimeError("Static cache forbids changing batch size (repeat_interleave)")

    def batch_select_indices(self, indices: torch.Tensor) -> None:
        if indices.numel() != self.batch_size:
            raise RuntimeError("Static cache forbids changing batch size (select_indices)")

    def lazy_initialization(self, *args, **kwargs):
        return

    def reset(self) -> None:
        self.size = 0
        self.cumulative_length = 0
        self._pos_dev.zero_()

    # ---- state hand-off (sequence-parallel prefill) ------------------------------------
    def carried_tensors(self):
        """Device tensors that define what the NEXT tokens of the sequence need from this layer: the ring (last W-1
        post-RoPE keys / values in slot order).  The position counter is not carried (inside an aggregate cache it is
        shared by all sliding layers and advanced once per forward): import_carried() sets it from `seen_tokens`."""
        return [t for t in (self._buf_keys, self._buf_values) if t is not None]

    def import_carried(self, seen_tokens: int) -> None:
        """Counters (host and device) after carried_tensors() were overwritten with the state of a sequence prefix of
        `seen_tokens` tokens."""
        self.size = int(min(self.capacity, seen_tokens))
        self.cumulative_length = int(seen_tokens)
        self._pos_dev.fill_(int(seen_tokens))

    def clone(self) -> "StaticSlidingWindowLayerPrealloc":
        """Deep copy (what the demo's clone_inference_cache does, demo:123-146)."""
        new = copy.copy(self)
        if self._buf_keys is not None:
            new._buf_keys = self._buf_keys.clone()
            new._buf_values = self._buf_values.clone()
        new._pos_dev = self._pos_dev.clone()
        # the clone's counter is private until an aggregate re-shares it (StaticCachePrealloc.clone does): a layer
        # cloned or driven on its own must advance it itself, or ring slots and band drift from the host counters
        new._advances_counter = True
        return new

    def copy_from(self, other: "StaticSlidingWindowLayerPrealloc") -> None:
        """In-place state copy (keeps tensor addresses: hipGraph-friendly branch/restore)."""
        if self._buf_keys is not None:
            self._buf_keys.copy_(other._buf_keys)
            self._buf_values.copy_(other._buf_values)
        self._pos_dev.copy_(other._pos_dev)
        self.size, self.cumulative_length = other.size, other.cumulative_length


class StaticLinearLayerPrealloc(_HFLayer):
    """Gated DeltaNet layer cache: conv states [B,D,W] + recurrent state [B,H,K,V] in the cache dtype
    (std:229-364)."""
    is_sliding = False

    def __init__(self, *, config, batch_size: int, device="cpu", dtype: torch.dtype = torch.float32,
                 zero_init: bool = False, recurrent_state_shape: Optional[Tuple[int, ...]] = None):
        if _HFLayer is not object:
            super().__init__()
        cfg = _get_decoder_cfg(config)
        self.num_linear_heads = int(getattr(cfg, "num_linear_heads", getattr(cfg, "num_attention_heads")))
        self.num_linear_kv_heads = int(getattr(cfg, "num_linear_key_value_heads", self.num_linear_heads))
        self.linear_head_dim = int(getattr(cfg, "linear_head_dim", getattr(cfg, "head_dim", 0) or 0))
        self.conv_size = int(getattr(cfg, "conv_size", 1))
        self.use_short_conv = bool(getattr(cfg, "use_short_conv", True))
        expand_v = float(getattr(cfg, "expand_v", 1.0))
        self.v_head_dim = int(round(self.linear_head_dim * expand_v))
        self.is_initialized = True
        self.dtype, self.device = dtype, device
        self.batch_size = int(batch_size)
        self.seq_len = 0
        self.start = False
        alloc = torch.zeros if zero_init else torch.empty
        B, Hq, Hk, C, Cv, K = (self.batch_size, self.num_linear_heads, self.num_linear_kv_heads,
                               self.linear_head_dim, self.v_head_dim, self.conv_size)
        if self.use_short_conv:
            self.conv_state_q = alloc((B, Hq * C, K), dtype=dtype, device=device)
            self.conv_state_k = alloc((B, Hk * C, K), dtype=dtype, device=device)
            self.conv_state_v = alloc((B, Hk * Cv, K), dtype=dtype, device=device)
        else:
            self.conv_state_q = self.conv_state_k = self.conv_state_v = None
        if recurrent_state_shape is None:
            recurrent_state_shape = (B, Hq, C, Cv)
        else:
            assert recurrent_state_shape[0] == B, "recurrent_state_shape batch dim must match pre-allocated batch_size"
        self.recurrent_state = alloc(recurrent_state_shape, dtype=dtype, device=device)

    def update(self, key_states=None, value_states=None, conv_state=None, recurrent_state=None,
               cache_kwargs: Optional[dict] = None) -> tuple:
        """get/set protocol of std:286-338 (first call returns Nones; shape-guarded in-place copies).
        A tensor that already IS the cache storage (kernels write in place) is not copied again."""
        if cache_kwargs is None:
            cache_kwargs = {}
        op = cache_kwargs.get("op", "get" if (conv_state is None and recurrent_state is None) else "set")
        if self.start is False:
            self.start = True
            return (None, None, None), None
        if op == "get":
            return (self.conv_state_q, self.conv_state_k, self.conv_state_v), self.recurrent_state
        if conv_state is not None and self.use_short_conv:
            assert isinstance(conv_state, (tuple, list)), "conv_state must be (cq, ck, cv)"
            cq, ck, cv = (tuple(conv_state) + (None, None, None))[:3]
            for name, src in (("q", cq), ("k", ck), ("v", cv)):
                dst = getattr(self, "conv_state_" + name)
                if src is None:
                    continue
                if tuple(src.shape) != tuple(dst.shape):
                    raise RuntimeError(f"conv_{name} shape changed: got {tuple(src.shape)} vs prealloc {tuple(dst.shape)}")
                if src.data_ptr() != dst.data_ptr():
                    dst.copy_(src)
        elif conv_state is not None and not self.use_short_conv:
            raise RuntimeError("config.use_short_conv=False, but conv_state was passed")
        if recurrent_state is not None:
            if tuple(recurrent_state.shape) != tuple(self.recurrent_state.shape):
                raise RuntimeError(f"recurrent_state shape changed: got {tuple(recurrent_state.shape)} vs prealloc "
                                   f"{tuple(self.recurrent_state.shape)}")
            if recurrent_state.data_ptr() != self.recurrent_state.data_ptr():
                self.recurrent_state.copy_(recurrent_state)
        self.seq_len += int(cache_kwargs.get("delta_len", 0))
        return (self.conv_state_q, self.conv_state_k, self.conv_state_v), self.recurrent_state

    def advance(self, T: int) -> None:
        self.start = True
        self.seq_len += int(T)

    def ensure_started(self) -> None:
        """Make the pre-allocated tensors equal to what the reference's first call uses (zero history,
        std:298-300) and flip `start`: afterwards a captured hipGraph, which always reads the cache tensors,
        computes the same thing as the eager first call that ignores them."""
        if not self.start:
            for n in ("conv_state_q", "conv_state_k", "conv_state_v", "recurrent_state"):
                t = getattr(self, n)
                if t is not None:
                    t.zero_()
            self.start = True

    def get_mask_sizes(self, cache_position: torch.Tensor) -> Tuple[int, int]:
        qlen = cache_position.shape[0] if cache_position is not None else 0
        return self.get_seq_length() + qlen, 0

    def get_seq_length(self, *a, **k) -> int:
        return int(self.seq_len)

    def get_max_cache_shape(self) -> int:
        return -1

    def get_max_length(self) -> int:
        return -1

    def crop(self, max_length: int) -> None:
        if max_length < 0:
            max_length = max(0, self.get_seq_length() - abs(max_length))
        self.seq_len = min(self.get_seq_length(), max_length)

    def batch_repeat_interleave(self, repeats: int) -> None:
        if repeats != 1:
            raise RuntimeError("Static cache forbids changing batch size (repeat_interleave)")

    def batch_select_indices(self, indices: torch.Tensor) -> None:
        if indices.numel() != self.batch_size:
            raise RuntimeError("Static cache forbids changing batch size (select_indices)")

    def lazy_initialization(self, *args, **kwargs):
        return

    def reset(self) -> None:
        """Back to the state of a fresh cache.  The tensors are zeroed as well: an eager first call ignores them
        (std:298-300), but a captured hipGraph always reads them, and must then see the same zero history."""
        self.seq_len = 0
        self.start = False
        for n in ("conv_state_q", "conv_state_k", "conv_state_v", "recurrent_state"):
            t = getattr(self, n)
            if t is not None:
                t.zero_()

    # ---- state hand-off (sequence-parallel prefill) ------------------------------------
    def carried_tensors(self):
        """conv states [B,D,4] x3 + recurrent state [B,H,K,V]: everything the next tokens need from this layer."""
        return [t for t in (self.conv_state_q, self.conv_state_k, self.conv_state_v, self.recurrent_state)
                if t is not None]

    def import_carried(self, seen_tokens: int) -> None:
        self.seq_len = int(seen_tokens)
        self.start = True            # the tensors hold a real prefix state: never take the first-call shortcut (Q4)

    def clone(self) -> "StaticLinearLayerPrealloc":
        new = copy.copy(self)
        for n in ("conv_state_q", "conv_state_k", "conv_state_v", "recurrent_state"):
            t = getattr(self, n)
            if t is not None:
                setattr(new, n, t.clone())
        return new

    def copy_from(self, other: "StaticLinearLayerPrealloc") -> None:
        for n in ("conv_state_q", "conv_state_k", "conv_state_v", "recurrent_state"):
            t = getattr(self, n)
            if t is not None:
                t.copy_(getattr(other, n))
        self.seq_len, self.start = other.seq_len, other.start


class DynamicLayer(_HFLayer):
    """Growing K/V cache for a full-attention layer (strm:67-157): the fallback the aggregate cache uses for layer
    types that are neither sliding nor linear.  InfiniteVL's shipped config has none; kept so that the dispatch by
    `layer_types` is total.  Pure tensor bookkeeping (torch.cat), the attention itself runs through
    `swa_attention_interface` with window=None (plain causal)."""
    is_sliding = False

    def __init__(self):
        if _HFLayer is not object:
            super().__init__()
        self._k: Optional[torch.Tensor] = None
        self._v: Optional[torch.Tensor] = None
        self.is_initialized = False

    @property
    def keys(self):
        return self._k

    @keys.setter
    def keys(self, value):            # HF's CacheLayerMixin.__init__ assigns None
        self._k = value

    @property
    def values(self):
        return self._v

    @values.setter
    def values(self, value):
        self._v = value

    def lazy_initialization(self, key_states: torch.Tensor, *a, **k):
        self.dtype, self.device = key_states.dtype, key_states.device
        self._k = key_states.new_empty(0)
        self._v = key_states.new_empty(0)
        self.is_initialized = True

    def update(self, key_states, value_states, conv_state=None, recurrent_state=None, cache_kwargs=None):
        if not self.is_initialized:
            self.lazy_initialization(key_states)
        self._k = key_states if self._k.numel() == 0 else torch.cat([self._k, key_states], dim=-2)
        self._v = value_states if self._v.numel() == 0 else torch.cat([self._v, value_states], dim=-2)
        return self._k, self._v

    def get_mask_sizes(self, cache_position: torch.Tensor) -> Tuple[int, int]:
        return self.get_seq_length(), 0

    def get_seq_length(self, *a, **k) -> int:
        return 0 if (not self.is_initialized or self._k.numel() == 0) else int(self._k.shape[-2])

    def get_max_cache_shape(self) -> int:
        return -1

    def get_max_length(self) -> int:
        return -1

    def crop(self, max_length: int) -> None:
        if max_length < 0:
            max_length = self.get_seq_length() - abs(max_length)
        if self.get_seq_length() <= max_length:
            return
        self._k, self._v = self._k[..., :max_length, :], self._v[..., :max_length, :]

    def batch_repeat_interleave(self, repeats: int) -> None:
        if self.get_seq_length() > 0:
            self._k, self._v = self._k.repeat_interleave(repeats, dim=0), self._v.repeat_interleave(repeats, dim=0)

    def batch_select_indices(self, indices: torch.Tensor) -> None:
        if self.get_seq_length() > 0:
            self._k, self._v = self._k[indices, ...], self._v[indices, ...]

    def advance(self, T: int) -> None:
        pass

    def reset(self) -> None:
        self._k = self._v = None
        self.is_initialized = False

    def carried_tensors(self):
        return [t for t in (self._k, self._v) if t is not None]

    def import_carried(self, seen_tokens: int) -> None:
        pass

    def clone(self) -> "DynamicLayer":
        new = copy.copy(self)
        if self._k is not None:
            new._k, new._v = self._k.clone(), self._v.clone()
        return new

    def copy_from(self, other: "DynamicLayer") -> None:
        self._k = None if other._k is None else other._k.clone()
        self._v = None if other._v is None else other._v.clone()
        self.is_initialized = other.is_initialized


class StaticCachePrealloc(_HFCache):
    """Aggregate cache: one pre-allocated layer object per decoder layer, dispatched by
    `config.layer_types` (std:366-443)."""

    def __init__(self, *, config, batch_size: int = 1, device="cpu", dtype: torch.dtype = torch.float32,
                 zero_init: bool = False, recurrent_state_shape: Optional[Tuple[int, ...]] = None,
                 offloading: bool = False, offload_only_non_sliding: bool = False):
        layers = []
        cfg = _get_decoder_cfg(config)
        layer_types = getattr(cfg, "layer_types", None)
        if layer_types is None:
            layer_types = ["linear_attention"] * int(getattr(cfg, "num_hidden_layers"))
        if hasattr(cfg, "num_kv_shared_layers"):
            layer_types = layer_types[: -int(getattr(cfg, "num_kv_shared_layers"))]
        for lt in layer_types:
            if lt in ("sliding_attention", "chunked_attention"):
                layers.append(StaticSlidingWindowLayerPrealloc(config=cfg, batch_size=batch_size, device=device,
                                                               dtype=dtype, zero_init=zero_init))
            elif lt in ("linear_attention", "delta_net", "retnet", "state_space"):
                layers.append(StaticLinearLayerPrealloc(config=cfg, batch_size=batch_size, device=device, dtype=dtype,
                                                        zero_init=zero_init,
                                                        recurrent_state_shape=recurrent_state_shape))
            else:                                   # full attention: dynamic cache, as strm:548-550
                layers.append(DynamicLayer())
        if _HFCache is not object:
            try:
                super().__init__(layers=layers, offloading=offloading, offload_only_non_sliding=offload_only_non_sliding)
            except TypeError:
                super().__init__(layers=layers)
        self.layers = layers
        self.layer_types = list(layer_types)
        self._share_position_counter()

    def _share_position_counter(self) -> None:
        """All sliding layers of one cache see the same number of tokens: point them at ONE device counter, advanced by
        the last sliding layer only (the decoder runs the layers in order, so every layer has read it by then)."""
        sliding = [l for l in self.layers if isinstance(l, StaticSlidingWindowLayerPrealloc)]
        for i, layer in enumerate(sliding):
            if i > 0:
                layer._pos_dev = sliding[0]._pos_dev
            layer._advances_counter = i == len(sliding) - 1

    def update(self, layer_idx: int, key_states=None, value_states=None, conv_state=None, recurrent_state=None,
               cache_kwargs: Optional[dict[str, Any]] = None):
        return self.layers[layer_idx].update(key_states, value_states, conv_state, recurrent_state, cache_kwargs)

    def get_seq_length(self, layer_idx: int = 0, *a, **k) -> int:
        if not self.layers:
            return 0
        return self.layers[layer_idx].get_seq_length()

    def advance(self, T: int) -> None:
        """Host-side bookkeeping for one forward of T tokens executed by a hipGraph replay."""
        for layer in self.layers:
            layer.advance(T)

    def reset(self) -> None:
        for layer in self.layers:
            layer.reset()

    def ensure_started(self) -> None:
        for layer in self.layers:
            if hasattr(layer, "ensure_started"):
                layer.ensure_started()

    def clone(self) -> "StaticCachePrealloc":
        new = copy.copy(self)
        new.layers = [layer.clone() for layer in self.layers]
        new._share_position_counter()
        return new

    def copy_from(self, other: "StaticCachePrealloc") -> None:
        for dst, src in zip(self.layers, other.layers):
            dst.copy_from(src)

    def to_legacy_cache(self):
        return tuple((getattr(l, "keys", None), getattr(l, "values", None)) for l in self.layers)

    def memory_bytes(self) -> int:
        total = 0
        for layer in self.layers:
            for t in vars(layer).values():
                if torch.is_tensor(t):
                    total += t.numel() * t.element_size()
        return total
