"""infinitevl_amd -- MI355X (gfx950) native kernels + drop-in modules for InfiniteVL's
hybrid-attention hot path (Gated DeltaNet chunk/recurrent rule + sliding-window attention).

Layout
    csrc/        hand-written HIP kernels + the C ABI (include/ivl_hip.h) -> libivl_hip.so
    _lib.py      ctypes binding (no fallback: a missing library raises)
    ops.py       operator-level API with the reference's operator names
    cache.py     StaticCachePrealloc & friends (ring-buffer / device-counter redesign)
    modules.py   GatedDeltaNet, InfiniteVLSelfAttention drop-in nn.Modules
    harness.py   decoder stack, hipGraph streaming step, greedy decode, cache clone
    dist.py      batch-sharded multi-GPU driver (RCCL all-gather of the last-token logits)
"""
from . import _lib
from .cache import StaticCachePrealloc, StaticLinearLayerPrealloc, StaticSlidingWindowLayerPrealloc
from .modules import GatedDeltaNet, InfiniteVLRotaryEmbedding, InfiniteVLSelfAttention
from .ops import (FusedRMSNormGated, RMSNorm, ShortConvolution, apply_mrope_inplace, chunk_gated_delta_rule,
                  fused_recurrent_gated_delta_rule, gdn_gate, get_unpad_data, index_first_axis, pad_input,
                  swa_attention_interface, swa_forward)

__all__ = [
    "StaticCachePrealloc", "StaticLinearLayerPrealloc", "StaticSlidingWindowLayerPrealloc",
    "GatedDeltaNet", "InfiniteVLSelfAttention", "InfiniteVLRotaryEmbedding",
    "FusedRMSNormGated", "RMSNorm", "ShortConvolution", "get_unpad_data", "index_first_axis", "pad_input", "chunk_gated_delta_rule", "fused_recurrent_gated_delta_rule",
    "gdn_gate", "apply_mrope_inplace", "swa_attention_interface", "swa_forward", "load_library",
]


def load_library():
    """Load libivl_hip.so now (raises ImportError with build instructions if it is missing)."""
    return _lib.load()
