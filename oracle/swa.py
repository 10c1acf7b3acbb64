"""Oracle (CPU) for the sliding-window-attention side of the hot path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
`std:` = /root/reference/infinitevl/infinitevl_standard/modeling_infinitevl.py.

Integer parts (window bounds) are numpy int64 and must match bit-exactly;
floating-point parts are fp32 torch.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch


# ----------------------------------------------------------------------------
# integers: which keys a query row may see
# ----------------------------------------------------------------------------
def n_prev_keys(window: int, tokens_seen_before: int) -> int:
    """Number of cached keys visible to a call: the cache keeps the last W-1
    pre-call keys (std:93, std:147)."""
    return int(min(window - 1, tokens_seen_before))


def window_bounds(n_prev: int, T: int, window: int) -> Tuple[np.ndarray, np.ndarray]:
    """Call-local inclusive key range [lo(i), hi(i)] for query row i in [0,T).

    Keys are indexed over cat(cached n_prev keys, T new keys).  Query at absolute
    position p sees keys j with p-W < j <= p that are still cached:
        hi(i) = n_prev + i,  lo(i) = max(0, n_prev + i - W + 1).
    This is FlashAttention-2's bottom-right aligned causal mask with
    window_size=(W-1, W-1) (transformers `_process_flash_attention_kwargs`: applied
    only when key_len > W, which changes nothing because for S <= W the band
    already covers [0,hi]) and HF `sliding_window_overlay` (kv > q - W) AND causal.
    """
    i = np.arange(T, dtype=np.int64)
    hi = n_prev + i
    lo = np.maximum(0, n_prev + i - window + 1)
    return lo, hi


def band_mask(n_prev: int, T: int, window: int) -> np.ndarray:
    """Boolean [T, n_prev+T] visibility mask from window_bounds."""
    lo, hi = window_bounds(n_prev, T, window)
    j = np.arange(n_prev + T, dtype=np.int64)[None, :]
    return (j >= lo[:, None]) & (j <= hi[:, None])


# ----------------------------------------------------------------------------
# floating point
# ----------------------------------------------------------------------------
def swa_attention(
    q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n_prev: int, window: int,
    scaling: Optional[float] = None, p_round_dtype: Optional[torch.dtype] = None,
    mma_rounding: Optional[torch.dtype] = None,
) -> torch.Tensor:
    """softmax_fp32(q k^T * scaling + band) v with GQA; the math of std:557-580.

    q [B,Hq,T,d]; k,v [B,Hkv,S,d] with S = n_prev + T (cached keys first).
    Returns [B,T,Hq,d] fp32 (the layout attention_interface returns, std:578).
    `p_round_dtype` reproduces that the probabilities are cast to the activation
    dtype before P@V (std:575).  `mma_rounding=torch.float8_e4m3fn` models the operand rounding of the build's fp8
    decode variant (BASELINE.json configs[4]; no counterpart in the reference): q, k and v are rounded to OCP e4m3
    before the two products (the probabilities are left exact: their e4m3 rounding depends on the tiling of the kernel).
    """
    B, Hq, T, d = q.shape
    Hkv, S = k.shape[1], k.shape[2]
    assert S == n_prev + T, (S, n_prev, T)
    if scaling is None:
        scaling = d ** -0.5
    rep = Hq // Hkv
    r8 = (lambda x: x) if mma_rounding is None else (lambda x: x.clamp(-448.0, 448.0).to(mma_rounding).float())
    kf = r8(k.float()).repeat_interleave(rep, dim=1)
    vf = r8(v.float()).repeat_interleave(rep, dim=1)
    scores = torch.matmul(r8(q.float()), kf.transpose(2, 3)) * scaling
    mask = torch.from_numpy(band_mask(n_prev, T, window))
    scores = scores.masked_fill(~mask[None, None], float("-inf"))
    p = torch.softmax(scores, dim=-1, dtype=torch.float32)
    if p_round_dtype is not None:
        p = p.to(p_round_dtype).float()
    out = torch.matmul(p, vf)
    return out.transpose(1, 2).contiguous()


def rotary_cos_sin(position_ids: torch.Tensor, head_dim: int, theta: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [3,B,T,head_dim] fp32 for 3-D (t,h,w) positions; std:918-930 with the
    'default' rope init inv_freq = theta^(-arange(0,d,2)/d)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    freqs = position_ids[..., None].float() * inv_freq          # [3,B,T,d/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def select_mrope(cos: torch.Tensor, mrope_section: List[int]) -> torch.Tensor:
    """Pick the t/h/w row per channel section: [3,B,T,d] -> [B,T,d]; std:974-980."""
    sec = list(mrope_section) * 2
    return torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1)


def apply_mrope(q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                mrope_section: List[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Multimodal rotary embedding on q [B,Hq,T,d], k [B,Hkv,T,d]; std:949-984."""
    c = select_mrope(cos, mrope_section).unsqueeze(1)
    s = select_mrope(sin, mrope_section).unsqueeze(1)
    return (q * c) + (rotate_half(q) * s), (k * c) + (rotate_half(k) * s)
