"""Oracle for the static cache's INTEGER bookkeeping (bit-exact contract).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
Restates `StaticSlidingWindowLayerPrealloc` (std:66-227) and
`StaticLinearLayerPrealloc` (std:229-364) counters without any tensors.
`std:` = /root/reference/infinitevl/infinitevl_standard/modeling_infinitevl.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple


@dataclass
class SwaCounters:
    """size / cumulative_length / get_mask_sizes of the sliding-window layer."""
    window: int
    size: int = 0
    cumulative_length: int = 0

    @property
    def capacity(self) -> int:              # std:93
        return max(self.window - 1, 0)

    def update(self, T: int) -> int:
        """Advance by a call of T new tokens; returns len(full_k) = size_before + T
        (std:142-173)."""
        full_len = self.size + T
        self.size = min(self.capacity, self.size + T)
        self.cumulative_length += T
        return full_len

    def get_mask_sizes(self, q_len: int) -> Tuple[int, int]:
        """(kv_len, kv_offset) AFTER update(); std:175-184."""
        pre_cum = max(self.cumulative_length - q_len, 0)
        kv_offset = max(pre_cum - self.window + 1, 0)
        if pre_cum >= self.window:
            kv_len = (self.window - 1) + q_len
        else:
            kv_len = pre_cum + q_len
        return kv_len, kv_offset


def swa_trace(window: int, steps: List[int]):
    """For a sequence of call lengths return per call
    (full_len, size_after, cumulative_after, kv_len, kv_offset, n_prev)."""
    c = SwaCounters(window)
    out = []
    for T in steps:
        n_prev = c.size
        full_len = c.update(T)
        kv_len, kv_off = c.get_mask_sizes(T)
        out.append((full_len, c.size, c.cumulative_length, kv_len, kv_off, n_prev))
    return out


@dataclass
class LinearCounters:
    """seq_len / start flag of the linear (GDN) layer; std:298-300, 337."""
    seq_len: int = 0
    start: bool = False

    def get(self) -> bool:
        """Returns True when the cached state is handed out, False on the very first
        call (which flips `start` and returns Nones, std:298-300)."""
        if not self.start:
            self.start = True
            return False
        return True

    def set(self, delta_len: int) -> None:
        if not self.start:          # a first `set` is swallowed the same way (std:298-300)
            self.start = True
            return
        self.seq_len += delta_len
