"""Host mirror of the reference's rope.py API (rope.py:6-22).

`precompute_frequencies` reproduces the reference's float32 arithmetic with numpy; the table is
handed to the HIP library (amt_set_rope), which applies the rotation inside the attention
kernels (rope.py:25-53 is never a separate pass on the device).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class RopeFreqs:
    cos_freq: np.ndarray  # (max_pos, dim // 2) float32
    sin_freq: np.ndarray


def precompute_frequencies(dim: int, max_pos: int, theta: float = 10000.0) -> RopeFreqs:
    """rope.py:12-22."""
    inv_freq = (np.float32(1.0) / (np.float32(theta) ** (np.arange(0, dim, 2, dtype=np.float32)[: dim // 2]
                                                            / np.float32(dim)))).astype(np.float32)
    t = np.arange(0, max_pos, dtype=np.float32)
    freqs = np.outer(t, inv_freq).astype(np.float32)
    return RopeFreqs(cos_freq=np.cos(freqs).astype(np.float32), sin_freq=np.sin(freqs).astype(np.float32))
