"""Closed-form pseudo-random policy weights shared by make_golden.py and the tests.

No state_dict is stored in the fixtures: every tensor of the AttentionModelPolicy
state_dict (key names = the reference contract, SURVEY.md section 8a footnote) is filled
from a splitmix64 stream seeded by the FNV-1a hash of its key name, so the golden
generator (reference policy) and the tests (this repo's policy / C oracle) build
bit-identical weights independently.  Pure numpy, no reference code involved.
"""
from __future__ import annotations

import numpy as np

_M64 = (1 << 64) - 1


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h ^= b
        h = (h * 0x100000001B3) & _M64
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def unit(name: str, numel: int) -> np.ndarray:
    """numel values in [-1, 1) exactly representable in float32 (24-bit grid)."""
    seed = np.uint64(_fnv1a64(name))
    with np.errstate(over="ignore"):
        idx = np.arange(numel, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + seed
    k = (_splitmix64(idx) >> np.uint64(40)).astype(np.int64)  # 24 bits
    return ((2 * k - (1 << 24)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def tensor_for(name: str, shape) -> np.ndarray | None:
    """Golden value for one state_dict entry, or None to leave it untouched."""
    shape = tuple(int(s) for s in shape)
    numel = int(np.prod(shape)) if len(shape) else 1
    if name.endswith("num_batches_tracked"):
        return None
    u = unit(name, numel)
    if name.endswith("running_var"):
        v = (u * np.float32(0.5) + np.float32(1.0))          # [0.5, 1.5)
    elif name.endswith("running_mean"):
        v = u * np.float32(0.1)
    elif ".normalizer.weight" in name:
        v = u * np.float32(0.2) + np.float32(1.0)             # [0.8, 1.2)
    elif ".normalizer.bias" in name:
        v = u * np.float32(0.1)
    elif name.endswith("W_placeholder"):
        v = u
    elif name.endswith(".bias"):
        v = u * np.float32(0.1)
    elif len(shape) == 2:
        v = u * np.float32(1.0 / np.sqrt(np.float32(shape[1])))  # fan_in = shape[1]
    else:
        v = u * np.float32(0.1)
    return v.astype(np.float32).reshape(shape)


def fill_state_dict(sd) -> dict:
    """Return {key: np.ndarray} for every float entry of a (torch) state_dict-like mapping."""
    out = {}
    for k, t in sd.items():
        v = tensor_for(k, tuple(t.shape))
        if v is not None:
            out[k] = v
    return out
