"""Host-side tensor plumbing with the reference's names and layouts (rl4co/utils/ops.py:13-95,
rl4co/data/transforms.py:16-153).  No arithmetic of the rollout happens here: these reshape / gather / replicate
tensors around the native calls (multistart "(s b)" layout, dihedral-8 instance augmentation)."""
from __future__ import annotations

import torch

from .tensordict_lite import TensorDict


def _batchify_single(x, repeats: int):
    s = x.shape
    return x.expand(repeats, *s).contiguous().view(s[0] * repeats, *s[1:])


def batchify(x, shape):
    """einops.repeat(x, 'b ... -> (r b) ...'): rows in (s b) order.  Works on tensors and TensorDicts."""
    shape = [shape] if isinstance(shape, int) else shape
    for s in reversed(shape):
        x = _batchify_single(x, s) if s > 0 else x
    return x


def _unbatchify_single(x, repeats: int):
    s = x.shape
    return x.view(repeats, s[0] // repeats, *s[1:]).permute(1, 0, *range(2, len(s) + 1))


def unbatchify(x, shape):
    """'(r b) ... -> b r ...'"""
    shape = [shape] if isinstance(shape, int) else shape
    for s in reversed(shape):
        x = _unbatchify_single(x, s) if s > 0 else x
    return x


def gather_by_index(src, idx, dim=1, squeeze=True):
    expanded = list(src.shape)
    expanded[dim] = -1
    idx = idx.view(idx.shape + (1,) * (src.dim() - idx.dim())).expand(expanded)
    squeeze = idx.size(dim) == 1 and squeeze
    out = src.gather(dim, idx)
    return out.squeeze(dim) if squeeze else out


def unbatchify_and_gather(x, idx, n: int):
    x = unbatchify(x, n)
    return gather_by_index(x, idx, dim=idx.dim())


def dihedral_8_augmentation(xy: torch.Tensor) -> torch.Tensor:
    """The 8 symmetries of the unit square applied to [B, N, 2] coordinates -> [8B, N, 2], identity first
    (POMO; rl4co/data/transforms.py:16-40)."""
    x, y = xy[..., 0:1], xy[..., 1:2]
    variants = [(x, y), (1 - x, y), (x, 1 - y), (1 - x, 1 - y), (y, x), (1 - y, x), (y, 1 - x), (1 - y, 1 - x)]
    return torch.cat([torch.cat(v, dim=-1) for v in variants], dim=0)


class StateAugmentation:
    """Instance augmentation of the coordinate features (dihedral-8 only; `symmetric` random rotations of the
    reference are not built).  td [B] -> td [num_augment * B] in (a b) order, identity first."""

    def __init__(self, num_augment: int = 8, augment_fn="dihedral8", first_aug_identity: bool = True, feats=None, **_):
        if augment_fn not in ("dihedral8", dihedral_8_augmentation):
            raise NotImplementedError("only the dihedral-8 augmentation is built")
        if num_augment != 8:
            raise NotImplementedError("dihedral-8 augmentation needs num_augment == 8")
        self.num_augment = num_augment
        self.feats = ["locs"] if feats is None else list(feats)

    def __call__(self, td):
        out = batchify(td, self.num_augment)
        if not isinstance(out, TensorDict):
            out = TensorDict(dict(out.items()), batch_size=out.batch_size)
        for f in self.feats:
            out.set(f, dihedral_8_augmentation(td[f]))
        return out
