"""Host-side tensor plumbing with the reference's names and layouts (rl4co/utils/ops.py:13-95,
rl4co/data/transforms.py:16-153).  No arithmetic of the rollout happens here: these reshape / gather / replicate
tensors around the native calls (multistart "(s b)" layout, dihedral-8 / symmetric instance augmentation)."""
from __future__ import annotations

import math

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


# ------------------------------------------------------------------------------------------------------------
# instance augmentation (rl4co/data/transforms.py:16-153) on eamrl_augment_xy
#
# Every augmentation here is a per-row map of the plane: row r of the batchified "(a b)" tensor is instance r % B under one
# of ten elementary transforms (ops.augment_xy: codes 0..7 = the symmetries of the unit square, 8 / 9 = rotation about the
# centre without / with the x <-> y swap).  The functions below only decide the per-row code and angle; the kernel applies
# them and performs the replication.  Names and call signatures are the reference's (they are what its evaluators and
# `StateAugmentation(augment_fn=...)` look up); a tensor already batchified by the caller is reduced to its first copy first.
# ------------------------------------------------------------------------------------------------------------
def _codes(values, device):
    return torch.tensor(values, dtype=torch.int32, device=device)


def _apply_codes(xy, codes, cs=None, copies=None):
    """xy [B, N, 2] (or any tensor whose first B rows are the instances) -> [len(codes), N, 2]."""
    from . import ops

    if xy.is_cuda:
        return ops.augment_xy(xy.contiguous(), codes, cs)
    raise RuntimeError("instance augmentation runs on the MI355X (eamrl_augment_xy); move the batch to the GPU first")


def dihedral_8_augmentation(xy: torch.Tensor) -> torch.Tensor:
    """[B, N, 2] -> [8B, N, 2]: copy a of the batch under the a-th symmetry of the unit square, identity first (POMO)."""
    B = xy.shape[0]
    return _apply_codes(xy, _codes([a for a in range(8) for _ in range(B)], xy.device))


def dihedral_8_augmentation_wrapper(xy: torch.Tensor, reduce: bool = True, *args, **kw) -> torch.Tensor:
    """The same for a tensor that is already 8 stacked copies: only its first eighth is read (reduce=False: all rows)."""
    return dihedral_8_augmentation(xy[: xy.shape[0] // 8] if reduce else xy)


def symmetric_augmentation(xy: torch.Tensor, num_augment: int = 8, first_augment: bool = False, phi: torch.Tensor = None):
    """SymNCO's augmentation of a batchified [A*B, N, 2] tensor: row r is rotated about (0.5, 0.5) by phi[r] in [0, 4 pi) and,
    where phi[r] > 2 pi, mirrored (x <-> y); unless first_augment, the first copy of the batch keeps phi = 0.  The angles are
    an input like the sampling noise (drawn with torch.rand on xy's device when None -- replaying a reference run means
    passing ITS angles); their cosines and sines are evaluated on the host in fp32, which is what makes the rows equal to the
    reference's CPU evaluation bit for bit."""
    R = xy.shape[0]
    if phi is None:
        phi = torch.rand(R, device=xy.device) * (4 * math.pi)
    ang = phi.detach().to(device="cpu", dtype=xy.dtype).clone()
    if not first_augment:
        ang[: R // num_augment] = 0.0
    cs = torch.stack((torch.cos(ang), torch.sin(ang)), 1).to(xy.device)
    codes = (8 + (ang > 2 * math.pi).to(torch.int32)).to(xy.device)
    return _apply_codes(xy, codes, cs)          # every row under its own angle (the kernel's replication is not used here)


def symmetric_transform(x, y, phi, offset: float = 0.5):
    """The per-row map itself for separate coordinate tensors x, y [R, N, 1] and angles phi [R, 1, 1] (transforms.py:51-72)."""
    from . import ops

    xy = torch.cat((x, y), -1).contiguous()
    ang = phi.detach().reshape(-1).to(device="cpu", dtype=xy.dtype)
    cs = torch.stack((torch.cos(ang), torch.sin(ang)), 1).to(xy.device)
    codes = (8 + (ang > 2 * math.pi).to(torch.int32)).to(xy.device)
    return ops.augment_xy(xy, codes, cs, offset=offset)


def min_max_normalize(x):
    lo, hi = x.min(), x.max()
    return (x - lo) / (hi - lo)


_AUGMENTATIONS = {"dihedral8": dihedral_8_augmentation_wrapper, "symmetric": symmetric_augmentation}


def get_augment_function(augment_fn):
    if callable(augment_fn):
        return augment_fn
    if augment_fn not in _AUGMENTATIONS:
        raise ValueError(f"Unknown augment_fn: {augment_fn}. Available options: 'symmetric', 'dihedral8' or a custom callable")
    return _AUGMENTATIONS[augment_fn]


class StateAugmentation:
    """td [B] -> td [num_augment * B] in "(a b)" order with the coordinate features augmented (rl4co/data/transforms.py:106-153).
    augment_fn: 'symmetric' (the reference's default), 'dihedral8' (needs num_augment == 8) or a callable
    `fn(batchified_feature, num_augment)`.  `phi` (not in the reference's signature; 'symmetric' only): angles to replay.
    first_aug_identity=False reproduces the reference as it is written: it saves and restores entry [B, 0] of the augmented
    feature -- `td_aug[feat][list(td.size()), 0]` indexes row B (the first instance's second copy), node 0, not the first copy
    -- pinned by tests/golden/eval_tsp20_augment_symmetric_noident.npz."""

    def __init__(self, num_augment: int = 8, augment_fn="symmetric", first_aug_identity: bool = True,
                 normalize: bool = False, feats=None, phi=None):
        self.augmentation = get_augment_function(augment_fn)
        if self.augmentation is dihedral_8_augmentation_wrapper and num_augment != 8:
            raise AssertionError("When using the `dihedral8` augmentation function, then num_augment must be 8")
        self.feats = ["locs"] if feats is None else list(feats)
        self.num_augment, self.normalize, self.first_aug_identity, self.phi = num_augment, normalize, first_aug_identity, phi

    def __call__(self, td):
        out = batchify(td, self.num_augment)
        if not isinstance(out, TensorDict):
            out = TensorDict(dict(out.items()), batch_size=out.batch_size)
        keep_at = list(td.batch_size)                   # the reference's index: [B]
        for name in self.feats:
            before = out[name]
            kept = None if self.first_aug_identity else before[keep_at, 0].clone()
            if self.augmentation is symmetric_augmentation and self.phi is not None:
                after = symmetric_augmentation(before, self.num_augment, phi=self.phi)
            else:
                after = self.augmentation(before, self.num_augment)
            if self.normalize:
                after = min_max_normalize(after)
            if kept is not None:
                after[keep_at, 0] = kept
            out.set(name, after)
        return out


def sample_n_random_actions(td, n: int, generator=None):
    """n random feasible first actions per instance, rows in (n b) order (rl4co/utils/ops.py:239-256).  Like the
    reference, the scores are drawn on the host with the global torch RNG (or `generator`)."""
    action_mask = td["action_mask"].cpu()
    replace = bool(torch.sum(action_mask[:, 1:], 1).min() < n)
    ps = torch.rand(action_mask.shape, generator=generator)
    ps[~action_mask] = -torch.inf
    ps = torch.softmax(ps, dim=1)
    selected = torch.multinomial(ps, n, replacement=replace, generator=generator)
    return selected.permute(1, 0).reshape(-1).to(td["action_mask"].device)
