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


def dihedral_8_augmentation(xy: torch.Tensor) -> torch.Tensor:
    """The 8 symmetries of the unit square applied to [B, N, 2] coordinates -> [8B, N, 2], identity first
    (POMO; rl4co/data/transforms.py:16-40)."""
    x, y = xy[..., 0:1], xy[..., 1:2]
    variants = [(x, y), (1 - x, y), (x, 1 - y), (1 - x, 1 - y), (y, x), (1 - y, x), (y, 1 - x), (1 - y, 1 - x)]
    return torch.cat([torch.cat(v, dim=-1) for v in variants], dim=0)


def dihedral_8_augmentation_wrapper(xy: torch.Tensor, reduce: bool = True, *args, **kw) -> torch.Tensor:
    """On an already batchified [8B, N, 2] tensor: the 8 symmetries of its first B rows (transforms.py:43-50)."""
    xy = xy[: xy.shape[0] // 8, ...] if reduce else xy
    return dihedral_8_augmentation(xy)


def symmetric_transform(x, y, phi, offset: float = 0.5):
    """Rotation by phi about (offset, offset), then a reflection (x <-> y) where phi > 2 pi (transforms.py:51-72).
    cos / sin are evaluated on the host in fp32 (phi is one angle per row): the device result is then bit-identical to
    the reference's CPU evaluation for the same angles."""
    x, y = x - offset, y - offset
    c, s_ = torch.cos(phi.cpu()).to(x.device), torch.sin(phi.cpu()).to(x.device)
    x_prime = c * x - s_ * y
    y_prime = s_ * x + c * y
    mask = phi > 2 * math.pi
    xy = torch.cat((x_prime, y_prime), dim=-1)
    xy = torch.where(mask, xy.flip(-1), xy)
    return xy + offset


def symmetric_augmentation(xy: torch.Tensor, num_augment: int = 8, first_augment: bool = False, phi: torch.Tensor = None):
    """Random rotation / reflection per row of a batchified [A*B, N, 2] tensor (transforms.py:75-90).  The angles are an
    input, like the sampling noise: phi [A*B] in [0, 4 pi) (drawn with torch.rand on xy's device when None; the
    reference draws them on ITS device, so replaying a reference run means passing its angles)."""
    if phi is None:
        phi = torch.rand(xy.shape[0], device=xy.device) * 4 * math.pi
    else:
        phi = phi.to(device=xy.device, dtype=xy.dtype).clone()
    if not first_augment:       # the first copy of every instance stays as it is
        phi[: xy.shape[0] // num_augment] = 0.0
    x, y = xy[..., [0]], xy[..., [1]]
    return symmetric_transform(x, y, phi[:, None, None])


def min_max_normalize(x):
    return (x - x.min()) / (x.max() - x.min())


def get_augment_function(augment_fn):
    if callable(augment_fn):
        return augment_fn
    if augment_fn == "dihedral8":
        return dihedral_8_augmentation_wrapper
    if augment_fn == "symmetric":
        return symmetric_augmentation
    raise ValueError(f"Unknown augment_fn: {augment_fn}. Available options: 'symmetric', 'dihedral8' or a custom callable")


class StateAugmentation:
    """Instance augmentation of the coordinate features (rl4co/data/transforms.py:106-153): td [B] -> td [num_augment * B]
    in (a b) order.  augment_fn: 'symmetric' (the reference's default: random rotations / reflections), 'dihedral8'
    (POMO's 8 symmetries; needs num_augment == 8) or a callable.  `phi` (symmetric only): the angles to use instead of
    fresh random ones."""

    def __init__(self, num_augment: int = 8, augment_fn="symmetric", first_aug_identity: bool = True,
                 normalize: bool = False, feats=None, phi=None):
        self.augmentation = get_augment_function(augment_fn)
        assert not (self.augmentation == dihedral_8_augmentation_wrapper and num_augment != 8), \
            "When using the `dihedral8` augmentation function, then num_augment must be 8"
        self.feats = ["locs"] if feats is None else list(feats)
        self.num_augment, self.normalize, self.first_aug_identity, self.phi = num_augment, normalize, first_aug_identity, phi

    def __call__(self, td):
        out = batchify(td, self.num_augment)
        if not isinstance(out, TensorDict):
            out = TensorDict(dict(out.items()), batch_size=out.batch_size)
        B = td.batch_size[0]
        for f in self.feats:
            src = out[f]
            if not self.first_aug_identity:
                init_aug_feat = src[list(range(B)), 0].clone()
            if self.augmentation is symmetric_augmentation and self.phi is not None:
                aug = symmetric_augmentation(src, self.num_augment, phi=self.phi)
            else:
                aug = self.augmentation(src, self.num_augment)
            if self.normalize:
                aug = min_max_normalize(aug)
            if not self.first_aug_identity:
                aug[list(range(B)), 0] = init_aug_feat
            out.set(f, aug)
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
