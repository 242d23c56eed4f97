"""Minimal TensorDict used at the RL4COEnvBase boundary when the real `tensordict` package is absent.

If `tensordict` is importable it is used instead (`from eam_rl4co_amd.tensordict_lite import TensorDict`
then returns the real class), so code written against RL4CO keeps working either way.  Only the container
surface the rollout path needs is implemented: keyed tensors sharing leading batch dims.
"""
from __future__ import annotations

import torch

try:  # pragma: no cover - not installed in the build image
    from tensordict import TensorDict as _RealTensorDict
except Exception:  # noqa: BLE001
    _RealTensorDict = None


class _LiteTensorDict:
    def __init__(self, source=None, batch_size=None, device=None, **_unused):
        if isinstance(batch_size, int):
            batch_size = [batch_size]
        self._bs = torch.Size(batch_size if batch_size is not None else [])
        self._d = {}
        self._device = torch.device(device) if device is not None else None
        for k, v in dict(source or {}).items():
            if not isinstance(v, (torch.Tensor, _LiteTensorDict)):
                v = torch.as_tensor(v)
            if self._device is not None and isinstance(v, torch.Tensor) and v.device != self._device:
                v = v.to(self._device)
            self._check(k, v)
            self._d[k] = v

    def _check(self, k, v):
        nb = len(self._bs)
        if tuple(v.shape[:nb]) != tuple(self._bs):
            raise RuntimeError(f"batch dimension mismatch for '{k}': got {tuple(v.shape)} with batch_size {tuple(self._bs)}")

    # ---- meta ----
    @property
    def batch_size(self):
        return self._bs

    @property
    def shape(self):
        return self._bs

    @property
    def device(self):
        if self._device is not None:
            return self._device
        for v in self._d.values():
            return v.device
        return None

    def dim(self):
        return len(self._bs)

    def size(self, i=None):
        return self._bs if i is None else self._bs[i]

    def __len__(self):
        return self._bs[0] if len(self._bs) else 0

    def __repr__(self):
        body = ", ".join(f"{k}: {tuple(v.shape)} {str(v.dtype).replace('torch.', '')}" for k, v in self._d.items())
        return f"TensorDict({{{body}}}, batch_size={list(self._bs)}, device={self.device})"

    # ---- dict surface ----
    def keys(self, *a, **k):
        return self._d.keys()

    def items(self):
        return self._d.items()

    def values(self):
        return self._d.values()

    def __contains__(self, k):
        return k in self._d

    def __iter__(self):
        raise TypeError("iterate over .keys() / .items()")

    def is_empty(self):
        return len(self._d) == 0

    def get(self, k, default=None):
        return self._d.get(k, default)

    def set(self, k, v, inplace=False):
        if not isinstance(v, (torch.Tensor, _LiteTensorDict)):
            v = torch.as_tensor(v)
        self._d[k] = v
        return self

    def pop(self, k, default=None):
        return self._d.pop(k, default)

    def update(self, other):
        for k, v in (other.items() if hasattr(other, "items") else other):
            self.set(k, v)
        return self

    def __setitem__(self, k, v):
        self.set(k, v)

    def __getitem__(self, idx):
        if isinstance(idx, str):
            return self._d[idx]
        probe = torch.empty(self._bs, device="meta")[idx]
        return _LiteTensorDict({k: v[idx] for k, v in self._d.items()}, batch_size=probe.shape)

    def select(self, *keys):
        return _LiteTensorDict({k: self._d[k] for k in keys}, batch_size=self._bs)

    def exclude(self, *keys):
        return _LiteTensorDict({k: v for k, v in self._d.items() if k not in keys}, batch_size=self._bs)

    def to_dict(self):
        return dict(self._d)

    # ---- batch-dim ops (batchify / unbatchify, rl4co/utils/ops.py:13-56) ----
    def _map(self, fn, new_bs):
        return _LiteTensorDict({k: fn(v) for k, v in self._d.items()}, batch_size=new_bs)

    def clone(self, *a, **k):
        return self._map(lambda v: v.clone(), self._bs)

    def to(self, device, non_blocking=False):
        if device is None:
            return self
        out = self._map(lambda v: v.to(device, non_blocking=non_blocking), self._bs)
        return out

    def cpu(self):
        return self.to("cpu")

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def contiguous(self):
        return self._map(lambda v: v.contiguous(), self._bs)

    def expand(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        nb = len(self._bs)
        return self._map(lambda v: v.expand(*shape, *v.shape[nb:]), torch.Size(shape))

    def view(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        nb = len(self._bs)
        return self._map(lambda v: v.reshape(*shape, *v.shape[nb:]), torch.Size(shape))

    reshape = view

    def permute(self, *dims):
        dims = tuple(dims[0]) if len(dims) == 1 and not isinstance(dims[0], int) else tuple(dims)
        nb = len(self._bs)
        return self._map(lambda v: v.permute(*dims, *range(nb, v.dim())), torch.Size(self._bs[d] for d in dims))


TensorDict = _RealTensorDict if _RealTensorDict is not None else _LiteTensorDict
USING_REAL_TENSORDICT = _RealTensorDict is not None
