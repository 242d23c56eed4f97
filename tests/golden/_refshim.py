"""Container-only import shim used by make_golden.py (NEVER imported by tests or the product).

The reference (/root/reference, rl4co fork) needs `tensordict` and `torchrl`, which are
not installed in this image and cannot be installed (no network).  Those two packages
carry no arithmetic of the rollout path: `TensorDict` is a keyed tensor container and
`torchrl.envs.EnvBase` only provides reset plumbing.  This module
registers minimal dict-backed stand-ins for exactly that container/plumbing surface so
that the reference's own env / decoder / decoding code runs UNMODIFIED and produces the
golden vectors.  It also pre-registers empty package modules for two reference
`__init__`s that would pull Lightning (absent) in, so their submodules import directly.

Everything numeric in the goldens therefore comes from the reference's source files
executed as they are; nothing from the reference is copied into this repository.
"""
from __future__ import annotations

import sys
import types

import torch


class TensorDict:
    """Dict of tensors sharing leading batch dims (stand-in for tensordict.TensorDict)."""

    def __init__(self, source=None, batch_size=None, device=None, **_unused):
        source = {} if source is None else source
        if isinstance(batch_size, int):
            batch_size = [batch_size]
        self._bs = torch.Size(batch_size if batch_size is not None else [])
        self._d = {}
        for k, v in dict(source).items():
            self._d[k] = torch.as_tensor(v) if not isinstance(v, (torch.Tensor, TensorDict)) else v
        self._device = device

    # --- meta -----------------------------------------------------------------
    @property
    def batch_size(self):
        return self._bs

    @property
    def shape(self):
        return self._bs

    @property
    def device(self):
        if self._device is not None:
            return torch.device(self._device)
        for v in self._d.values():
            return v.device
        return None

    def dim(self):
        return len(self._bs)

    def size(self, i=None):
        return self._bs if i is None else self._bs[i]

    def __len__(self):
        return self._bs[0] if len(self._bs) else 0

    # --- dict surface ---------------------------------------------------------
    def keys(self, *a, **k):
        return self._d.keys()

    def items(self):
        return self._d.items()

    def values(self):
        return self._d.values()

    def __contains__(self, k):
        return k in self._d

    def is_empty(self):
        return len(self._d) == 0

    def get(self, k, default=None):
        return self._d.get(k, default)

    def set(self, k, v):
        self._d[k] = v
        return self

    def update(self, other):
        for k, v in (other.items() if hasattr(other, "items") else other):
            self._d[k] = v
        return self

    def __setitem__(self, k, v):
        self._d[k] = v

    def __getitem__(self, idx):
        if isinstance(idx, str):
            return self._d[idx]
        out = {k: v[idx] for k, v in self._d.items()}
        probe = torch.empty(self._bs, device="meta")[idx]
        return TensorDict(out, batch_size=probe.shape)

    def exclude(self, *keys):
        return TensorDict({k: v for k, v in self._d.items() if k not in keys}, batch_size=self._bs)

    # --- batch-dim ops used by rl4co.utils.ops.batchify / unbatchify ------------
    def _map(self, fn, new_bs):
        return TensorDict({k: fn(v) for k, v in self._d.items()}, batch_size=new_bs)

    def clone(self, *a, **k):
        return self._map(lambda v: v.clone(), self._bs)

    def to(self, device):
        if device is None:
            return self
        return self._map(lambda v: v.to(device), self._bs)

    def cpu(self):
        return self.to("cpu")

    def contiguous(self):
        return self._map(lambda v: v.contiguous(), self._bs)

    def expand(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else shape
        nb = len(self._bs)
        lead = len(shape) - nb
        return self._map(lambda v: v.expand(*shape, *v.shape[nb:]) if lead >= 0 else v, torch.Size(shape))

    def view(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else shape
        nb = len(self._bs)
        return self._map(lambda v: v.reshape(*shape, *v.shape[nb:]), torch.Size(shape))

    def permute(self, *dims):
        dims = tuple(dims[0]) if len(dims) == 1 and not isinstance(dims[0], int) else dims
        nb = len(self._bs)
        new_bs = torch.Size([self._bs[d] for d in dims])
        return self._map(lambda v: v.permute(*dims, *range(nb, v.dim())), new_bs)

    def gather(self, dim, index):
        """tensordict's gather along batch dim `dim`: `index` has the batch rank, trailing feature dims are broadcast."""
        nb = len(self._bs)

        def g(v):
            idx = index.view(*index.shape, *([1] * (v.dim() - nb))).expand(*index.shape, *v.shape[nb:])
            return v.gather(dim, idx)

        return self._map(g, index.shape)

    def squeeze(self, dim):
        nb = len(self._bs)
        d = dim if dim >= 0 else dim + nb
        new_bs = torch.Size([s for i, s in enumerate(self._bs) if not (i == d and s == 1)])
        return self._map(lambda v: v.squeeze(d) if v.shape[d] == 1 else v, new_bs)


class EnvBase:
    """Stand-in for torchrl.envs.EnvBase: only the reset plumbing RL4COEnvBase relies on."""

    def __init__(self, device="cpu", batch_size=None, run_type_checks=False, allow_done_after_reset=False, **_):
        self.device = torch.device(device) if device is not None else None
        self.batch_size = torch.Size(batch_size if batch_size is not None else [])

    def set_seed(self, seed=None, static_seed=False):
        self._set_seed(seed)
        return seed

    def to(self, device):
        self.device = torch.device(device)
        return self

    def reset(self, td=None, **kwargs):
        out = self._reset(td, **kwargs)
        bs = kwargs.get("batch_size", out.batch_size)
        # TorchRL's reset adds done/terminated of shape [*B, 1] when _reset does not set them
        for key in ("done", "terminated"):
            if key not in out:
                out.set(key, torch.zeros(*bs, 1, dtype=torch.bool, device=out.device))
        return out


class _Spec:
    def __init__(self, *a, **k):
        pass


def install():
    """Register the stand-ins in sys.modules (idempotent) and put the reference on sys.path."""
    if "tensordict" in sys.modules and getattr(sys.modules["tensordict"], "_eamrl_shim", False):
        return
    td_mod = types.ModuleType("tensordict")
    td_mod.TensorDict = TensorDict
    td_mod.__version__ = "0.6.0"
    td_mod._eamrl_shim = True
    td_sub = types.ModuleType("tensordict.tensordict")
    td_sub.TensorDict = TensorDict
    td_mod.tensordict = td_sub
    sys.modules["tensordict"] = td_mod
    sys.modules["tensordict.tensordict"] = td_sub

    rl_mod = types.ModuleType("torchrl")
    rl_envs = types.ModuleType("torchrl.envs")
    rl_envs.EnvBase = EnvBase
    rl_data = types.ModuleType("torchrl.data")
    for n in ("Bounded", "Unbounded", "Composite", "BoundedTensorSpec", "UnboundedContinuousTensorSpec",
              "UnboundedDiscreteTensorSpec", "CompositeSpec"):
        setattr(rl_data, n, _Spec)
    rl_mod.envs, rl_mod.data = rl_envs, rl_data
    sys.modules["torchrl"] = rl_mod
    sys.modules["torchrl.envs"] = rl_envs
    sys.modules["torchrl.data"] = rl_data

    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")

    # skip the two package __init__s that import Lightning (absent); their submodules are plain files
    import importlib.util
    for pkg in ("rl4co.models.common", "rl4co.models.zoo.am", "rl4co.models.zoo"):
        spec = importlib.util.find_spec("rl4co")
        base = spec.submodule_search_locations[0]
        m = types.ModuleType(pkg)
        m.__path__ = [base + "/" + "/".join(pkg.split(".")[1:])]
        sys.modules[pkg] = m


# ---------------------------------------------------------------------------------------------------------
# numba stand-in for the fork's evolution operators (rl4co/models/zoo/earl/evolution.py)
# ---------------------------------------------------------------------------------------------------------
class _NbType:
    """nb.int64 / nb.float32 / nb.boolean ...: usable as a numpy dtype (`.dtype`), as a cast (`nb.int64(50)`)
    and inside signatures (`nb.int64[:, :]`, `nb.float32[:](...)`)."""

    def __init__(self, np_type):
        import numpy as np

        self._t = np_type
        self.dtype = np.dtype(np_type)

    def __getitem__(self, _):
        return self

    def __call__(self, *a, **k):
        if len(a) == 1 and not k and not isinstance(a[0], _NbType):
            return self._t(a[0])
        return self            # a signature such as nb.float32[:](nb.float32[:], nb.int64)


def install_numba():
    """`numba` is absent (no network).  Its decorators carry no arithmetic: with `njit` as the identity and
    `prange` as `range`, the reference's evolution operators run as the plain Python/numpy they are written in
    (sequentially, so the order of np.random draws is the program order)."""
    if "numba" in sys.modules:
        return
    import numpy as np

    nb = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not isinstance(a[0], _NbType) and not k:
            return a[0]
        return lambda f: f

    nb.njit = njit
    nb.jit = njit
    nb.prange = range
    nb.set_num_threads = lambda n: None
    for name in ("int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float32", "float64"):
        setattr(nb, name, _NbType(getattr(np, name)))
    nb.boolean = _NbType(np.bool_)
    nb.void = _NbType(np.float64)
    nb_types = types.ModuleType("numba.types")
    nb_types.Tuple = lambda members: (lambda *a, **k: None)
    nb.types = nb_types
    core = types.ModuleType("numba.core")
    config = types.ModuleType("numba.core.config")
    config.NUMBA_NUM_THREADS = 1
    core.config = config
    nb.core = core
    sys.modules["numba"] = nb
    sys.modules["numba.types"] = nb_types
    sys.modules["numba.core"] = core
    sys.modules["numba.core.config"] = config
