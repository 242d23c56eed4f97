"""Multi-GPU helpers: one process per GPU, independent instance shards, one collective for training.

The rollout path has no exchange step: rank r simply owns instances [r*B/W, (r+1)*B/W).
The only collective of the system is the policy-gradient all-reduce, which the reference gets from
Lightning's DDPStrategy(find_unused_parameters=True) (rl4co/utils/trainer.py:72-89): here it is ONE flat
fp32 buffer (<= 5.2 MB for POMO, latency-bound over xGMI) reduced with a single RCCL call; parameters that
received no gradient contribute zeros, exactly like DDP with unused parameters.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    if torch.cuda.is_available() and torch.cuda.device_count() > local_rank:
        # one process per GPU: tensors created on "cuda" and the library's launches go to this rank's device
        torch.cuda.set_device(local_rank)
    return rank, world, local_rank


def shard_range(total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of `total` instances for `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_tensordict(td, rank: int | None = None, world: int | None = None):
    """This rank's slice of a batch (all starts of an instance stay on one GPU: shard BEFORE multistart)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(td.batch_size[0], rank, world)
    return td[lo:hi]


def gather_rows(x: torch.Tensor, total: int | None = None):
    """All ranks' rows concatenated in rank order (rewards / actions for reporting).  Ragged shards are padded."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    world = dist.get_world_size()
    n = torch.tensor([x.shape[0]], device=x.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros(mx, *x.shape[1:], device=x.device, dtype=x.dtype)
    pad[: x.shape[0]] = x
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], 0)


class FlatGradBuffer:
    """All gradients of a module in ONE pre-allocated flat fp32 buffer; every `p.grad` is a view into it.

    The reference gets its collective from DDP(gradient_as_bucket_view=True) (rl4co/utils/trainer.py:72-89); here the
    whole policy (<= 5.2 MB for POMO) is one bucket, so a training step has exactly one all-reduce and no copy kernels
    around it: autograd accumulates into the views in place, `allreduce` reduces the buffer, `clip_` scales it, the
    optimizer reads the views.  Parameters that receive no gradient keep their zeros (DDP with
    find_unused_parameters=True behaves the same).  Use `zero_()` instead of `optimizer.zero_grad()` (which would
    drop the views with set_to_none=True)."""

    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradBuffer: the module has no trainable parameters")
        dev = self.params[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise TypeError("FlatGradBuffer: fp32 parameters on one device required")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def attached(self) -> bool:
        """True while every p.grad still is its view (e.g. optimizer.zero_grad(set_to_none=True) detaches them)."""
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                return False
            off += p.numel()
        return True

    @torch.no_grad()
    def zero_(self):
        self.flat.zero_()

    @torch.no_grad()
    def allreduce(self, average: bool = True) -> int:
        """One collective over the whole buffer; returns the number of elements reduced."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if average:
                self.flat.div_(dist.get_world_size())
        return self.flat.numel()

    @torch.no_grad()
    def clip_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ on the buffer (global 2-norm AFTER the reduce, as DDP + Lightning's
        gradient_clip_val do, rl4co/utils/trainer.py:55): returns the norm before clipping."""
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total


@torch.no_grad()
def broadcast_module_state(module: torch.nn.Module, src: int = 0) -> int:
    """Every rank takes rank `src`'s parameters and buffers -- what DistributedDataParallel does at construction
    (and, for buffers, before each forward; rl4co/utils/trainer.py:72-89).  One flat fp32 buffer for the floating-point
    tensors (parameters, BatchNorm running statistics), one int64 buffer for integer buffers (`num_batches_tracked`).
    Returns the number of elements sent; a no-op outside a process group."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    tensors = [p.data for p in module.parameters()] + [b for b in module.buffers()]
    sent = 0
    for is_float in (True, False):
        group = [t for t in tensors if t.is_floating_point() == is_float]
        if not group:
            continue
        flat = torch.cat([t.reshape(-1).to(torch.float32 if is_float else torch.int64) for t in group])
        dist.broadcast(flat, src=src)
        off = 0
        for t in group:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        sent += flat.numel()
    return sent


@torch.no_grad()
def allreduce_buffers(module: torch.nn.Module) -> int:
    """Mean of the floating-point buffers over the ranks (BatchNorm running statistics under `policy.train()`: each rank
    updates them from its own shard; DDP would instead overwrite them with rank 0's -- the mean keeps every shard's
    statistics and leaves all ranks identical, which is what the bit-identical-replica guarantee needs)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    bufs = [b for b in module.buffers() if b.is_floating_point()]
    if not bufs:
        return 0
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    dist.all_reduce(flat)
    flat /= dist.get_world_size()
    off = 0
    for b in bufs:
        b.copy_(flat[off:off + b.numel()].view_as(b))
        off += b.numel()
    return flat.numel()


@torch.no_grad()
def allreduce_gradients(module: torch.nn.Module, average: bool = True):
    """Sum (mean) gradients across ranks with one flat all-reduce; returns the number of elements reduced.
    With a `FlatGradBuffer` attached to the module (`module._flat_grads`) this is the buffer's zero-copy all-reduce;
    otherwise the gradients are packed into a temporary flat buffer and copied back."""
    buf = getattr(module, "_flat_grads", None)
    if isinstance(buf, FlatGradBuffer) and buf.attached():
        return buf.allreduce(average)
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return 0
    dev = params[0].device
    flat = torch.zeros(sum(p.numel() for p in params), device=dev, dtype=torch.float32)
    views, off = [], 0
    for p in params:
        views.append(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
    have = [(v, p.grad) for v, p in zip(views, params) if p.grad is not None]
    if have:
        torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= dist.get_world_size()
    for v, p in zip(views, params):
        if p.grad is None:
            p.grad = v.clone()
        else:
            p.grad.copy_(v)
    return flat.numel()


@torch.no_grad()
def allreduce_scalars(values: dict, average: bool = True) -> dict:
    """Metric reduction (the reference logs with sync_dist=True, rl4co/models/rl/common/base.py:233-240)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(values)
    keys = sorted(values)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(values[k]) for k in keys], dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    if average:
        t /= dist.get_world_size()
    return {k: float(v) for k, v in zip(keys, t)}


DEFAULT_METRICS = {"train": ("loss", "reward"), "val": ("reward",), "test": ("reward",)}


@torch.no_grad()
def sync_metrics(metric_dict: dict, phase: str, metrics: dict | None = None, dataloader_name: str = "") -> dict:
    """What `RL4COLitModule.log_metrics` hands to Lightning with `sync_dist=True` (rl4co/models/rl/common/base.py:111-119,
    216-241): of a step's outputs only the phase's metric names (default train: loss, reward; val / test: reward), each
    reduced to its mean, keyed `"{phase}/{name}"` (+ `"/{dataloader}"`), then averaged over the ranks -- ONE all-reduce of
    a few scalars per call.  Returns {key: float}."""
    wanted = (metrics or DEFAULT_METRICS).get(phase, ())
    suffix = f"/{dataloader_name}" if dataloader_name else ""
    local = {}
    for k, v in metric_dict.items():
        if k in wanted:
            local[f"{phase}/{k}{suffix}"] = float(v.float().mean()) if isinstance(v, torch.Tensor) else float(v)
    return allreduce_scalars(local, average=True)
