"""CPU, world_size 2 over gloo: instance sharding and the single flat gradient all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import dist as ed

    r, w, _ = ed.init_distributed("gloo")
    assert (r, w) == (rank, world)
    # --- sharding: every instance on exactly one rank, order preserved --------------------------------
    env = ea.get_env("cvrp", generator_params=dict(num_loc=10), seed=3)
    torch.manual_seed(3)
    td = env.reset(batch_size=[7])            # same seed on both ranks -> same global batch
    mine = ed.shard_tensordict(td)
    lo, hi = ed.shard_range(7, rank, world)
    assert mine.batch_size[0] == hi - lo and torch.equal(mine["locs"], td["locs"][lo:hi])
    back = ed.gather_rows(mine["demand"])
    assert torch.equal(back, td["demand"])
    # --- gradient all-reduce: mean over ranks, unused parameters contribute zeros -------------------------
    pol = ea.AttentionModelPolicy(env_name="tsp", num_encoder_layers=1)
    for i, p in enumerate(pol.parameters()):
        if i % 3 != 2:                          # every third parameter has no gradient on this rank
            p.grad = torch.full_like(p, float(rank + 1))
    n = ed.allreduce_gradients(pol)
    assert n == sum(p.numel() for p in pol.parameters())
    for i, p in enumerate(pol.parameters()):
        expect = 1.5 if i % 3 != 2 else 0.0     # mean of (1, 2); zeros where nobody had a gradient
        assert torch.allclose(p.grad, torch.full_like(p, expect))
    m = ed.allreduce_scalars({"reward": -10.0 * (rank + 1), "loss": float(rank)})
    assert abs(m["reward"] + 15.0) < 1e-12 and abs(m["loss"] - 0.5) < 1e-12
    dist.barrier()
    dist.destroy_process_group()
    out.put(rank)


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert sorted(out.get(timeout=5) for _ in range(2)) == [0, 1]


def test_shard_range_covers_everything():
    from eam_rl4co_amd.dist import shard_range

    for total in (0, 1, 7, 1024):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _flat_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import dist as ed

    ed.init_distributed("gloo")
    torch.manual_seed(0)                                   # same initial weights on both ranks
    pol = ea.AttentionModelPolicy(env_name="cvrp", num_encoder_layers=2)
    buf = ed.FlatGradBuffer(pol)
    pol._flat_grads = buf
    opt = torch.optim.Adam(pol.parameters(), lr=1e-3)
    ptrs = [p.grad.data_ptr() for p in pol.parameters()]
    for it in range(3):
        buf.zero_()
        g = torch.Generator().manual_seed(100 * rank + it)
        loss = sum((p * torch.randn(p.shape, generator=g)).sum() for i, p in enumerate(pol.parameters()) if i % 4 != 3)
        loss.backward()                                    # autograd accumulates INTO the views
        assert buf.attached() and [p.grad.data_ptr() for p in pol.parameters()] == ptrs
        n = ed.allreduce_gradients(pol)                    # dispatches to the buffer: one collective, no copies
        assert n == buf.flat.numel()
        norm = buf.clip_(1.0)
        assert float(torch.linalg.vector_norm(buf.flat)) <= 1.0 + 1e-5 < float(norm)
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in pol.parameters()])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "ranks hold different parameters after the training steps"
    for i, p in enumerate(pol.parameters()):               # every fourth parameter never had a gradient: zeros, unchanged
        if i % 4 == 3:
            assert float(p.grad.abs().max()) == 0.0
    dist.barrier()
    dist.destroy_process_group()
    out.put(rank)


def test_flat_grad_buffer_two_ranks_end_with_identical_parameters():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert sorted(out.get(timeout=5) for _ in range(2)) == [0, 1]


def _init_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import dist as ed
    from eam_rl4co_amd.train import PolicyGradientStep

    ed.init_distributed("gloo")
    torch.manual_seed(1000 + rank)                         # DIFFERENT initial weights per rank
    pol = ea.AttentionModelPolicy(env_name="tsp", num_encoder_layers=1)        # batch norm: running statistics are buffers
    with torch.no_grad():
        for b in pol.buffers():
            if b.is_floating_point():
                b.add_(float(rank))
    before = torch.cat([p.detach().reshape(-1) for p in pol.parameters()]).clone()
    step = PolicyGradientStep(pol, env=None, num_starts=0)  # construction broadcasts rank 0's parameters and buffers

    def gathered(t):
        both = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(both, t)
        return both

    flat = torch.cat([p.detach().reshape(-1) for p in pol.parameters()])
    both = gathered(flat)
    assert torch.equal(both[0], both[1]), "ranks start from different parameters"
    assert torch.equal(flat, before) == (rank == 0)        # rank 0 kept its own, rank 1 took rank 0's
    bufs = torch.cat([b.detach().reshape(-1).float() for b in pol.buffers()])
    bb = gathered(bufs)
    assert torch.equal(bb[0], bb[1])
    # the collective half of a step (the rollout half needs the GPU): per-rank gradients -> all-reduce -> clip -> Adam
    step.grads.zero_()
    g = torch.Generator().manual_seed(rank)
    sum((p * torch.randn(p.shape, generator=g)).sum() for p in pol.parameters()).backward()
    step.grads.allreduce(average=True)
    step.grads.clip_(1.0)
    step.optimizer.step()
    with torch.no_grad():                                   # each rank's shard moved the running statistics differently
        for b in pol.buffers():
            if b.is_floating_point():
                b.add_(0.25 * (rank + 1))
    ed.allreduce_buffers(pol)
    both = gathered(torch.cat([p.detach().reshape(-1) for p in pol.parameters()]))
    assert torch.equal(both[0], both[1]), "ranks hold different parameters after one step"
    bb = gathered(torch.cat([b.detach().reshape(-1).float() for b in pol.buffers()]))
    assert torch.equal(bb[0], bb[1])
    # metric reduction with the reference's keys (log_metrics(..., sync_dist=True), rl/common/base.py:216-241)
    out_dict = {"loss": torch.tensor(float(rank)), "reward": torch.full((5,), -10.0 * (rank + 1)), "actions": torch.zeros(5, 3),
                "log_likelihood": torch.ones(5)}
    m = ed.sync_metrics(out_dict, "train")
    assert sorted(m) == ["train/loss", "train/reward"] and abs(m["train/loss"] - 0.5) < 1e-12 and abs(m["train/reward"] + 15) < 1e-12
    v = ed.sync_metrics(out_dict, "val", dataloader_name="tsp100")
    assert list(v) == ["val/reward/tsp100"] and abs(v["val/reward/tsp100"] + 15) < 1e-12
    dist.barrier()
    dist.destroy_process_group()
    out.put(rank)


def test_policy_gradient_step_broadcasts_rank0_state_and_syncs_metrics():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_init_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert sorted(out.get(timeout=5) for _ in range(2)) == [0, 1]


def test_bench_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` (no torchrun, no WORLD_SIZE): the parent spawns two ranks before touching any GPU;
    rank 0 prints ONE JSON line with n_gpus == 2.  The selftest workload runs the collective half of the training step
    (flat all-reduce -> clip -> Adam) over gloo and reports whether both ranks ended with identical parameters."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "dist_selftest",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["params_identical"] is True
    single = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "dist_selftest", "--steps", "2",
                             "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    one = json.loads([l for l in single.stdout.splitlines() if l.startswith("{")][0])
    assert one["n_gpus"] == 1 and one["params_checksum"] != line["params_checksum"]    # two ranks averaged different gradients
