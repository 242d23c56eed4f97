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
