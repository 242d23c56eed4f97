#!/usr/bin/env python3
"""Micro-benchmarks of single kernels at the TSP-100 B=1024 shapes (run on the GPU box).

    python tools/kernel_bench.py gemm | mha | decode | ea | train | all      [--iters 20]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eam_rl4co_amd import ops  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def bench_gemm(iters):
    rows = 102400
    dev = "cuda"
    shapes = [("qkv", 128, 384, False, False), ("out_proj+res+bn", 128, 128, True, True), ("ffn1+relu", 128, 512, False, False),
              ("ffn2+res+bn", 512, 128, True, True), ("kvl", 128, 384, False, False), ("pa", 128, 128, False, False)]
    for name, k, n, res, bn in shapes:
        x = torch.randn(rows, k, device=dev)
        W = torch.randn(n, k, device=dev) / k ** 0.5
        b = torch.randn(n, device=dev)
        r = torch.randn(rows, n, device=dev) if res else None
        bnp = (torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev), torch.randn(n, device=dev),
               torch.rand(n, device=dev) + 0.5, 1e-5) if bn else None
        out = torch.empty(rows, n, device=dev)
        us = timeit(lambda: ops.linear(x, W, b, relu=("relu" in name), residual=r, out=out, bn=bnp), iters)
        fl = 2.0 * rows * k * n
        by = 4.0 * (rows * k + rows * n * (2 if res else 1))
        print(f"gemm {name:18s} K={k:4d} N={n:4d}: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  {by / us / 1e3:6.0f} GB/s")


def bench_mha(iters):
    qkv = torch.randn(1024, 100, 384, device="cuda")
    us = timeit(lambda: ops.mha_encoder(qkv, 8), iters)
    print(f"mha_encoder B=1024 N=100: {us:8.1f} us")


def bench_decode(iters, t_max=None):
    """Resident rollout kernel alone at TSP-100 B=1024 with random cache contents (timing only)."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import _lib
    from eam_rl4co_amd.policy import state_from_td

    dev = "cuda"
    B, M, E = 1024, 100, 128
    env = ea.get_env("tsp", generator_params=dict(num_loc=M))
    td0 = env.reset(batch_size=[B]).to(dev)
    buf = torch.randn(B, M, 6 * E, device=dev) * 0.3
    emb = torch.randn(B, M, E, device=dev)
    cache = ops.DecodeCache("tsp", buf, torch.randn(E, device=dev), torch.randn(B, E, device=dev), emb, 8)

    states = [state_from_td("tsp", td0.clone(), 0) for _ in range(iters + 2)]
    torch.cuda.synchronize()
    times = []
    for st in states:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        ops.rollout(st, cache, "greedy", t_max=t_max or M)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3)
    print(f"decode resident t_max={t_max or M}: {sorted(times[2:])[len(times[2:]) // 2]:8.1f} us (median, events around ops.rollout)")


def bench_ea(iters):
    """eamrl_ea_tsp_run at the POMO training shape: 1024 instances x 100 starts x 100 nodes, 3 generations."""
    import eam_rl4co_amd as ea

    B, S, N, G = 1024, 100, 100, 3
    env = ea.get_env("tsp", generator_params=dict(num_loc=N))
    td = env.reset(batch_size=[B]).to("cuda")
    init = torch.stack([torch.stack([torch.cat([torch.tensor([s]), torch.tensor([x for x in torch.randperm(N).tolist() if x != s])])
                                     for s in range(S)]) for _ in range(4)]).repeat(B // 4, 1, 1).cuda()
    for rates in ((0.1, 0.6, 0.2), (0.5, 0.9, 1.0)):
        runner = ea.EA(env, dict(num_generations=G, mutation_rate=rates[0], crossover_rate=rates[1], selection_rate=rates[2]))
        d = ea.EADraws.sample(G, B, S, N, rates[2], "cuda")
        us = timeit(lambda: runner.run(init, td, draws=d), iters)
        print(f"ea_tsp_run B={B} S={S} N={N} G={G} rates={rates}: {us:8.1f} us  ({B * S * G / us:.1f} M individuals-generations/s)")


def bench_train(iters):
    """One EAM training step of the fork (zoo/earl/model.py:129-247) end to end: sampled multistart rollout (native),
    evolutionary improvement (native), teacher-forced re-evaluation with autograd, backward, Adam step."""
    import time

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import train

    for env_name, N, B, S in (("tsp", 50, 64, 50), ("tsp", 100, 64, 100), ("cvrp", 50, 64, 50)):
        env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=3)
        pol = ea.AttentionModelPolicy(env_name=env_name, num_encoder_layers=6, normalization="instance",
                                      use_graph_context=False).to("cuda")
        opt = torch.optim.Adam(pol.parameters(), lr=1e-4)
        runner = ea.EA(env, dict(num_generations=3, mutation_rate=0.1, crossover_rate=0.6, selection_rate=0.2))
        gen = torch.Generator(device="cuda").manual_seed(5)
        td = env.reset(batch_size=[B]).to("cuda")
        def step():
            t0 = time.perf_counter()
            res = train.eam_loss(pol, env, td, runner, num_starts=S, generator=gen,
                                 return_entropy=os.environ.get("EAM_ENTROPY", "1") == "1")   # (the reference's step asks for it)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            opt.zero_grad()
            res["loss"].backward()
            opt.step()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            return (t1 - t0) * 1e3, (t2 - t1) * 1e3

        for _ in range(4):          # the first steps pay one-time costs (library tuning of the new shapes, Adam state)
            step()
        # median of the steps: the evolved tours change the step count T from step to step, and a first-seen shape pays a
        # one-time allocation / library set-up (tens of ms) that a mean over a few steps would smear over them
        ts = sorted((step() for _ in range(max(9, iters // 2))), key=lambda p: p[0] + p[1])
        fwd, bwd = ts[len(ts) // 2]
        tot = fwd + bwd
        print(f"EAM training step {env_name}{N} B={B} S={S} (POMO policy): {tot:8.1f} ms  "
              f"forward (rollout + EA + re-evaluation) {fwd:.1f} ms  backward + Adam {bwd:.1f} ms"
              f"  = {B * S * N / tot * 1e3 / 1e6:.1f} M sampled env-steps/s incl. the update (median of {len(ts)} steps)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--bm128", type=int, default=0)
    ap.add_argument("--generic-epilogue", type=int, default=0, help="1: run-time configured GEMM epilogue (A/B against the templates)")
    a = ap.parse_args()
    from eam_rl4co_amd import _lib
    _lib.load().eamrl_debug_set(4, a.bm128)
    _lib.load().eamrl_debug_set(10, a.generic_epilogue)
    if a.what in ("gemm", "all"):
        bench_gemm(a.iters)
    if a.what in ("mha", "all"):
        bench_mha(a.iters)
    if a.what in ("ea", "all"):
        bench_ea(a.iters)
    for kv in filter(None, os.environ.get("EAMRL_DEBUG_KEYS", "").split(",")):   # kernel A/B experiments only
        from eam_rl4co_amd import _lib
        _lib.load().eamrl_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))
    if a.what in ("train",):
        bench_train(a.iters)
    if a.what in ("decode",):
        for tm in (None, 1, 11, 51):
            bench_decode(a.iters, tm)


if __name__ == "__main__":
    main()
