#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the AttentionModel construction rollout on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload tsp100|cvrp100|tsp20|cvrp500|pomo100|sdvrp100|pctsp100|op100|cvrptw100] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full rollout of one batch: encoder + decoder cache + the whole decode loop (state update,
mask, masked single-query attention, selection) + tour-length reward, inputs already resident in HBM
(env.reset / instance generation excluded).  Default workload = BASELINE.json configs[1]:
TSP num_loc=100, batch=1024, greedy.  Metric: env-steps/s = ranks * batch * num_loc * K / wall time.
Multi-GPU: independent instance batches per rank (weak scaling), no data-path collective.

Besides the contract keys the JSON line carries
  roofline      the decode-loop kernel (the dominant hot-loop kernel): algorithmic bytes per launch
                (154,852 B per TSP-100 decode step) / its mean duration measured with HIP events
  cpu_baseline  the CPU oracle (a port of the reference's algorithm, oracle/) timed on this host's cores on a
                bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (env, num_loc, default batch, decode type)
    "tsp100": ("tsp", 100, 1024, "greedy"),        # BASELINE.json configs[1]  (headline)
    "tsp20": ("tsp", 20, 128, "greedy"),           # configs[0]
    "cvrp100": ("cvrp", 100, 1024, "sampling"),    # configs[2]
    "cvrp500": ("cvrp", 500, 512, "greedy"),       # configs[4]
    "sdvrp100": ("sdvrp", 100, 1024, "greedy"),    # sibling env: split deliveries, dynamic embedding
    "pctsp100": ("pctsp", 100, 1024, "greedy"),    # sibling env: prize collecting
    "cvrptw100": ("cvrptw", 100, 1024, "greedy"),  # sibling env: CVRP with time windows
    "op100": ("op", 100, 1024, "greedy"),          # sibling env: orienteering (distance-dependent mask)
    # configs[3], per-GPU share: POMO policy (6 layers, instance norm, no graph context), num_starts = num_loc
    "pomo100": ("tsp", 100, 1024, "multistart_sampling"),
}
POMO_KW = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
F32_PEAK_TFLOPS = 157.3  # dense fp32 (vector = f32 MFMA) peak, same guide


def algorithmic_flops_per_query_step(M, E=128):
    """6*M*E (QK^T, AV, logits) + 2*(2E)*E (context projection) + 2*E*E (output projection)."""
    return 6 * M * E + 4 * E * E + 2 * E * E


def algorithmic_bytes_per_decode_step(env, M, E=128, S=1):
    """12*M*E (K,V,L) + 4*E*g (context rows) + 2*M (mask r/w) + c; multistart (per
    instance-step, S queries sharing K/V/L): 12*M*E + S*(8*E + 2*M + 28)."""
    if S > 1:
        return 12 * M * E + S * (8 * E + 2 * M + 28)
    if env == "tsp":
        return 12 * M * E + 4 * E * 2 + 2 * M + 28
    if env == "sdvrp":      # mask r/w + remaining demand r/w instead of the visited bytes
        return 12 * M * E + 4 * E * 1 + 2 * M + 32 + 8 * M
    return 12 * M * E + 4 * E * 1 + 2 * M + 32 + 2 * M


def measured_traffic(workload, batch):
    """HBM bytes per decode-loop launch from the committed PMC passes (profiles/r01_traffic.json): rocprofv3
    counters cannot be collected from inside the timed run, so `traffic` is the separately profiled value for
    exactly this workload, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            t = json.load(f)
        return t.get(f"{workload}_b{batch}", {}).get("traffic_bytes")
    except (OSError, ValueError):
        return None


def build_policy(env_name, device, pomo=False):
    import eam_rl4co_amd as ea
    from _util import golden_weights

    pol = ea.AttentionModelPolicy(env_name=env_name, **(POMO_KW if pomo else {})).eval()
    torch.manual_seed(0)
    sd = pol.state_dict()
    for k, v in golden_weights(("pomo_" if pomo else "am_") + env_name).items():   # closed-form weights (untrained)
        sd[k].copy_(torch.from_numpy(v))
    return pol.to(device)


def cpu_baseline(env_name, num_loc, decode_type, num_starts=0, pomo=False, seconds_budget=20.0):
    """Time the CPU oracle on a bounded sample (batch chosen so the run takes ~10-30 s)."""
    from _util import golden_weights
    from oracle import oracle as orc
    import eam_rl4co_amd as ea

    threads = orc.set_threads(min(orc.usable_cpus(), 64))   # the box's CPU share, not its core count
    sd = golden_weights(("pomo_" if pomo else "am_") + env_name)
    env = ea.get_env(env_name, generator_params=dict(num_loc=num_loc), seed=1234)
    S = max(num_starts, 1)

    def run(batch):
        torch.manual_seed(1234)
        td = env.reset(batch_size=[batch])
        locs = td["locs"].numpy()
        demand = td["demand"].numpy() if env_name in ("cvrp", "sdvrp") else None
        if env_name == "pctsp":
            demand = {k: td[k].numpy() for k in ("expected_prize", "real_prize", "penalty", "prize_required")}
        if env_name == "op":
            demand = {k: td[k].numpy() for k in ("prize", "max_length")}
        if env_name == "cvrptw":
            demand = {k: td[k].numpy() for k in ("demand", "time_windows", "durations")}
        noise = None
        if "sampling" in decode_type:
            M = locs.shape[1]
            noise = torch.empty(batch * S, (3 if env_name == "sdvrp" else 2) * M + 1, M).exponential_(1).numpy()
        t0 = time.perf_counter()
        out = orc.policy_rollout(sd, env_name, locs, demand, decode_type=decode_type, num_starts=num_starts, noise=noise)
        return time.perf_counter() - t0, out["steps"]

    b = 16 if S == 1 else 2
    t, _ = run(b)                                  # calibration (also warms the library)
    target = max(b, int(b * min(seconds_budget / max(t, 1e-3), 64)))
    target = min(target, 1024)
    t, steps = run(target)
    return {"value": round(target * S * num_loc / t, 1), "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{env_name.upper()}-{num_loc} {decode_type} rollout, batch={target}"
                      + (f" x {S} starts" if S > 1 else "") + f", {steps} decode steps, "
                      f"{t:.2f} s on {threads} OpenMP threads (oracle/eamrl_oracle.c)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="tsp100", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a HIP graph")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: the workload's batch is the GLOBAL batch, split over the ranks (default: weak, "
                         "the batch is per GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("EAMRL_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm; gloo only for rehearsals
    if os.environ.get("EAMRL_BENCH_SINGLE_DEVICE") == "1":      # rehearse N ranks on a 1-GPU box (with gloo)
        local_rank = 0
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import eam_rl4co_amd as ea
    from eam_rl4co_amd import ops

    for kv in filter(None, os.environ.get("EAMRL_DEBUG_KEYS", "").split(",")):   # kernel A/B experiments only
        from eam_rl4co_amd import _lib
        _lib.load().eamrl_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))

    env_name, num_loc, batch, decode_type = WORKLOADS[args.workload]
    batch = args.batch or batch
    if args.strong:
        if batch % world:
            raise SystemExit(f"--strong: global batch {batch} is not divisible by {world} ranks")
        batch //= world
    env = ea.get_env(env_name, generator_params=dict(num_loc=num_loc), seed=1234 + rank)
    torch.manual_seed(1234 + rank)
    td0 = env.reset(batch_size=[batch]).to(device)         # synthetic uniform-[0,1]^2 instances, resident in HBM
    pomo = args.workload.startswith("pomo")
    policy = build_policy(env_name, device, pomo=pomo)
    M = td0["locs"].shape[1]
    S = num_loc if "multistart" in decode_type else 1
    dkw = dict(num_starts=S) if S > 1 else {}

    # time the decode-loop kernel with HIP events on the launch stream (torch's current stream)
    kernel_ms, decode_steps = [], []
    orig_rollout = ops.rollout

    def timed_rollout(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_rollout(*a, **k)
        e1.record()
        kernel_ms.append((e0, e1))
        return r

    graphed = None if args.no_graph else ea.GraphedRollout(policy, env, td0, decode_type=decode_type, **dkw)

    def one_step():
        if graphed is not None:
            return graphed(td0)                      # copy inputs into the captured buffers + one graph replay
        return policy(td0.clone(), env, phase="test", decode_type=decode_type, **dkw)

    # duration of the decode-loop launch, and of all nn.Linear launches (the encoder / cache GEMMs): HIP events around
    # them on the launch stream, in eager passes before the timed region
    gemm_ev, gemm_flops = [], []
    orig_linear, orig_mmr = ops.linear, ops.matmul_right

    def timed(fn, flops_of):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            gemm_ev.append((e0, e1))
            gemm_flops.append(flops_of(a, r))
            return r
        return wrapper

    ops.rollout = timed_rollout
    ops.linear = timed(orig_linear, lambda a, r: 2.0 * a[0].numel() * r.shape[-1])
    ops.matmul_right = timed(orig_mmr, lambda a, r: 2.0 * a[0].numel() * r.shape[-1])
    n_pass = 5
    for i in range(n_pass):
        if i == 2:
            gemm_ev.clear(); gemm_flops.clear()
        policy(td0.clone(), env, phase="test", decode_type=decode_type, **dkw)
    torch.cuda.synchronize()
    ops.rollout, ops.linear, ops.matmul_right = orig_rollout, orig_linear, orig_mmr
    kernel_ms = kernel_ms[2:]
    gemm_ms_per_rollout = sum(a.elapsed_time(b) for a, b in gemm_ev) / (n_pass - 2)
    gemm_tflops = sum(gemm_flops) / (n_pass - 2) / (gemm_ms_per_rollout * 1e-3) / 1e12 if gemm_ev else 0.0
    for _ in range(args.warmup):
        out = one_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
        decode_steps.append(out["actions"].shape[1])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:      # the slowest rank defines the step time
        tt = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * batch * S * num_loc * args.steps / elapsed
        kern = float(np.mean([a.elapsed_time(b) for a, b in kernel_ms]))
        T = float(np.mean(decode_steps))
        alg_bytes = algorithmic_bytes_per_decode_step(env_name, M, S=S) * batch * T
        achieved = alg_bytes / (kern * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(args.workload, batch),
                    "kernel": "decode loop (eamrl_am_rollout)", "kernel_ms": round(kern, 4),
                    "algorithmic_bytes_per_launch": int(alg_bytes)}
        if S > 1:
            # S queries share one K/V/L tile: 63 flop/B, beyond the fp32 ridge -> priced against the fp32 peak
            # (SURVEY.md 8d); the HBM view is kept next to it
            flops = algorithmic_flops_per_query_step(M) * batch * S * T
            tf = flops / (kern * 1e-3) / 1e12
            roofline = {"bound": "mfma", "achieved": round(tf, 2), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tf / F32_PEAK_TFLOPS, 4), "traffic": None,
                        "kernel": "decode loop (eamrl_am_rollout), fp32 VALU: one workgroup per row, starts of an "
                                  "instance looped on register-resident K/V/L",
                        "kernel_ms": round(kern, 4), "algorithmic_flops_per_launch": int(flops),
                        "hbm_view": {"achieved_GBs": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4)}}
        line = {
            "metric": "env-steps/sec (batch x num_loc / s), AttentionModel construction rollout",
            "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{env_name.upper()} num_loc={num_loc} batch={batch}/GPU "
                                   + (f"x {S} starts POMO" if pomo else "AM") + f" {decode_type} rollout "
                                   f"(encoder + cache + decode loop + reward)",
                       "decode_steps": T, "reward_mean": round(float(out["reward"].mean()), 4),
                       "launch": "eager" if args.no_graph else "hipGraph replay"},
            "roofline": roofline,
            # the launches that take the largest share of the step time: the one-shot encoder / cache Linears on fp32 MFMA
            "roofline_gemm": {"bound": "mfma", "achieved": round(gemm_tflops, 2), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(gemm_tflops / F32_PEAK_TFLOPS, 4), "kernel": "encoder + cache nn.Linear launches",
                              "ms_per_step": round(gemm_ms_per_rollout, 4), "launches_per_step": len(gemm_ev) // (n_pass - 2)},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(env_name, num_loc, decode_type, num_starts=S if S > 1 else 0, pomo=pomo)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
