#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the AttentionModel construction rollout on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload W] [--batch B] [--strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: the parent process (which never
touches the GPU) runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child and exits with its
code; rank 0 of the child prints the one JSON line with n_gpus == N.

Rollout workloads (tsp100 = BASELINE.json configs[1] is the default): one "step" = one full rollout of one batch --
encoder + decoder cache + the whole decode loop (state update, mask, masked single-query attention, selection) +
tour-length reward, inputs already resident in HBM (env.reset / instance generation excluded).
Training workload (pomo100_train = configs[3], per-GPU share 1024 instances x 100 starts): one step = sampled multistart
rollout -> teacher-forced re-evaluation + backward -> ONE flat RCCL all-reduce -> clip 1.0 -> Adam
(eam_rl4co_amd.train.PolicyGradientStep, the reference's POMO.shared_step + trainer).
Metric: env-steps/s = ranks * batch * starts * num_loc * K / wall time (max over ranks, barrier on both sides).
Multi-GPU: independent instance batches per rank, no data-path collective ("weak": the batch is per GPU; `--strong`: the
workload's batch is the GLOBAL batch, split over the ranks).  A weak run on N > 1 ranks also times the strong split of
the same global batch and reports it under "strong_scaling".

Besides the contract keys the JSON line carries
  roofline         the launch family with the largest share of the step (today the encoder / cache GEMMs on fp32 MFMA):
                   algorithmic flops per step / its time measured with HIP events on the launch stream, against the
                   dense fp32 MFMA peak
  roofline_decode  the decode-loop kernel against the bound it actually sits on: per-ROLLOUT bytes for
                   the register-resident kernel, per-step bytes for the streaming one, plus the step-API view
  cpu_baseline     the CPU oracle (a port of the reference's algorithm, oracle/) timed on this host's cores on a
                   bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOADS = {
    # name: (env, num_loc, default batch per GPU, decode type, kind)
    "tsp100": ("tsp", 100, 1024, "greedy", "rollout"),        # BASELINE.json configs[1]  (headline)
    "tsp20": ("tsp", 20, 128, "greedy", "rollout"),           # configs[0]
    "cvrp100": ("cvrp", 100, 1024, "sampling", "rollout"),    # configs[2]
    "cvrp500": ("cvrp", 500, 512, "greedy", "rollout"),       # configs[4]
    "sdvrp100": ("sdvrp", 100, 1024, "greedy", "rollout"),    # sibling env: split deliveries, dynamic embedding
    "pctsp100": ("pctsp", 100, 1024, "greedy", "rollout"),    # sibling env: prize collecting
    "cvrptw100": ("cvrptw", 100, 1024, "greedy", "rollout"),  # sibling env: CVRP with time windows
    "op100": ("op", 100, 1024, "greedy", "rollout"),          # sibling env: orienteering (distance-dependent mask)
    # configs[3], per-GPU share: POMO policy (6 layers, instance norm, no graph context), num_starts = num_loc
    "pomo100": ("tsp", 100, 1024, "multistart_sampling", "rollout"),
    "pomo100_train": ("tsp", 100, 1024, "multistart_sampling", "train"),   # configs[3]: the whole REINFORCE step
    "pomo_cvrp100": ("cvrp", 100, 1024, "multistart_sampling", "rollout"),  # the fork's other POMO setting (CVRP-100 x 100 starts)
    "pomo_cvrp100_train": ("cvrp", 100, 1024, "multistart_sampling", "train"),
    "pomo20_train": ("tsp", 20, 64, "multistart_sampling", "train"),       # small rehearsal of the same step
    # no GPU work at all: the launcher, the rendezvous and the collective half of the training step (flat all-reduce ->
    # clip -> Adam on synthetic per-rank gradients) -- what the CPU (gloo) tests drive
    "dist_selftest": ("tsp", 20, 4, "multistart_sampling", "selftest"),
}
POMO_KW = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
F32_PEAK_TFLOPS = 157.3  # dense fp32 (vector = f32 MFMA) peak, same guide


def algorithmic_flops_per_query_step(M, E=128):
    """6*M*E (QK^T, AV, logits) + 2*(2E)*E (context projection) + 2*E*E (output projection)."""
    return 6 * M * E + 4 * E * E + 2 * E * E


def executed_flops_per_query_step(M, E=128):
    """What the decode kernels execute per query-step: 6*M*E.  The 6*E*E of SURVEY 8(d) (context and output projections) are
    folded into the Pa / Pb / Lp cache slots and are executed -- and counted -- once per node in the encoder
    launch's cache term, so pricing the decode loop with them would count them twice."""
    return 6 * M * E


def measured_counters(workload, family):
    """MFMA-pipe busy fraction and achieved clock of a kernel family from the committed PMC pass (profiles/r0N*_mfma_counters.json,
    tools/collect_mfma_counters.sh + tools/summarize_counters.py): SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), and
    GRBM_GUI_ACTIVE / 8 XCDs / duration.  Counters cannot be collected inside the timed run; None when no pass is committed."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_mfma_counters.json")), reverse=True):
        try:
            with open(path) as f:
                e = json.load(f).get(workload, {}).get(family)
        except (OSError, ValueError):
            continue
        if e:
            c = e["counters_per_launch"]
            out = {"source": os.path.relpath(path, ROOT), "kernel_ms_profiled": e["mean_duration_ms"], "clock_ghz": e.get("clock_ghz"),
                   "valu_insts_per_launch": c.get("SQ_INSTS_VALU")}
            if c.get("SQ_BUSY_CU_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
                out["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"]), 4)
                out["mfma_executed_tflop"] = round(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512 / 1e12, 4)   # incl. padded rows
            return out
    return None


def algorithmic_bytes_per_decode_step(env, M, E=128, S=1):
    """SURVEY.md 8(d), step-at-a-time API: 12*M*E (K,V,L) + 4*E*g (context rows) + 2*M (mask r/w) + c; multistart (per
    instance-step, S queries sharing K/V/L): 12*M*E + S*(8*E + 2*M + 28)."""
    if S > 1:
        return 12 * M * E + S * (8 * E + 2 * M + 28)
    if env == "tsp":
        return 12 * M * E + 4 * E * 2 + 2 * M + 28
    if env == "sdvrp":      # mask r/w + remaining demand r/w instead of the visited bytes
        return 12 * M * E + 4 * E * 1 + 2 * M + 32 + 8 * M
    return 12 * M * E + 4 * E * 1 + 2 * M + 32 + 2 * M


def resident_bytes_per_rollout(env, M, T, E=128, S=1, sampling=False):
    """Bytes the register-resident decode kernel must move per INSTANCE and rollout: K, V, Lp once
    (12*M*E), the context rows Pa (, Pb) once (4*M*E*g), and per row the state (mask / visited bytes in, final state out),
    the outputs (action i64 + log-prob f32 per step) and, for sampling, the Exp(1) noise (4*M per step)."""
    g = 2 if env == "tsp" else 1
    per_row = 2 * M + 64 + T * 12 + (4 * M * T if sampling else 0)
    return 12 * M * E + 4 * M * E * g + max(S, 1) * per_row


def measured_traffic(workload, batch, suffix=""):
    """HBM bytes per decode-loop launch from the committed PMC passes (profiles/r0N_traffic.json): rocprofv3 counters
    cannot be collected from inside the timed run, so `traffic` is the separately profiled value for exactly this
    workload, or None."""
    for name in ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f).get(f"{workload}_b{batch}{suffix}", {}).get("traffic_bytes")
            if t is not None:
                return t
        except (OSError, ValueError):
            pass
    return None


def build_policy(env_name, device, pomo=False):
    import torch

    import eam_rl4co_amd as ea
    from _util import golden_weights

    pol = ea.AttentionModelPolicy(env_name=env_name, **(POMO_KW if pomo else {})).eval()
    sd = pol.state_dict()
    # closed-form weights (untrained, the goldens' streams) instead of SURVEY 8d's torch.manual_seed(0) default init: every rank
    # then holds the same policy whatever its seed, and TSP timing does not depend on the weights (CVRP: T is reported)
    for k, v in golden_weights(("pomo_" if pomo else "am_") + env_name).items():
        sd[k].copy_(torch.from_numpy(v))
    return pol.to(device)


def cpu_baseline(env_name, num_loc, decode_type, num_starts=0, pomo=False, seconds_budget=20.0):
    """The reference's CPU path on this host's cores, on a bounded sample (~10-30 s of CPU work).

    TSP / CVRP: oracle/torch_cpu_baseline.py, the reference's rollout restated as the same sequence of torch-CPU ops (validated
    against the reference itself in the build container: profiles/r03_cpu_baseline_validation.json, 1.00x / 0.98x its time at
    C2 / C3).  The C oracle (a port of the ALGORITHM with a defined arithmetic order, 2-3x slower than the reference's MKL /
    flash-attention kernels) is timed too and reported under "c_oracle"; it is the only baseline of the sibling envs."""
    import torch

    import eam_rl4co_amd as ea
    from _util import golden_weights
    from oracle import oracle as orc

    threads = orc.set_threads(min(orc.usable_cpus(), 64))   # the box's CPU share, not its core count
    torch.set_num_threads(threads)
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
        out = orc.policy_rollout(sd, env_name, locs, demand, decode_type=decode_type, num_starts=num_starts, noise=noise,
                                 use_graph_context=not pomo)
        return time.perf_counter() - t0, out["steps"]

    def run_torch(batch):
        from oracle import torch_cpu_baseline as tb

        torch.manual_seed(1234)
        td = env.reset(batch_size=[batch])
        keys = ("locs", "first_node", "current_node", "i", "action_mask", "done", "demand", "used_capacity", "vehicle_capacity",
                "visited")
        tdd = {k: td[k].clone() for k in keys if k in td.keys()}
        sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
        if not pomo:            # eval-mode batch norm: the (default) running statistics are part of the state_dict
            for k, v in ea.AttentionModelPolicy(env_name=env_name).state_dict().items():
                sdt.setdefault(k, v)
        with torch.inference_mode():
            t0 = time.perf_counter()
            out = tb.rollout(sdt, env_name, tdd, decode_type=decode_type, num_starts=num_starts, use_graph_context=not pomo)
            return time.perf_counter() - t0, out["steps"]

    b = 16 if S == 1 else 2
    t, _ = run(b)                                  # calibration (also warms the library)
    target = max(b, int(b * min(0.5 * seconds_budget / max(t, 1e-3), 64)))
    target = min(target, 1024)
    t, steps = run(target)
    c_oracle = {"value": round(target * S * num_loc / t, 1), "unit": "env-steps/s", "cores": threads,
                "sample": f"batch={target}" + (f" x {S} starts" if S > 1 else "") + f", {steps} decode steps, {t:.2f} s "
                          f"(oracle/eamrl_oracle.c, OpenMP)"}
    if env_name not in ("tsp", "cvrp"):
        return dict(c_oracle, kind="port", sample=f"{env_name.upper()}-{num_loc} {decode_type} rollout, " + c_oracle["sample"])
    bt = 1024 if S == 1 else 128
    run_torch(max(1, bt // 8))                     # warms torch's thread pool and kernels
    t, steps = run_torch(bt)
    return {"value": round(bt * S * num_loc / t, 1), "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{env_name.upper()}-{num_loc} {decode_type} rollout, batch={bt}" + (f" x {S} starts" if S > 1 else "")
                      + f", {steps} decode steps, {t:.2f} s on {threads} torch threads: the reference's op sequence on torch-CPU "
                        "(oracle/torch_cpu_baseline.py)",
            "validated": "profiles/r03_cpu_baseline_validation.json: 1.00x / 0.98x the reference's own time at TSP-100 x 1024 / "
                         "CVRP-100 x 1024 on the build container's 8 cores (0.89x at TSP-20 x 128, where the reference's "
                         "TensorDict bookkeeping dominates)",
            "c_oracle": c_oracle}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """Parent of a `--gpus N` run: start the N ranks as a child `torch.distributed.run` and mirror its exit code.
    Nothing here initialises the GPU (no torch.cuda call, no HIP library loaded), and nothing is exec'ed."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


class FamilyTimer:
    """HIP events (torch.cuda.Event on the launch stream = torch's current stream, which is the stream the C ABI gets)
    around the calls of one launch family, in eager passes before the timed region."""

    def __init__(self, torch):
        self.torch, self.ev, self.flops = torch, {}, {}

    def wrap(self, fam, fn, flops_of=None):
        def wrapper(*a, **k):
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            self.ev.setdefault(fam, []).append((e0, e1))
            if flops_of is not None:
                self.flops[fam] = self.flops.get(fam, 0.0) + (flops_of(a, r, k) if flops_of.__code__.co_argcount == 3 else flops_of(a, r))
            return r
        return wrapper

    def clear(self):
        self.ev.clear(); self.flops.clear()

    def ms(self, fam, passes):
        return sum(a.elapsed_time(b) for a, b in self.ev.get(fam, [])) / passes

    def launches(self, fam, passes):
        return len(self.ev.get(fam, [])) // passes


def params_checksum(policy, torch):
    """Order-independent bit checksum of all parameters (int64 sum of the fp32 bit patterns)."""
    tot = torch.zeros((), dtype=torch.int64, device=next(policy.parameters()).device)
    for p in policy.parameters():
        tot += p.detach().view(torch.int32).to(torch.int64).sum()
    return tot


def run_selftest(args, torch, dist, rank, world):
    """No GPU: N ranks over gloo run the collective half of the training step on synthetic per-rank gradients."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd.dist import FlatGradBuffer

    torch.manual_seed(0)
    pol = ea.AttentionModelPolicy(env_name="tsp", **POMO_KW)
    buf = FlatGradBuffer(pol)
    opt = torch.optim.Adam(pol.parameters(), lr=1e-4, weight_decay=1e-6)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for it in range(args.warmup + args.steps):
        buf.zero_()
        g = torch.Generator().manual_seed(1000 * rank + it)          # every rank its own gradients
        loss = sum((p * torch.randn(p.shape, generator=g)).sum() for p in pol.parameters())
        loss.backward()
        assert buf.attached()
        buf.allreduce(average=True)
        buf.clip_(1.0)
        opt.step()
    elapsed = time.perf_counter() - t0
    cs = params_checksum(pol, torch)
    lo, hi = cs.clone(), cs.clone()
    if world > 1:
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "dist_selftest (no GPU work)", "value": round(args.steps / elapsed, 2), "unit": "steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / max(args.steps + args.warmup, 1) * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "flat gradient all-reduce -> clip 1.0 -> Adam on synthetic gradients (POMO policy, "
                                                 f"{buf.flat.numel()} parameters)", "backend": dist.get_backend() if world > 1 else None},
                          "params_identical": bool(lo.item() == hi.item()), "params_checksum": int(cs.item())}), flush=True)


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

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env_name, num_loc, batch, decode_type, kind = WORKLOADS[args.workload]
    backend = os.environ.get("EAMRL_DIST_BACKEND", "gloo" if kind == "selftest" else "nccl")   # "nccl" is RCCL on ROCm
    if os.environ.get("EAMRL_BENCH_SINGLE_DEVICE") == "1":      # rehearse N ranks on a 1-GPU box (with gloo)
        local_rank = 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if kind == "selftest":
        if world > 1:
            dist.init_process_group(backend)
        run_selftest(args, torch, dist, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import eam_rl4co_amd as ea
    from eam_rl4co_amd import ops

    for kv in filter(None, os.environ.get("EAMRL_DEBUG_KEYS", "").split(",")):   # kernel A/B experiments only
        from eam_rl4co_amd import _lib
        _lib.load().eamrl_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))

    batch = args.batch or batch
    global_batch = batch if args.strong else batch * world
    if args.strong:
        if batch % world:
            raise SystemExit(f"--strong: global batch {batch} is not divisible by {world} ranks")
        batch //= world
    env = ea.get_env(env_name, generator_params=dict(num_loc=num_loc), seed=1234 + rank)
    torch.manual_seed(1234 + rank)
    td0 = env.reset(batch_size=[batch]).to(device)         # synthetic uniform-[0,1]^2 instances, resident in HBM
    # the timed steps rotate through NBATCH different synthetic batches (all resident before the clock starts), so that no
    # step meets the instances of the step before it in a cache
    NBATCH = 4
    tds = [td0] + [env.reset(batch_size=[batch]).to(device) for _ in range(NBATCH - 1)]
    pomo = args.workload.startswith("pomo")
    policy = build_policy(env_name, device, pomo=pomo)
    M = td0["locs"].shape[1]
    S = num_loc if "multistart" in decode_type else 1
    dkw = dict(num_starts=S) if S > 1 else {}
    train = kind == "train"
    stepper = None
    if train:
        from eam_rl4co_amd.train import PolicyGradientStep

        policy.train()
        stepper = PolicyGradientStep(policy, env, num_starts=S)       # Adam(1e-4, wd 1e-6), clip 1.0, shared baseline

    def make_step(batches, graph=True):
        it = {"i": 0}

        def nxt():
            it["i"] += 1
            return batches[it["i"] % len(batches)]

        if train:
            return lambda: stepper(nxt())
        g = None if (args.no_graph or not graph) else ea.GraphedRollout(policy, env, batches[0], decode_type=decode_type, **dkw)
        if g is not None:
            return lambda: g(nxt())                   # copy inputs into the captured buffers + one graph replay
        return lambda: policy(nxt().clone(), env, phase="test", decode_type=decode_type, **dkw)

    one_step = make_step(tds)

    # per launch family: HIP events around the calls on the launch stream, in eager passes before the timed region
    ft = FamilyTimer(torch)
    orig = {n: getattr(ops, n) for n in ("rollout", "linear", "matmul_right", "mha_encoder", "encoder_fused")}

    def fused_flops(a, r, k):   # per instance and layer: qkv + QK^T + PV + out_proj + FFN (algorithmic, unpadded)
        if a[0] is not None:
            Bq, Mq, Eq = a[0].shape
        else:       # init embedding computed inside the kernel: shapes from the feature tensor and its weight
            (Bq, Mq, _), Eq = k["init"]["feat"].shape, k["init"]["W"].shape[0]
        per = 2.0 * Mq * Eq * 3 * Eq + 4.0 * Mq * Mq * Eq + 2.0 * Mq * Eq * Eq + 4.0 * Mq * Eq * a[3]
        tot = per * Bq * len(a[1])
        if k.get("cache") is not None:      # + the decoder cache projections done from LDS: (nproj + 1) x [M, E] x [E, E]
            tot += 2.0 * Mq * Eq * Eq * (k["cache"][3] + 1) * Bq
        return tot

    ops.encoder_fused = ft.wrap("encoder_fused", orig["encoder_fused"], fused_flops)
    ops.rollout = ft.wrap("decode", orig["rollout"])
    ops.linear = ft.wrap("gemm", orig["linear"], lambda a, r: 2.0 * a[0].numel() * r.shape[-1])
    ops.matmul_right = ft.wrap("gemm", orig["matmul_right"], lambda a, r: 2.0 * a[0].numel() * r.shape[-1])
    ops.mha_encoder = ft.wrap("attention", orig["mha_encoder"],
                              lambda a, r: 4.0 * r.shape[0] * r.shape[1] * r.shape[1] * r.shape[2])
    n_pass, skip = (3, 1) if train else (5, 2)
    rollout_ms = []
    for i in range(n_pass):
        if i == skip:
            ft.clear()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            policy(td0.clone(), env, phase="train" if train else "test", decode_type=decode_type, **dkw)
        e1.record()
        rollout_ms.append((e0, e1))
    torch.cuda.synchronize()
    for n, f in orig.items():
        setattr(ops, n, f)
    passes = n_pass - skip
    rollout_only_ms = float(np.mean([a.elapsed_time(b) for a, b in rollout_ms[skip:]]))
    fam_ms = {f: ft.ms(f, passes) for f in ("gemm", "attention", "decode", "encoder_fused")}

    def timed(step_fn, steps, warmup):
        out = None
        for _ in range(warmup):
            out = step_fn()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nsteps = []
        for _ in range(steps):
            out = step_fn()
            nsteps.append(out["actions"].shape[1])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:      # the slowest rank defines the step time
            tt = torch.tensor([el], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, out, nsteps

    elapsed, out, decode_steps = timed(one_step, args.steps, args.warmup)

    strong = None
    if world > 1 and not args.strong and not train and batch % world == 0:
        # the same GLOBAL batch as the 1-GPU run, split over the ranks (north_star: strong scaling)
        el_s, _, _ = timed(make_step([t_[: batch // world] for t_ in tds]), args.steps, max(1, args.warmup))
        strong = {"global_batch": batch, "batch_per_gpu": batch // world, "ms_per_step": round(el_s / args.steps * 1e3, 4),
                  "value": round(batch * S * num_loc * args.steps / el_s, 1), "unit": "env-steps/s"}

    identical = None
    if train:
        cs = params_checksum(policy, torch)
        lo, hi = cs.clone(), cs.clone()
        if world > 1:
            if backend != "nccl":
                lo, hi = lo.cpu(), hi.cpu()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        identical = bool(lo.item() == hi.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * batch * S * num_loc * args.steps / elapsed
        kern = fam_ms["decode"]
        T = float(np.mean(decode_steps)) - (1 if S > 1 else 0)
        # ---- decode loop: the bound it sits on ----------------------------------------------------------------
        step_api_bytes = algorithmic_bytes_per_decode_step(env_name, M, S=S) * batch * T
        step_api = {"achieved_GBs": round(step_api_bytes / (kern * 1e-3) / 1e9, 1),
                    "frac_of_hbm_peak": round(step_api_bytes / (kern * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "step-at-a-time API bytes (SURVEY 8d: K, V, L re-read every step); > 1 means the kernel does not "
                            "re-read them"}
        traffic = measured_traffic(args.workload.replace("_train", ""), batch)
        if M <= 128:
            rb = resident_bytes_per_rollout(env_name, M, T, S=S, sampling="sampling" in decode_type) * batch
            flops = executed_flops_per_query_step(M) * batch * S * T
            tf = flops / (kern * 1e-3) / 1e12
            roofline_decode = {
                "kernel": ("k_rollout_ms_mfma (decode loop of all starts of an instance on fp32 MFMA, K/V/Lp as register fragments)"
                           if S > 1 and env_name in ("tsp", "cvrp") and M <= 112 else
                           "k_rollout_resident (decode loop, K/V/Lp read once per rollout into VGPRs)"),
                "bound": "hbm", "achieved": round(rb / (kern * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(rb / (kern * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(rb), "kernel_ms": round(kern, 4),
                "issue_bound": {"bound": "mfma" if S > 1 and env_name in ("tsp", "cvrp") and M <= 112 else "fp32 valu issue",
                                "achieved": round(tf, 2), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(tf / F32_PEAK_TFLOPS, 4),
                                "note": "the kernel is instruction-issue / latency bound, not HBM bound (DESIGN.md 4): executed "
                                        "flops 6*M*E per query-step against the fp32 peak (the 6*E*E context / output "
                                        "projections of SURVEY 8d are folded into the cache and counted in the encoder launch)",
                                "counters": measured_counters(args.workload.replace("_train", ""),
                                                              "k_rollout_ms_mfma" if S > 1 and env_name in ("tsp", "cvrp") and M <= 112
                                                              else "k_rollout_resident")},
                "step_api_view": step_api}
        else:
            ach = (traffic or step_api_bytes) / (kern * 1e-3) / 1e9
            roofline_decode = {
                "kernel": "k_rollout_stream (decode loop, K/V/Lp re-read every step; masked rows skipped)",
                "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel_ms": round(kern, 4),
                "basis": "measured HBM traffic (PMC)" if traffic else "algorithmic bytes (no PMC pass for this workload)",
                "algorithmic_bytes_per_launch": int(step_api_bytes), "step_api_view": step_api}
        # ---- the family with the largest share of the step ------------------------------------------------------
        gemm_tf = ft.flops.get("gemm", 0.0) / passes / (fam_ms["gemm"] * 1e-3) / 1e12 if fam_ms["gemm"] else 0.0
        att_tf = ft.flops.get("attention", 0.0) / passes / (fam_ms["attention"] * 1e-3) / 1e12 if fam_ms["attention"] else 0.0
        roofline_gemm = {"kernel": "k_linear_mfma (encoder + cache nn.Linear launches, v_mfma_f32_32x32x2_f32)",
                         "bound": "mfma", "achieved": round(gemm_tf, 2), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tf / F32_PEAK_TFLOPS, 4), "traffic": None,
                         "ms_per_step": round(fam_ms["gemm"], 4), "launches_per_step": ft.launches("gemm", passes),
                         "algorithmic_flops_per_step": int(ft.flops.get("gemm", 0.0) / passes)}
        roofline_att = {"kernel": "k_mha_encoder (encoder self-attention)", "bound": "mfma", "achieved": round(att_tf, 2),
                        "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(att_tf / F32_PEAK_TFLOPS, 4),
                        "ms_per_step": round(fam_ms["attention"], 4), "launches_per_step": ft.launches("attention", passes)}
        enc_tf = (ft.flops.get("encoder_fused", 0.0) / passes / (fam_ms["encoder_fused"] * 1e-3) / 1e12
                  if fam_ms["encoder_fused"] else 0.0)
        roofline_enc = {"kernel": "k_encoder_fused (all encoder layers of an instance in one workgroup, v_mfma_f32_16x16x4_f32)",
                        "bound": "mfma", "achieved": round(enc_tf, 2), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(enc_tf / F32_PEAK_TFLOPS, 4),
                        "traffic": measured_traffic(args.workload.replace("_train", ""), batch, "_encoder_fused"),
                        "kernel_ms": round(fam_ms["encoder_fused"], 4), "launches_per_step": ft.launches("encoder_fused", passes),
                        "algorithmic_flops_per_launch": int(ft.flops.get("encoder_fused", 0.0) / passes),
                        "counters": measured_counters(args.workload.replace("_train", ""), "k_encoder_fused")}
        dominant = max(fam_ms, key=lambda f: fam_ms[f])
        roofline = dict(roofline_enc if dominant == "encoder_fused" else roofline_gemm if dominant == "gemm" else
                        roofline_decode["issue_bound"] | {"kernel": roofline_decode["kernel"], "kernel_ms": round(kern, 4)}
                        if (dominant == "decode" and M <= 128) else roofline_decode if dominant == "decode" else roofline_att)
        roofline["share_of_step"] = {f: round(v / rollout_only_ms, 3) for f, v in fam_ms.items()}
        roofline["dominant_family"] = dominant
        cfg = {"workload": f"{env_name.upper()} num_loc={num_loc} batch={batch}/GPU "
                           + (f"x {S} starts POMO" if pomo else "AM") + f" {decode_type} "
                           + ("REINFORCE training step (rollout + re-evaluation + backward + flat all-reduce + clip + Adam)"
                              if train else "rollout (encoder + cache + decode loop + reward)"),
               "decode_steps": T, "reward_mean": round(float(out["reward"].mean()), 4),
               "global_batch": global_batch,
               "launch": "eager" if (args.no_graph or train) else "hipGraph replay"}
        if train:
            cfg["rollout_ms"] = round(rollout_only_ms, 3)
            cfg["gradient_side_ms"] = round(ms_per_step - rollout_only_ms, 3)
        line = {
            "metric": "env-steps/sec (batch x num_loc / s), AttentionModel construction rollout",
            "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": f"synthetic (uniform instances, {NBATCH} different batches in rotation; closed-form untrained weights)",
            "config": cfg, "roofline": roofline, "roofline_decode": roofline_decode, "roofline_gemm": roofline_gemm,
            "roofline_attention": roofline_att, "roofline_encoder_fused": roofline_enc,
        }
        if strong is not None:
            line["strong_scaling"] = strong
        # inputs of DESIGN.md 5's strong-scaling projection: this rank's share and how its step splits into the one-shot
        # encoder and the sequential decode loop (the first real multi-GPU run can be compared with the per-share table)
        line["per_gpu_share"] = {"batch_per_gpu": batch, "encoder_ms": round(fam_ms["encoder_fused"] + fam_ms["gemm"] + fam_ms["attention"], 4),
                                 "decode_ms": round(fam_ms["decode"], 4), "rollout_ms_eager": round(rollout_only_ms, 4)}
        if identical is not None:
            line["params_identical"] = identical
        if not args.no_cpu_baseline and world == 1:      # a reported baseline of the 1-GPU run only (the other ranks would wait)
            line["cpu_baseline"] = cpu_baseline(env_name, num_loc, decode_type, num_starts=S if S > 1 else 0, pomo=pomo)
            if train:
                line["cpu_baseline"]["sample"] += " -- rollout part only (the oracle has no backward)"
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
