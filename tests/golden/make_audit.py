#!/usr/bin/env python3
"""Full-batch reference-vs-oracle audit at BASELINE.json's sizes.

Run only in the build container (the reference never travels):

    python tests/golden/make_audit.py [case ...]

The small goldens of make_golden.py pin the oracle on 4..16 instances.  This script RUNS THE REFERENCE
(unmodified, through `_refshim`) on the full batches the bench quotes and stores what a test needs to hold the
oracle and the HIP path against it:

  audit_tsp100_b1024_greedy      C2  AM,   TSP-100 x 1024, greedy
  audit_cvrp100_b1024_sampling   C3  AM,   CVRP-100 x 1024, sampling
  audit_pomo_tsp100_b16_s100     C4  POMO, TSP-100 x 16 instances x 100 starts, multistart sampling (per-instance shape)
  audit_cvrp500_b4_greedy        C5  AM,   CVRP-500 x 4, greedy (the reference has no fixture above 200 nodes)

Stored per case: the reference's tours (int16), rewards and summed log-likelihoods, and the NEAR-TIE TABLE -- every
(row, step) at which the reference's best and second-best selection scores are closer than NEAR (1e-3; the
selection score is the log-prob for greedy decoding and log-prob - log(noise) for sampling, i.e. the log of the ratio
torch.multinomial maximises).  An independent implementation whose floats differ in the last bits from torch's CPU
kernels can only part ways with the reference at such a step; the tests assert exactly that (first divergence of
every differing row is in the table with a gap < 1e-4) and state the match fraction.

Instances are not stored (0.8 MB of incompressible floats per case): they are the generator's output for the
pinned seed, which `tests/test_host_cpu.py` holds bit-identical between the reference's and the package's
generators; a CRC of the reference's tensors is stored and checked.  Sampling noise is not stored either (84 MB at C3):
it is the counter-based Exp(1) field of DESIGN.md section 2, `noise(seed, row, step, node)`, computed by the oracle's
integer Philox; `torch.multinomial` is replaced by `argmax(probs / noise)`, the identity make_golden.py's Recorder
asserts on every sampled step of every sampling golden.
"""
from __future__ import annotations

import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refshim  # noqa: E402

_refshim.install()

import torch  # noqa: E402

import goldweights  # noqa: E402
import rl4co.utils.decoding as ref_decoding  # noqa: E402
from rl4co.envs.routing.cvrp.env import CVRPEnv  # noqa: E402
from rl4co.envs.routing.tsp.env import TSPEnv  # noqa: E402
from rl4co.models.zoo.am.policy import AttentionModelPolicy  # noqa: E402

NEAR = 1e-3
POMO = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False)
CASES = {
    "audit_tsp100_b1024_greedy": dict(env="tsp", N=100, B=1024, decode="greedy", policy={}, data_seed=1234),
    "audit_cvrp100_b1024_sampling": dict(env="cvrp", N=100, B=1024, decode="sampling", policy={}, data_seed=1234,
                                         noise_seed=20261005),
    "audit_pomo_tsp100_b16_s100": dict(env="tsp", N=100, B=16, S=100, decode="multistart_sampling", policy=POMO,
                                       data_seed=99, noise_seed=4242),
    "audit_cvrp500_b4_greedy": dict(env="cvrp", N=500, B=4, decode="greedy", policy={}, data_seed=500),
}


def crc(a) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def make_policy(env_name, **kw):
    pol = AttentionModelPolicy(env_name=env_name, **kw).eval()
    sd = pol.state_dict()
    for k, v in goldweights.fill_state_dict(sd).items():
        sd[k].copy_(torch.from_numpy(v))
    return pol


class Audit:
    """Hooks the reference's process_logits (per-step log-probs) and torch.multinomial (selection with the given noise);
    keeps only the top-2 gap of the selection score per (row, step)."""

    def __init__(self, noise=None):
        self.noise, self.gaps, self.t = noise, [], 0

    def _gap(self, score):
        top = torch.topk(score.double(), 2, dim=-1).values
        return (top[:, 0] - top[:, 1]).float()            # inf when a single node is feasible

    def __enter__(self):
        self._pl, self._mn = ref_decoding.process_logits, torch.multinomial

        def pl(*a, **k):
            out = self._pl(*a, **k)
            if self.noise is None:                        # greedy: argmax of the log-probs
                self.gaps.append(self._gap(out))
            return out

        def mn(probs, num_samples, *a, **k):
            assert num_samples == 1
            q = torch.from_numpy(self.noise[:, self.t])
            ratio = probs / q                             # what torch.multinomial maximises (A6)
            self.gaps.append(self._gap(torch.log(ratio.double())))
            self.t += 1
            return torch.argmax(ratio, dim=-1, keepdim=True)

        ref_decoding.process_logits = pl
        torch.multinomial = mn
        return self

    def __exit__(self, *exc):
        ref_decoding.process_logits, torch.multinomial = self._pl, self._mn


def first_divergence(a, b):
    """Per row: index of the first differing step of two [R, T] arrays (T if none), after padding to a common T with 0."""
    T = max(a.shape[1], b.shape[1])
    pa = np.zeros((a.shape[0], T), np.int64)
    pb = np.zeros_like(pa)
    pa[:, :a.shape[1]], pb[:, :b.shape[1]] = a, b
    ne = pa != pb
    return np.where(ne.any(1), ne.argmax(1), T)


def run(name, env, N, B, decode, policy, data_seed, S=None, noise_seed=None):
    from oracle import oracle as orc

    Env = {"tsp": TSPEnv, "cvrp": CVRPEnv}[env]
    renv = Env(generator_params=dict(num_loc=N), seed=data_seed)
    torch.manual_seed(data_seed)
    td = renv.reset(batch_size=[B])
    pol = make_policy(env, **policy)
    M = td["locs"].shape[1]
    R = B * (S or 1)
    pre = 1 if S else 0
    t_noise = (M - pre) if env == "tsp" else 2 * M + 1
    noise = orc.exp1_noise(noise_seed, R, t_noise, M) if noise_seed is not None else None
    kw = dict(decode_type=decode)
    if S:
        kw["num_starts"] = S
    t0 = time.time()
    with torch.inference_mode(), Audit(noise) as au:
        out = pol(td.clone(), renv, phase="test", return_sum_log_likelihood=False, **kw)
    t_ref = time.time() - t0
    acts = out["actions"].numpy()
    gaps = torch.stack(au.gaps, 1).numpy()                # [R, T - pre]: gap of the step that chose actions[:, pre + t]
    rows, steps = np.nonzero(gaps < NEAR)
    fx = {
        "torch_version": np.array(torch.__version__), "env_name": np.array(env), "decode_type": np.array(decode),
        "num_loc": np.array(N, np.int64), "batch": np.array(B, np.int64), "num_starts": np.array(S or 0, np.int64),
        "data_seed": np.array(data_seed, np.int64), "noise_seed": np.array(-1 if noise_seed is None else noise_seed, np.int64),
        "pomo": np.array(bool(policy)), "locs_crc": np.array(crc(td["locs"].numpy()), np.int64),
        "actions": acts.astype(np.int16), "reward": out["reward"].numpy(),
        "log_likelihood": out["log_likelihood"].sum(1).numpy(),
        "near_rows": rows.astype(np.int32), "near_steps": (steps + pre).astype(np.int32),
        "near_gaps": gaps[rows, steps].astype(np.float32), "near_threshold": np.array(NEAR, np.float32),
    }
    demand = None
    if env == "cvrp":
        demand = td["demand"].numpy()
        fx["demand_crc"] = np.array(crc(demand), np.int64)
    # informational: where the oracle stands today (the tests recompute this live)
    t0 = time.time()
    cfg = ("pomo_" if policy else "am_") + env
    sys.path.insert(0, os.path.dirname(HERE))
    from _util import golden_weights

    o = orc.policy_rollout(golden_weights(cfg), env, td["locs"].numpy(), demand, decode_type=decode, num_starts=S or 0,
                           noise=noise, use_graph_context=not policy)
    t_orc = time.time() - t0
    fd = first_divergence(acts, o["actions"])
    T = max(acts.shape[1], o["actions"].shape[1])
    div = np.nonzero(fd < T)[0]
    table = {(int(r), int(s)): float(g) for r, s, g in zip(fx["near_rows"], fx["near_steps"], fx["near_gaps"])}
    dg = [table.get((int(r), int(fd[r])), float("inf")) for r in div]
    fx["oracle_div_rows"], fx["oracle_div_steps"] = div.astype(np.int32), fd[div].astype(np.int32)
    fx["oracle_div_gaps"] = np.array(dg, np.float32)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"{name}: T={acts.shape[1]} reference {t_ref:.1f}s oracle {t_orc:.1f}s ({torch.get_num_threads()} threads); "
          f"{len(rows)} near-tie entries; oracle == reference on {R - len(div)}/{R} rows, divergences at "
          f"{[(int(r), int(fd[r]), f'{g:.1e}') for r, g in zip(div, dg)]} -> {os.path.getsize(path) / 1024:.0f} KiB",
          flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for n in names:
        run(n, **CASES[n])
