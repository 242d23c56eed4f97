#!/usr/bin/env python3
"""Build container only: times oracle/torch_cpu_baseline.py (the restated op sequence bench.py reports as `cpu_baseline`) against
the REFERENCE ITSELF (imported unmodified through tests/golden/_refshim.py) on the same instances, weights and thread count,
and writes profiles/r03_cpu_baseline_validation.json.  BASELINE.md section 4's bar: within +-10 % at C1 / C2 (and C3).

    PYTHONPATH=/root/reference python tools/validate_cpu_baseline.py
"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import _refshim  # noqa: E402

_refshim.install()
import torch  # noqa: E402

import goldweights  # noqa: E402
from oracle import torch_cpu_baseline as tb  # noqa: E402
from rl4co.envs.routing.cvrp.env import CVRPEnv  # noqa: E402
from rl4co.envs.routing.tsp.env import TSPEnv  # noqa: E402
from rl4co.models.zoo.am.policy import AttentionModelPolicy  # noqa: E402


def main():
    threads = torch.get_num_threads()
    out = {"_how": "median (and minimum) of 6 interleaved rollouts each, torch.inference_mode, same instances / closed-form weights / threads; reference = "
                   "rl4co AttentionModelPolicy imported unmodified via tests/golden/_refshim.py; restatement = "
                   "oracle/torch_cpu_baseline.rollout", "threads": threads, "torch": torch.__version__, "cases": {}}
    for name, env_name, N, B, decode in (("C1 TSP-20 x 128 greedy", "tsp", 20, 128, "greedy"),
                                         ("C2 TSP-100 x 1024 greedy", "tsp", 100, 1024, "greedy"),
                                         ("C3 CVRP-100 x 1024 sampling", "cvrp", 100, 1024, "sampling")):
        Env = {"tsp": TSPEnv, "cvrp": CVRPEnv}[env_name]
        env = Env(generator_params=dict(num_loc=N), seed=1234)
        torch.manual_seed(1234)
        gen = env.generator(batch_size=[B])
        td_ref = env.reset(gen.clone())
        pol = AttentionModelPolicy(env_name=env_name).eval()
        sd = pol.state_dict()
        for k, v in goldweights.fill_state_dict(sd).items():
            sd[k].copy_(torch.from_numpy(v))
        sd_t = {k: v.clone() for k, v in pol.state_dict().items()}
        td_mine = tb.reset_td(env_name, {k: v.clone() for k, v in gen.items()})
        t_ref, t_mine, same = [], [], None
        with torch.inference_mode():
            for it in range(7):
                torch.manual_seed(7)
                t0 = time.perf_counter()
                o_ref = pol(td_ref.clone(), env, phase="test", decode_type=decode)
                t1 = time.perf_counter()
                torch.manual_seed(7)
                o_mine = tb.rollout(sd_t, env_name, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in td_mine.items()},
                                    decode_type=decode)
                t2 = time.perf_counter()
                if it:          # first pass warms both
                    t_ref.append(t1 - t0)
                    t_mine.append(t2 - t1)
                same = bool(o_ref["actions"].shape == o_mine["actions"].shape and torch.equal(o_ref["actions"], o_mine["actions"]))
        r, m = statistics.median(t_ref), statistics.median(t_mine)
        out["cases"][name] = {"reference_s": round(r, 4), "restatement_s": round(m, 4), "ratio": round(m / r, 4),
                              "reference_min_s": round(min(t_ref), 4), "restatement_min_s": round(min(t_mine), 4),
                              "ratio_of_minima": round(min(t_mine) / min(t_ref), 4),
                              "same_tours": same, "decode_steps": int(o_mine["steps"]),
                              "reference_env_steps_per_s": round(B * N / r, 1), "restatement_env_steps_per_s": round(B * N / m, 1)}
        print(name, out["cases"][name], flush=True)
    with open(os.path.join(ROOT, "profiles", "r03_cpu_baseline_validation.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
