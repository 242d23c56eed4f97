"""Development only: host-side cost of one GraphedRollout call (cProfile) -- python tools/prof_host.py [workload]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eam_rl4co_amd as ea  # noqa: E402
from eam_rl4co_amd.policy import GraphedRollout  # noqa: E402

N, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 1024)
env = ea.get_env("tsp", generator_params=dict(num_loc=N))
pol = ea.AttentionModelPolicy(env_name="tsp").eval().to("cuda")
td = env.reset(batch_size=[B]).to("cuda")
g = GraphedRollout(pol, env, td, decode_type="greedy")
for _ in range(5):
    g(td)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    g(td)
torch.cuda.synchronize()
print("ms per call", (time.perf_counter() - t0) / 50 * 1e3)
# host time of the parts
t_copy = t_replay = t_finish = t_clone = 0.0
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    g(td)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
