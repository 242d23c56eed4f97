"""Development only: where the forward half of the EAM step (TSP-100, 64 x 100) spends its time."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eam_rl4co_amd as ea  # noqa: E402
from eam_rl4co_amd import train  # noqa: E402

N, B, S = 100, 64, 100
env = ea.get_env("tsp", generator_params=dict(num_loc=N), seed=3)
pol = ea.AttentionModelPolicy(env_name="tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False).to("cuda")
runner = ea.EA(env, dict(num_generations=3, mutation_rate=0.1, crossover_rate=0.6, selection_rate=0.2))
gen = torch.Generator(device="cuda").manual_seed(5)
td = env.reset(batch_size=[B]).to("cuda")


def sync_time(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


ms, out = sync_time(lambda: pol(td, env, phase="train", decode_type="multistart_sampling", num_starts=S))
print(f"policy(train, grad) sampled rollout + graph: {ms:.2f} ms")
with torch.no_grad():
    ms, out_ng = sync_time(lambda: pol(td, env, phase="train", decode_type="multistart_sampling", num_starts=S))
print(f"policy(no grad) sampled rollout:            {ms:.2f} ms")
acts = out_ng["actions"]
ms, imp = sync_time(lambda: ea.evolution_worker(acts, td, runner, env, generator=gen))
print(f"evolution_worker:                            {ms:.2f} ms")
improved = torch.cat([acts[:, :1], imp[0]], -1)
ms, _ = sync_time(lambda: pol(td, env, phase="train", actions=improved, num_starts=S))
print(f"policy(train, grad, actions=improved):       {ms:.2f} ms")
with torch.no_grad():
    ms, _ = sync_time(lambda: pol(td, env, phase="train", actions=improved, num_starts=S))
print(f"policy(no grad, actions=improved):           {ms:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    pol(td, env, phase="train", decode_type="multistart_sampling", num_starts=S)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
