"""Development only: the EAM step (TSP-100, 64 x 100) with / without the optimizer step between iterations."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eam_rl4co_amd as ea  # noqa: E402
from eam_rl4co_amd import train  # noqa: E402

N, B, S = 100, 64, 100
env = ea.get_env("tsp", generator_params=dict(num_loc=N), seed=3)
pol = ea.AttentionModelPolicy(env_name="tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False).to("cuda")
opt = torch.optim.Adam(pol.parameters(), lr=1e-4)
runner = ea.EA(env, dict(num_generations=3, mutation_rate=0.1, crossover_rate=0.6, selection_rate=0.2))
gen = torch.Generator(device="cuda").manual_seed(5)
td = env.reset(batch_size=[B]).to("cuda")
for variant in ("no backward", "backward, no optimizer step", "backward + Adam", "backward + Adam (again)"):
    ts = []
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = train.eam_loss(pol, env, td, runner, num_starts=S, generator=gen)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if variant != "no backward":
            opt.zero_grad()
            res["loss"].backward()
            if "Adam" in variant:
                opt.step()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    print(variant, " forward ms:", [round(a, 1) for a, _ in ts], " rest ms:", [round(b, 1) for _, b in ts])
