"""Development only: EAM training steps at one size (default TSP-100, 64 instances x 100 starts) for rocprofv3 --kernel-trace
--stats; prints wall time per step so that GPU-busy time (sum of kernel durations / steps) can be set against it."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import eam_rl4co_amd as ea  # noqa: E402
from eam_rl4co_amd import train  # noqa: E402

env_name, N, B, S = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("tsp", 100, 64, 100)
steps = 10
env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=3)
pol = ea.AttentionModelPolicy(env_name=env_name, num_encoder_layers=6, normalization="instance", use_graph_context=False).to("cuda")
opt = torch.optim.Adam(pol.parameters(), lr=1e-4)
runner = ea.EA(env, dict(num_generations=3, mutation_rate=0.1, crossover_rate=0.6, selection_rate=0.2))
gen = torch.Generator(device="cuda").manual_seed(5)
td = env.reset(batch_size=[B]).to("cuda")
ts = []
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = train.eam_loss(pol, env, td, runner, num_starts=S, generator=gen)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    opt.zero_grad(); res["loss"].backward(); opt.step()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((round((t1 - t0) * 1e3, 1), round((t2 - t1) * 1e3, 1)))
print(f"{env_name}{N} B={B} S={S}: (forward ms, backward + Adam ms) per step:", ts, flush=True)
