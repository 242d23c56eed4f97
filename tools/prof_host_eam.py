"""Development only: cProfile of the host side of EAM training steps (TSP-100, 64 x 100)."""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import eam_rl4co_amd as ea  # noqa: E402
from eam_rl4co_amd import train  # noqa: E402

env = ea.get_env("tsp", generator_params=dict(num_loc=100), seed=3)
pol = ea.AttentionModelPolicy(env_name="tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False).to("cuda")
opt = torch.optim.Adam(pol.parameters(), lr=1e-4)
runner = ea.EA(env, dict(num_generations=3, mutation_rate=0.1, crossover_rate=0.6, selection_rate=0.2))
gen = torch.Generator(device="cuda").manual_seed(5)
td = env.reset(batch_size=[64]).to("cuda")


def step():
    res = train.eam_loss(pol, env, td, runner, num_starts=100, generator=gen)
    opt.zero_grad()
    res["loss"].backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
