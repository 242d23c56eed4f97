"""Development only: time of the backward pass of the re-evaluation (logits + glimpse + gather kernels, then the encoder's
autograd) at the POMO training size; run under rocprofv3 --kernel-trace --stats for the per-kernel times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eam_rl4co_amd as ea  # noqa: E402
from eam_rl4co_amd.train import evaluate_log_likelihood  # noqa: E402

cases = [("tsp", 100, 1024, 100), ("cvrp", 100, 256, 100), ("tsp", 50, 64, 50)]
if len(sys.argv) > 1:
    cases = cases[: int(sys.argv[1])]
for env_name, N, B, S in cases:
    env = ea.get_env(env_name, generator_params=dict(num_loc=N))
    pol = ea.AttentionModelPolicy(env_name=env_name, num_encoder_layers=6, normalization="instance",
                                  use_graph_context=False).train().to("cuda")
    td = env.reset(batch_size=[B]).to("cuda")
    with torch.no_grad():
        out = pol(td.clone(), env, phase="train", decode_type="multistart_sampling", num_starts=S,
                  return_sum_log_likelihood=False)
    acts, rl = out["actions"], out["log_likelihood"]
    w = torch.randn(acts.shape, device="cuda")
    ms = []
    for it in range(4):
        pol.zero_grad()
        lp = evaluate_log_likelihood(pol, td, env, acts, num_starts=S, native=True, rollout_logp=rl)
        loss = (lp * w).sum()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        loss.backward()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    print(f"{env_name}{N} B={B} S={S}: whole backward (re-evaluation kernels + encoder autograd) {min(ms[1:]):.2f} ms", flush=True)
