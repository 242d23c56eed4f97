"""Development only: POMO rollout on CVRP-100 (1024 x 100 starts), MFMA start-sharing kernel vs the VALU one."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eam_rl4co_amd as ea  # noqa: E402
from eam_rl4co_amd import _lib  # noqa: E402

N, B, S = 100, 1024, 100
env = ea.get_env("cvrp", generator_params=dict(num_loc=N), seed=1)
pol = ea.AttentionModelPolicy(env_name="cvrp", num_encoder_layers=6, normalization="instance", use_graph_context=False).eval().to("cuda")
td = env.reset(batch_size=[B]).to("cuda")
for key, label in ((0, "MFMA start-sharing kernel"), (1, "VALU start-sharing kernel")):
    _lib.load().eamrl_debug_set(14, key)
    for mode in ("multistart_greedy", "multistart_sampling"):
        with torch.no_grad():
            for _ in range(2):
                out = pol(td.clone(), env, phase="test", decode_type=mode, num_starts=S)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 4
            for _ in range(n):
                out = pol(td.clone(), env, phase="test", decode_type=mode, num_starts=S)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"POMO CVRP-100 x {B} x {S} {mode}: {label}: {ms:.1f} ms per rollout ({out['actions'].shape[1]} steps), "
              f"{B * S * N / ms / 1e3:.0f} M env-steps/s")
_lib.load().eamrl_debug_set(14, 0)
