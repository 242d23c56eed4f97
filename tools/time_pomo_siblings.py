"""Development only: multistart rollouts of the sibling envs (1024 x 100 starts, POMO-style policy), MFMA start-sharing kernel vs
the VALU one (debug key 14).  python tools/time_pomo_siblings.py [env ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eam_rl4co_amd as ea  # noqa: E402
from eam_rl4co_amd import _lib  # noqa: E402

N, B, S = 100, 1024, 100
for env_name in (sys.argv[1:] or ["cvrptw", "pctsp", "op"]):
    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=1)
    pol = ea.AttentionModelPolicy(env_name=env_name, num_encoder_layers=6, normalization="instance", use_graph_context=False).eval().to("cuda")
    td = env.reset(batch_size=[B]).to("cuda")
    starts = (torch.arange(S, device="cuda").repeat_interleave(B) % N) + 1
    ok = td["action_mask"].repeat(S, 1).gather(1, starts[:, None]).squeeze(1)
    starts = torch.where(ok, starts, td["action_mask"][:, 1:].float().argmax(1).repeat(S) + 1)
    kw = dict(num_starts=S, select_start_nodes_fn=lambda td_, env_, n: starts)
    for key, label in ((0, "MFMA start-sharing kernel"), (1, "VALU start-sharing kernel")):
        _lib.load().eamrl_debug_set(14, key)
        for mode in ("multistart_greedy", "multistart_sampling"):
            with torch.no_grad():
                for _ in range(2):
                    out = pol(td.clone(), env, phase="test", decode_type=mode, **kw)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 3
                for _ in range(n):
                    out = pol(td.clone(), env, phase="test", decode_type=mode, **kw)
                torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            print(f"POMO {env_name.upper()}-100 x {B} x {S} {mode}: {label}: {ms:.1f} ms per rollout ({out['actions'].shape[1]} steps)", flush=True)
    _lib.load().eamrl_debug_set(14, 0)
