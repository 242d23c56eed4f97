"""Development only: cost of return_entropy=True, single-launch path vs the host-driven step loop."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eam_rl4co_amd as ea  # noqa: E402

for env_name, N, B, S in (("tsp", 100, 1024, 0), ("tsp", 100, 64, 100), ("cvrp", 100, 1024, 0)):
    env = ea.get_env(env_name, generator_params=dict(num_loc=N))
    kw = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False) if S else {}
    pol = ea.AttentionModelPolicy(env_name=env_name, **kw).eval().to("cuda")
    td = env.reset(batch_size=[B]).to("cuda")
    dk = dict(decode_type="multistart_sampling", num_starts=S) if S else dict(decode_type="sampling")
    for mode, label in (("0", "single launch + one re-evaluation pass"), ("1", "host-driven step loop")):
        os.environ["EAMRL_ENTROPY_STEPWISE"] = mode
        with torch.no_grad():
            for _ in range(2):
                pol(td.clone(), env, phase="train", return_entropy=True, **dk)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                out = pol(td.clone(), env, phase="train", return_entropy=True, **dk)
            torch.cuda.synchronize()
        print(f"{env_name}{N} B={B} S={S}: return_entropy=True {label}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms, "
              f"mean entropy {float(out['entropy'].mean()):.3f}")
    os.environ.pop("EAMRL_ENTROPY_STEPWISE", None)
