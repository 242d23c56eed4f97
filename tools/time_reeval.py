"""Development only: times the SDVRP re-evaluation (forward + backward) with the HIP kernels and with the PyTorch fallback
(python tools/time_reeval.py N B [env])."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in ("", "tests", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p_))
import eam_rl4co_amd as ea
from eam_rl4co_amd.train import evaluate_log_likelihood
from test_gpu_parity import make_policy

N, B = int(sys.argv[1]), int(sys.argv[2])
ENV = sys.argv[3] if len(sys.argv) > 3 else "sdvrp"
env = ea.get_env(ENV, generator_params=dict(num_loc=N), seed=1)
torch.manual_seed(0)
td = env.reset(batch_size=[B]).to("cuda")
pol = make_policy("am_" + ENV)
with torch.no_grad():
    out = pol(td, env, phase="train", decode_type="sampling", return_sum_log_likelihood=False)
acts = out["actions"]
print("actions", tuple(acts.shape))
for native in (True, False):
    for it in range(3):
        pol.zero_grad()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lp = evaluate_log_likelihood(pol, td, env, acts, native=native)
        lp.sum().backward()
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{ENV}{N} x {B}: native={native}: forward+backward {1e3*(t1-t0):.1f} ms")
