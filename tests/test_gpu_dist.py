"""GPU: the collective of the training step on RCCL (torch.distributed backend "nccl" IS RCCL on ROCm).

A one-GPU box can hold one RCCL rank, so this is the world-size-1 case: it proves that the code path the 8-GPU run takes --
`dist.init_distributed("nccl")`, the flat-buffer all-reduce of `PolicyGradientStep`, the rank-0 broadcast, `sync_metrics` --
initialises RCCL on the MI355X and runs its kernels; the multi-rank arithmetic (means over ranks, identical parameters) is
covered on CPU with gloo (tests/test_dist_gloo.py).  Runs in a child process (a process group is process-global state)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, torch
sys.path.insert(0, os.environ["EAMRL_ROOT"])
import torch.distributed as dist
import eam_rl4co_amd as ea
from eam_rl4co_amd import dist as ed
from eam_rl4co_amd.train import PolicyGradientStep

os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["EAMRL_PORT"])
assert ed.init_distributed("nccl") == (0, 1, 0)               # world size 1: init_distributed leaves the group to the caller
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
env = ea.get_env("tsp", generator_params=dict(num_loc=20), seed=1)
pol = ea.AttentionModelPolicy(env_name="tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False).to("cuda").train()
step = PolicyGradientStep(pol, env, num_starts=20)            # broadcast_module_state over RCCL
td = env.reset(batch_size=[8]).to("cuda")
before = torch.cat([p.detach().reshape(-1) for p in pol.parameters()]).clone()
out = step(td)                                                # rollout, backward, ONE RCCL all-reduce, clip, Adam
torch.cuda.synchronize()
after = torch.cat([p.detach().reshape(-1) for p in pol.parameters()])
assert torch.isfinite(after).all() and not torch.equal(before, after)
assert float(out["grad_norm"]) > 0
m = ed.sync_metrics(out, "train")                             # scalar all-reduce on the device
assert sorted(m) == ["train/loss", "train/reward"] and abs(m["train/reward"] - float(out["reward"].mean())) < 1e-4
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK", m)
"""


def test_training_step_collective_runs_on_rccl():
    import socket

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, EAMRL_ROOT=root, EAMRL_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
