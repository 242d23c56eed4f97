"""Development only: per-stage cycle breakdown of the streaming decode kernel (thread 0's view of each row-step).
Needs the instrumented library: bash tools/build_stamps.sh; EAMRL_HIP_LIB=tools/_stamps/libeamrl_hip.so python tools/stamps_stream.py [cvrp500|tsp100]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from eam_rl4co_amd import _lib  # noqa: E402
import eam_rl4co_amd as ea  # noqa: E402
import bench  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cvrp500"
    env_name, num_loc, batch, decode_type = bench.WORKLOADS[wl]
    lib = _lib.load()
    lib.eamrl_debug_set(1, 1)               # streaming kernel
    env = ea.get_env(env_name, generator_params=dict(num_loc=num_loc), seed=1234)
    td = env.reset(batch_size=[batch]).to("cuda")
    pol = bench.build_policy(env_name, torch.device("cuda"))
    out = (C.c_ulonglong * 16)()
    for it in range(2):
        lib.eamrl_debug_read_stream_stamps(out, 1)
        o = pol(td, env, phase="test", decode_type=decode_type)
        torch.cuda.synchronize()
        lib.eamrl_debug_read_stream_stamps(out, 0)
    T = o["actions"].shape[1]
    names = ["D1+D2 query, scores (K loads)", "D3 softmax weights", "D4 glimpse (V loads) + heads", "D5 logit partials (Lp loads)",
             "D6-D8 clip, log-softmax, selection"]
    tot = sum(out[i] for i in range(5))
    rows_steps = batch * T
    print(f"{wl}: {T} steps, decode-row cycles per row-step {tot / rows_steps:.0f} (env step not included)")
    for i, n in enumerate(names):
        print(f"   {n:40s} {out[i] / rows_steps:9.0f} cycles  {100 * out[i] / tot:5.1f} %")


if __name__ == "__main__":
    main()
