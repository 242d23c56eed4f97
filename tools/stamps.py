"""Development only: per-stage cycle breakdown of the register-resident decode kernel (wavefront 0's view).
Needs the instrumented library: bash tools/build_stamps.sh; EAMRL_HIP_LIB=tools/_stamps/libeamrl_hip.so python tools/stamps.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eam_rl4co_amd import _lib, ops  # noqa: E402
import eam_rl4co_amd as ea  # noqa: E402
from eam_rl4co_amd.policy import state_from_td  # noqa: E402


def main():
    lib = _lib.load()
    B, M, E = 1024, 100, 128
    env = ea.get_env("tsp", generator_params=dict(num_loc=M))
    td0 = env.reset(batch_size=[B]).to("cuda")
    buf = torch.randn(B, M, 6 * E, device="cuda") * 0.3
    cache = ops.DecodeCache("tsp", buf, torch.randn(E, device="cuda"), torch.randn(B, E, device="cuda"),
                            torch.randn(B, M, E, device="cuda"), 8)
    out = (C.c_ulonglong * 8)()
    for it in range(3):
        st = state_from_td("tsp", td0.clone(), 0)
        torch.cuda.synchronize()
        lib.eamrl_debug_read_stamps(out, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.rollout(st, cache, "greedy", t_max=M)
        e1.record()
        torch.cuda.synchronize()
        lib.eamrl_debug_read_stamps(out, 0)
        steps = out[5]
        names = ["S1 scores+softmax w (+barrier)", "S2 glimpse partials (+barrier)", "S4 heads+logit partials (+barrier)",
                 "S5 finish (wave 0)", "final barrier"]
        tot = sum(out[i] for i in range(5))
        print(f"run {it}: kernel {e0.elapsed_time(e1)*1e3:.0f} us, row-steps {steps}, cycles/step {tot/steps:.0f} (s_memtime ticks)")
        for i, n in enumerate(names):
            print(f"   {n:40s} {out[i]/steps:8.0f} ticks/step  {100*out[i]/tot:5.1f} %")


if __name__ == "__main__":
    main()
