"""Development only: phase breakdown of the MFMA start-sharing rollout kernel (needs tools/build_stamps.sh;
EAMRL_HIP_LIB=tools/_stamps/libeamrl_hip.so python tools/stamps_ms.py [N] [B] [S] [mode])."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eam_rl4co_amd import _lib  # noqa: E402
import eam_rl4co_amd as ea  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    mode = sys.argv[4] if len(sys.argv) > 4 else "multistart_sampling"
    lib = _lib.load()
    env = ea.get_env("tsp", generator_params=dict(num_loc=N))
    pol = ea.AttentionModelPolicy(env_name="tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False).eval().to("cuda")
    td = env.reset(batch_size=[B]).to("cuda")
    out = (C.c_ulonglong * 24)()
    names = ["q tile build", "  barrier", "scores mfma + mask + max", "softmax exps", "value mfma + store", "  barrier",
             "logit mfma", "noise", "tanh / mask / tile max", "  barrier", "exp + tile sum", "  barrier",
             "lse + selection (+2 barriers when sampling)", "  barrier", "env transition", "-"]
    for it in range(3):
        torch.cuda.synchronize()
        lib.eamrl_debug_read_ms_stamps(out, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            pol(td.clone(), env, phase="test", decode_type=mode, num_starts=S)
        e1.record()
        torch.cuda.synchronize()
        lib.eamrl_debug_read_ms_stamps(out, 0)
        waves = out[16]
        tot = sum(out[i] for i in range(16))
        print(f"run {it}: forward {e0.elapsed_time(e1):.2f} ms, waves {waves}, ticks/wave {tot/waves:.0f}")
        for i, n in enumerate(names):
            print(f"   {n:46s} {out[i]/waves:11.0f} ticks/wave  {100*out[i]/tot:5.1f} %")


if __name__ == "__main__":
    main()
