"""Development only: phase breakdown of the fused encoder kernel (needs tools/build_stamps.sh;
EAMRL_HIP_LIB=tools/_stamps/libeamrl_hip.so python tools/stamps_enc.py [M] [B])."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", "tests/golden"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), p))
from eam_rl4co_amd import _lib  # noqa: E402
import eam_rl4co_amd as ea  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    lib = _lib.load()
    pol = ea.AttentionModelPolicy(env_name="tsp").eval().to("cuda")
    h = torch.randn(B, M, 128, device="cuda") * 0.5
    out = (C.c_ulonglong * 24)()
    names = ["load h", "P1 epilogue (q k v store)", "  barrier", "P2 attention", "  barrier", "P3 out_proj partial", "  barrier",
             "norm1", "P4 epilogue (relu store)", "  barrier", "P5 ffn2 partial", "  barrier", "norm2", "store h",
             "P4 prologue (b0, bias)", "P4 gemm", "P1 prologue", "P1 gemm", "-", "-"]
    for it in range(3):
        torch.cuda.synchronize()
        lib.eamrl_debug_read_enc_stamps(out, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            pol.encoder.net(h)
        e1.record()
        torch.cuda.synchronize()
        lib.eamrl_debug_read_enc_stamps(out, 0)
        waves = out[20]
        tot = sum(out[i] for i in range(20))
        print(f"run {it}: kernel {e0.elapsed_time(e1)*1e3:.0f} us, waves {waves}, ticks/wave {tot/waves:.0f}")
        for i, n in enumerate(names):
            print(f"   {n:28s} {out[i]/waves:9.0f} ticks/wave  {100*out[i]/tot:5.1f} %")


if __name__ == "__main__":
    main()
