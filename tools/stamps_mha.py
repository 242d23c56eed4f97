"""Development only: phase breakdown of the matrix-core attention kernel (needs tools/build_stamps.sh)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eam_rl4co_amd import _lib, ops  # noqa: E402


def main():
    lib = _lib.load()
    qkv = torch.randn(1024, 100, 384, device="cuda")
    out = (C.c_ulonglong * 8)()
    for it in range(3):
        torch.cuda.synchronize()
        lib.eamrl_debug_read_mha_stamps(out, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.mha_encoder(qkv, 8)
        e1.record()
        torch.cuda.synchronize()
        lib.eamrl_debug_read_mha_stamps(out, 0)
        waves = out[6]
        names = ["stage q,k,v -> LDS (+barrier)", "operand registers from LDS", "q operand + score MFMAs (all tiles)",
                 "softmax (all tiles)", "value MFMAs (all tiles)", "divide + store (all tiles)"]
        tot = sum(out[i] for i in range(6))
        print(f"run {it}: kernel {e0.elapsed_time(e1)*1e3:.0f} us, waves {waves}, cycles/wave {tot/waves:.0f}")
        for i, n in enumerate(names):
            print(f"   {n:40s} {out[i]/waves:8.0f} cycles/wave  {100*out[i]/tot:5.1f} %")


if __name__ == "__main__":
    main()
