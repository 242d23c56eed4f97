"""Builds libeamrl_hip.so (gfx950) in-tree with hipcc; no CPU fallback exists.

    python -m eam_rl4co_amd.build [--force]

-ffp-contract=off and correctly-rounded divide/sqrt are part of the numerical contract
(DESIGN.md "Canonical arithmetic"): the kernels spell every fused operation explicitly.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(PKG, "lib", "obj")
LIB = os.path.join(LIBDIR, "libeamrl_hip.so")

SOURCES = ["abi.hip", "decode_step.hip", "rollout_resident.hip", "env_reward.hip", "encoder.hip", "evolution.hip", "evolution_prize.hip", "pointer.hip", "encoder_fused.hip", "reeval.hip", "rollout_multistart.hip", "train_gemm.hip", "train_norm.hip", "encoder_attn_mfma.hip", "augment.hip"]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libeamrl_hip.so cannot be built (ROCm toolchain required)")
    return exe


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(PKG), "include", "eamrl.h"))
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + headers):
            cmd = [cc] + FLAGS + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj, True
        return obj, False

    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or not os.path.exists(LIB):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
