#!/bin/bash
# Development only: builds an instrumented copy of the library (per-stage cycle stamps in the resident decode kernel)
# into tools/_stamps/libeamrl_hip.so; use with EAMRL_HIP_LIB=tools/_stamps/libeamrl_hip.so python tools/stamps.py
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_stamps
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -DEAMRL_STAMPS"
for f in abi decode_step rollout_resident env_reward encoder evolution evolution_prize pointer encoder_fused reeval rollout_multistart train_gemm train_norm encoder_attn_mfma augment; do
  /opt/rocm/bin/hipcc $FLAGS -c eam_rl4co_amd/csrc/$f.hip -o tools/_stamps/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_stamps/libeamrl_hip.so tools/_stamps/*.o
rm -f tools/_stamps/*.o
echo built tools/_stamps/libeamrl_hip.so
