#!/bin/bash
# Development only: builds a copy of the library with extra -D flags into tools/_variants/<name>/libeamrl_hip.so, for
# A/B measurements inside one gpurun call (EAMRL_HIP_LIB=tools/_variants/<name>/libeamrl_hip.so python bench.py ...).
#   usage: bash tools/build_variant.sh du5_4 -DEAMRL_DU5=4
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
D=tools/_variants/$NAME
mkdir -p $D
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden $*"
for f in abi decode_step rollout_resident env_reward encoder evolution; do
  if [ "$f" = decode_step ] || [ ! -f eam_rl4co_amd/lib/obj/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c eam_rl4co_amd/csrc/$f.hip -o $D/$f.o &
  else
    cp eam_rl4co_amd/lib/obj/$f.o $D/$f.o      # unchanged translation units: reuse the objects of the regular build
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libeamrl_hip.so $D/*.o
rm -f $D/*.o
echo built $D/libeamrl_hip.so
