set -o pipefail
EAMRL_HIP_LIB=$GRAFT_REPO_ROOT/tools/_stamps/libeamrl_hip.so timeout -k 10 300 python tools/stamps.py > gpurun_out/stamps.txt 2>&1
tail -6 gpurun_out/stamps.txt
if [ -f tools/_noslp/libeamrl_hip.so ]; then
EAMRL_HIP_LIB=$GRAFT_REPO_ROOT/tools/_noslp/libeamrl_hip.so timeout -k 10 300 python tools/stamps.py > gpurun_out/stamps_noslp.txt 2>&1
echo NOSLP; tail -6 gpurun_out/stamps_noslp.txt
fi
timeout -k 10 300 python tools/kernel_bench.py decode --iters 10 > gpurun_out/kb_decode.txt 2>&1; head -3 gpurun_out/kb_decode.txt
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "rollout or policy or start_sharing" > gpurun_out/pytest_gpu.log 2>&1; echo "exit=$?" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
