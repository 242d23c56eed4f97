set -o pipefail
for v in _v1 _v2 _v3; do
if [ -f tools/$v/libeamrl_hip.so ]; then echo "== $v"; EAMRL_HIP_LIB=$GRAFT_REPO_ROOT/tools/$v/libeamrl_hip.so timeout -k 10 300 python tools/kernel_bench.py decode --iters 10 2>&1 | grep "t_max=100"; fi
done
echo "== production"; timeout -k 10 300 python tools/kernel_bench.py decode --iters 10 2>&1 | grep "t_max=100"
