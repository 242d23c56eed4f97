set -o pipefail
for d in 0 512 1024 2048 4096; do echo -n "dbg=$d  "; timeout -k 10 300 python tools/kernel_bench.py mha --iters 20 --dbg $d 2>&1 | grep mha; done
