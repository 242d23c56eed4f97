set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "exit=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
grep -q "exit=0" gpurun_out/pytest_gpu.log && \
timeout -k 10 300 python tools/kernel_bench.py mha --iters 20 2>&1 | grep mha && \
timeout -k 10 300 python tools/kernel_bench.py decode --iters 10 2>&1 | grep "t_max=100" && \
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/tsp100.json 2> gpurun_out/tsp100.err
cut -c1-330 gpurun_out/tsp100.json
