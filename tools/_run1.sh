set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "exit=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
grep -q "exit=0" gpurun_out/pytest_gpu.log && \
timeout -k 10 300 python bench.py --workload pomo100 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/pomo_b256_ms.json 2> gpurun_out/pomo_b256_ms.err && \
EAMRL_DEBUG_KEYS="6=1" timeout -k 10 300 python bench.py --workload pomo100 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/pomo_b256_single.json 2> gpurun_out/pomo_b256_single.err && \
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/tsp100.json 2> gpurun_out/tsp100.err
cat gpurun_out/pomo_b256_ms.json gpurun_out/pomo_b256_single.json gpurun_out/tsp100.json
