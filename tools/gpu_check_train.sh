#!/bin/bash
# Development only: training-side parity tests, then the POMO training bench line and the EAM step times.
set -o pipefail
OUT=gpurun_out/train
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_next.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout -k 10 400 python bench.py --workload pomo100_train --steps 5 --warmup 3 --no-cpu-baseline > $OUT/bench_pomo100_train.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench_pomo100_train.json
timeout -k 10 400 python bench.py --workload pomo_cvrp100_train --steps 5 --warmup 3 --no-cpu-baseline > $OUT/bench_pomo_cvrp100_train.json 2>> $OUT/bench.err; cut -c1-400 $OUT/bench_pomo_cvrp100_train.json
timeout -k 10 300 python tools/kernel_bench.py train > $OUT/eam_steps.log 2>&1; grep "EAM" $OUT/eam_steps.log
