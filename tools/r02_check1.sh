#!/bin/bash
# round-2 first GPU pass: whole GPU suite, bench lines (rollout + training), two-rank rehearsal on one GPU over gloo
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" ; tail -3 gpurun_out/pytest_gpu.log
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_tsp100.json 2> gpurun_out/bench_tsp100.err; echo "bench rc=$?"; cat gpurun_out/bench_tsp100.json
python bench.py --workload pomo20_train --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_pomo20_train.json 2> gpurun_out/bench_pomo20_train.err; echo "train20 rc=$?"; cat gpurun_out/bench_pomo20_train.json
EAMRL_BENCH_SINGLE_DEVICE=1 EAMRL_DIST_BACKEND=gloo python bench.py --gpus 2 --workload pomo20_train --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_pomo20_train_2rank.json 2> gpurun_out/bench_pomo20_train_2rank.err; echo "train20x2 rc=$?"; cat gpurun_out/bench_pomo20_train_2rank.json; tail -5 gpurun_out/bench_pomo20_train_2rank.err
EAMRL_BENCH_SINGLE_DEVICE=1 EAMRL_DIST_BACKEND=gloo python bench.py --gpus 2 --workload tsp100 --batch 512 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_tsp100_2rank.json 2> gpurun_out/bench_tsp100_2rank.err; echo "tsp100x2 rc=$?"; cat gpurun_out/bench_tsp100_2rank.json
timeout -k 10 600 python bench.py --workload pomo100_train --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_pomo100_train.json 2> gpurun_out/bench_pomo100_train.err; echo "train100 rc=$?"; cat gpurun_out/bench_pomo100_train.json; tail -5 gpurun_out/bench_pomo100_train.err
