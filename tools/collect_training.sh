#!/bin/bash
# Runs on the MI355X box (through gpurun): the whole GPU test suite and smoke, the headline and POMO rollout lines, the two
# POMO training-step lines, the EAM step times and the kernel stats of the training step and of an EAM step.
#   usage: bash tools/collect_training.sh r02h
set -o pipefail
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "exit=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
grep -q "exit=0" $OUT/pytest_gpu.log || exit 1
timeout -k 10 120 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 || exit 1
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > $OUT/bench_tsp100.json 2> $OUT/bench_tsp100.err || exit 1
for w in pomo100 pomo_cvrp100; do
  timeout -k 10 400 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err || echo "bench $w failed"
done
for w in pomo100_train pomo_cvrp100_train; do
  timeout -k 10 400 python bench.py --workload $w --steps 5 --warmup 3 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err || echo "bench $w failed"
done
timeout -k 10 300 python tools/kernel_bench.py train > $OUT/eam_steps.log 2>&1 || echo "eam step bench failed"
grep EAM $OUT/eam_steps.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -- python3 $R/bench.py --workload pomo100_train --steps 2 --warmup 1 --no-cpu-baseline > $OUT/train_stats.log 2>&1 || echo "train stats failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eam_stats -- python3 $R/tools/prof_eam_step.py > $OUT/eam_stats.log 2>&1 || echo "eam stats failed"
cd $R
cut -c1-260 $OUT/bench_*.json
