#!/bin/bash
# Round-3 evidence in one gpurun call: GPU suite + smoke, bench lines of every workload, rocprofv3 kernel stats, the two
# HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes) and the MFMA-pipe counters (tools/collect_mfma_counters.sh).
#   usage: bash tools/collect_r03.sh r03q
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "exit=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
grep -q "exit=0" $OUT/pytest_gpu.log || exit 1
timeout -k 10 120 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 || exit 1
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > $OUT/bench_tsp100.json 2> $OUT/bench_tsp100.err || exit 1
for w in tsp20 cvrp100 cvrp500 pomo100 pomo_cvrp100 sdvrp100 pctsp100 op100 cvrptw100 pomo100_train; do
  timeout -k 10 400 python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err || echo "bench $w failed"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-graph > $OUT/stats.log 2>&1 || echo "stats pass failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $OUT/pmc_$c.log 2>&1 || echo "pmc $c failed"
done
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_lds -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $OUT/pmc_lds.log 2>&1 || echo "pmc lds failed"
cd $R
bash tools/collect_mfma_counters.sh $TAG "tsp100 pomo100 pomo100_train" > $OUT/mfma.log 2>&1 || echo "mfma counters failed"
cut -c1-260 $OUT/bench_*.json
