#!/bin/bash
# Round-2 extras on the MI355X box (through gpurun): kernel stats of the POMO rollout and of the POMO training step, the
# training bench line, HBM traffic of the POMO decode launch (the noise tensor is gone), and the LDS / wait counters of the
# CVRP-500 streaming kernel.  Everything lands in gpurun_out/<tag>/.
#   usage: bash tools/collect_extras.sh r02d
set -o pipefail
TAG=${1:-extras}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 400 python bench.py --workload pomo100_train --steps 5 --warmup 3 > $OUT/bench_pomo100_train.json 2> $OUT/bench_pomo100_train.err || echo "train bench failed"
timeout -k 10 300 python tools/kernel_bench.py train > $OUT/eam_steps.log 2>&1 || echo "eam step bench failed"
tail -4 $OUT/eam_steps.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pomo_stats -- python3 $R/bench.py --workload pomo100 --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $OUT/pomo_stats.log 2>&1 || echo "pomo stats failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -- python3 $R/bench.py --workload pomo100_train --steps 2 --warmup 1 --no-cpu-baseline > $OUT/train_stats.log 2>&1 || echo "train stats failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_pomo100_$c -- python3 $R/bench.py --workload pomo100 --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/pmc_pomo100_$c.log 2>&1 || echo "pmc pomo100 $c failed"
done
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_cvrp500_lds -- python3 $R/bench.py --workload cvrp500 --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/pmc_cvrp500_lds.log 2>&1 || echo "pmc cvrp500 lds failed"
timeout -k 10 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_cvrp500_wait -- python3 $R/bench.py --workload cvrp500 --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/pmc_cvrp500_wait.log 2>&1 || echo "pmc cvrp500 wait failed"
find $OUT -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
cut -c1-300 $OUT/bench_pomo100_train.json
