#!/bin/bash
# MFMA-pipe counters for the MFMA kernels: k_encoder_fused (tsp100), k_rollout_ms_mfma (pomo100),
# k_reeval_bwd_* (pomo100_train).  One --pmc pass per workload, program directly after `--`; kernel stats of the same
# command in a separate pass.  Output in gpurun_out/<tag>/; tools/summarize_counters.py turns the CSVs into
# profiles/<tag>_counters.json.
#   usage: bash tools/collect_mfma_counters.sh r03a "tsp100 pomo100 pomo100_train"
set -o pipefail
TAG=${1:-mfma}
WL=${2:-"tsp100 pomo100 pomo100_train"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  extra="--no-graph"; case $w in *_train) extra="";; esac
  timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $OUT/pmc_mfma_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline $extra > $OUT/pmc_mfma_$w.log 2>&1 || echo "pmc mfma $w failed"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 $R/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline $extra > $OUT/stats_$w.log 2>&1 || echo "stats $w failed"
  echo "$w done"
done
python3 $R/tools/summarize_counters.py $OUT > $OUT/counters.json && cat $OUT/counters.json
