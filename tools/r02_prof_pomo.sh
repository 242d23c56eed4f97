#!/bin/bash
# kernel-trace stats of the POMO rollout (configs[3] rollout) and of the POMO training step, current code
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_pomo; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rollout -- python3 $R/bench.py --workload pomo100 --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $OUT/rollout.log 2>&1; echo "rollout rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 $R/bench.py --workload pomo100_train --steps 2 --warmup 1 --no-cpu-baseline > $OUT/train.log 2>&1; echo "train rc=$?"
for d in rollout train; do f=$(ls $OUT/$d/*/*kernel_stats.csv | head -1); cp $f $OUT/${d}_kernel_stats.csv; echo "== $d"; head -14 $f | cut -c1-150; done
tail -2 $OUT/train.log | cut -c1-400
