#!/bin/bash
# kernel-trace stats of the C4 training step
set -o pipefail
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -- python3 $GRAFT_REPO_ROOT/bench.py --workload pomo100_train --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1
echo "rc=$?"
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_train/*/*kernel_stats.csv | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/prof_train_kernel_stats.csv
head -25 $f | cut -c1-160
