#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c13; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eam -- python3 $R/tools/kernel_bench.py train --iters 12 > $OUT/eam.log 2>&1; echo "rc=$?"
grep "EAM" $OUT/eam.log
f=$(ls $OUT/eam/*/*kernel_stats.csv | head -1); cp $f $OUT/eam_kernel_stats.csv; cut -c1-150 $f | head -14
