#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c11; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in tsp100 cvrp100; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -- python3 $R/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-graph > $OUT/$w.log 2>&1; echo "$w rc=$?"
f=$(ls $OUT/$w/*/*kernel_stats.csv | head -1); cp $f $OUT/${w}_kernel_stats.csv; cp $(ls $OUT/$w/*/*kernel_trace.csv | head -1) $OUT/${w}_kernel_trace.csv
cut -c1-140 $f | head -16
done
