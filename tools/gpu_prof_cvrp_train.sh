#!/bin/bash
# Development only: kernel stats of the POMO CVRP-100 training step.
set -o pipefail
OUT=gpurun_out/cvrp_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/bench.py --workload pomo_cvrp100_train --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/stats.log 2>&1
cd $R
f=$(ls $OUT/stats/*/*kernel_stats.csv | head -1); cp $f $OUT/kernel_stats.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/cvrp_train/kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"kernel time total {tot/1e6:.1f} ms (3 steps + 3 rollout-only passes)")
for r in rows[:24]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e6:9.2f} ms", f"{float(r['AverageNs'])/1e3:9.1f} us")
PY
