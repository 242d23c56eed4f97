#!/bin/bash
# cvrp100 sampling: where the time goes (kernel stats, eager launches) + streaming kernel at 4 workgroups per CU (debug key 12)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/c7; mkdir -p $OUT
cd $R
for k in "" "12=1"; do
  EAMRL_DEBUG_KEYS=$k timeout -k 10 300 python bench.py --workload cvrp500 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/cvrp500_$k.json 2> $OUT/cvrp500_$k.err; echo "cvrp500 key=$k rc=$?"
done
python - <<PY
import json
for k in ("", "12=1"):
    d=json.load(open("gpurun_out/c7/cvrp500_%s.json"%k)); print("cvrp500 key", k, "ms/step", d["ms_per_step"], "decode ms", d["roofline_decode"]["kernel_ms"])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --workload cvrp100 --steps 5 --warmup 1 --no-cpu-baseline --no-graph > $OUT/stats.log 2>&1 || echo "stats pass failed"
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cut -c1-160 $f | head -14
