#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/c8; mkdir -p $OUT
cd $R
timeout -k 10 300 python bench.py --workload cvrp100 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/g.json 2> $OUT/g.err; echo "graph rc=$?"
timeout -k 10 300 python bench.py --workload cvrp100 --steps 10 --warmup 2 --no-cpu-baseline --no-graph > $OUT/e.json 2> $OUT/e.err; echo "eager rc=$?"
python - <<PY
import json
for k in "ge":
    d=json.load(open("gpurun_out/c8/%s.json"%k)); print(k, "ms/step", d["ms_per_step"], d["config"]["launch"])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload cvrp100 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || echo "trace pass failed"
ls $OUT/trace/*
