#!/bin/bash
# full GPU suite + the headline benches (after a change on the common rollout path)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/check_all; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -5 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
for w in tsp100 tsp20 cvrp100 pomo100; do
  timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/$w.json 2> $OUT/$w.err; echo "$w rc=$?"
done
python - <<PY
import json
for w in ("tsp100","tsp20","cvrp100","pomo100"):
    d=json.load(open("gpurun_out/check_all/%s.json"%w)); print(w, "ms/step", d["ms_per_step"], "value", round(d["value"]/1e6,2), "M/s  shares", d["roofline"].get("share_of_step"))
PY
