#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, two separate --pmc passes as MI355X_MICROARCH.md prescribes) of the decode-loop
# launch for the non-headline workloads.  Runs on the MI355X box through gpurun; output in gpurun_out/<tag>/.
#   usage: bash tools/collect_traffic.sh r01m "cvrp100 cvrp500 pomo100"
set -o pipefail
TAG=${1:-traffic}
WL=${2:-"cvrp100 cvrp500 pomo100"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${w}_$c -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/pmc_${w}_$c.log 2>&1 || echo "pmc $w $c failed"
    echo "$w $c done"
  done
done
find $OUT -name "*counter_collection.csv" | head -20
