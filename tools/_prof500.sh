set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/c500; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --workload cvrp500 --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $OUT/stats.log 2>&1 || echo failed
find $OUT -name "*kernel_stats.csv" | head -2
