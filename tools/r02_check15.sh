#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c15; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --workload cvrp500 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/cvrp500.json 2> $OUT/cvrp500.err; python -c "import json; d=json.load(open('$OUT/cvrp500.json')); print('cvrp500', d['ms_per_step'], 'decode', d['roofline_decode']['kernel_ms'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_lds -- python3 $R/bench.py --workload cvrp500 --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/pmc_lds.log 2>&1 || echo "pmc failed"
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/pmc_lds/*/*counter_collection.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_rollout_stream' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items(): print(k, sum(v)/len(v))
PY
