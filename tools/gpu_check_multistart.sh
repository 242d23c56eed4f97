#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/check_ms; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "start or multistart or pomo or seeded or eam or train or reeval or reinforce" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/kernel_bench.py train --iters 12 > $OUT/eam.log 2>&1; grep EAM $OUT/eam.log
EAMRL_DEBUG_KEYS=13=1 timeout -k 10 300 python tools/kernel_bench.py train --iters 12 > $OUT/eam_nosplit.log 2>&1; grep EAM $OUT/eam_nosplit.log
timeout -k 10 300 python bench.py --workload pomo100 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pomo100.json 2> $OUT/pomo100.err; python -c "import json; d=json.load(open('$OUT/pomo100.json')); print('pomo100', d['ms_per_step'])"
