#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c12; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --workload pomo100_train --steps 3 --warmup 1 --no-cpu-baseline > $OUT/train.json 2> $OUT/train.err; echo "train rc=$?"
EAMRL_REEVAL_FORWARD=1 timeout -k 10 600 python bench.py --workload pomo100_train --steps 3 --warmup 1 --no-cpu-baseline > $OUT/train_fwd.json 2> $OUT/train_fwd.err; echo "train(with fwd) rc=$?"
python - <<PY
import json
for k in ("train","train_fwd"):
    d=json.load(open("gpurun_out/c12/%s.json"%k)); print(k, "ms/step", d["ms_per_step"], "value", round(d["value"]/1e6,2), "rollout", d["config"].get("rollout_ms"), "grad side", d["config"].get("gradient_side_ms"))
PY
