#!/bin/bash
set -o pipefail
python bench.py --workload pomo100 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b6_pomo100.json 2> gpurun_out/b6_pomo100.err; echo "pomo100 rc=$?"
EAMRL_DEBUG_KEYS=11=1 python bench.py --workload pomo100 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b6_pomo100_valu.json 2> gpurun_out/b6_pomo100_valu.err; echo "pomo100 valu rc=$?"
python - <<PY
import json
for k in ("", "_valu"):
    d=json.load(open("gpurun_out/b6_pomo100%s.json"%k)); print("pomo100"+k, "ms/step", d["ms_per_step"], "value", d["value"], "decode ms", d["roofline_decode"]["kernel_ms"], "issue frac", d["roofline_decode"]["issue_bound"]["frac"])
PY
timeout -k 10 600 python bench.py --workload pomo100_train --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b6_pomo100_train.json 2> gpurun_out/b6_pomo100_train.err; echo "train rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/b6_pomo100_train.json")); print("train: ms/step",d["ms_per_step"],"rollout",d["config"]["rollout_ms"],"grad side",d["config"]["gradient_side_ms"],"value",d["value"])
PY
