#!/bin/bash
set -o pipefail
timeout -k 10 600 python bench.py --workload pomo100_train --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b5_pomo100_train.json 2> gpurun_out/b5_pomo100_train.err; echo "train100 native rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/b5_pomo100_train.json")); print("native: ms/step",d["ms_per_step"],"rollout",d["config"]["rollout_ms"],"grad side",d["config"]["gradient_side_ms"],"value",d["value"])
PY
timeout -k 10 300 python bench.py --workload pomo20_train --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/b5_pomo20_train.json 2> gpurun_out/b5_pomo20_train.err; echo "train20 rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/b5_pomo20_train.json")); print("pomo20 native: ms/step",d["ms_per_step"],"rollout",d["config"]["rollout_ms"],"grad side",d["config"]["gradient_side_ms"])
PY
python tools/kernel_bench.py train > gpurun_out/b5_eam.log 2>&1; tail -8 gpurun_out/b5_eam.log
