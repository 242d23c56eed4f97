#!/bin/bash
# full GPU suite + bench lines of all workloads (fused encoder + fused cache)
set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
for w in tsp100 tsp20 cvrp100 cvrp500 pomo100 sdvrp100 pctsp100 op100 cvrptw100; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b4_${w}.json 2> gpurun_out/b4_${w}.err; echo "$w rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/b4_${w}.json")); e=d["roofline_encoder_fused"]
    print("${w} ms/step",d["ms_per_step"],"value",d["value"],"| enc_fused ms",e["kernel_ms"],"TF",e["achieved"],"| gemm ms",d["roofline_gemm"]["ms_per_step"],"att",d["roofline_attention"]["ms_per_step"],"decode ms",d["roofline_decode"]["kernel_ms"])
except Exception as ex: print("${w} failed", ex)
PY
done
