#!/bin/bash
# fused encoder A/B: bench lines with and without it, same box
set -o pipefail
for w in tsp100 tsp20 cvrp100 pomo100; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b2_${w}_fused.json 2> gpurun_out/b2_${w}_fused.err; echo "$w fused rc=$?"
  EAMRL_FUSED_ENCODER=0 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b2_${w}_unfused.json 2> gpurun_out/b2_${w}_unfused.err; echo "$w unfused rc=$?"
  python - <<PY
import json
for k in ("fused","unfused"):
    try:
        d=json.load(open("gpurun_out/b2_${w}_%s.json"%k))
        r=d["roofline"]; e=d["roofline_encoder_fused"]; g=d["roofline_gemm"]; a=d["roofline_attention"]
        print("${w}",k,"ms/step",d["ms_per_step"],"value",d["value"],"| enc_fused ms",e["kernel_ms"],"TF",e["achieved"],"| gemm ms",g["ms_per_step"],"att ms",a["ms_per_step"],"decode ms",d["roofline_decode"]["kernel_ms"])
    except Exception as ex:
        print("${w}",k,"failed",ex)
PY
done
