#!/bin/bash
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_encoder_fused.py -x -q -m gpu > gpurun_out/t3.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t3.log
for w in tsp100 tsp20; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b3_${w}.json 2> gpurun_out/b3_${w}.err; echo "$w rc=$?"
  python - <<PY
import json
d=json.load(open("gpurun_out/b3_${w}.json")); e=d["roofline_encoder_fused"]
print("${w} ms/step",d["ms_per_step"],"| enc_fused ms",e["kernel_ms"],"TF",e["achieved"],"| gemm ms",d["roofline_gemm"]["ms_per_step"],"decode ms",d["roofline_decode"]["kernel_ms"])
PY
done
EAMRL_HIP_LIB=tools/_stamps/libeamrl_hip.so python tools/stamps_enc.py 100 1024 > gpurun_out/stamps_enc.log 2>&1; tail -15 gpurun_out/stamps_enc.log
