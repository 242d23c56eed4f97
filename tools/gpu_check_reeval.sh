#!/bin/bash
# Development only: re-evaluation kernels -- parity tests, then old-vs-new timing, then kernel stats of both.
set -o pipefail
OUT=gpurun_out/reeval
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout -k 10 300 python tools/time_reeval_bwd.py > $OUT/time.log 2>&1; grep -v amdgpu.ids $OUT/time.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/tools/time_reeval_bwd.py 1 > $R/$OUT/stats.log 2>&1
cd $R
f=$(ls $OUT/stats/*/*kernel_stats.csv | head -1)
head -16 $f | cut -c1-200
