#!/bin/bash
# Development only: training-side parity tests, then the POMO training bench line and the EAM step times.
set -o pipefail
OUT=gpurun_out/train
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_next.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout -k 10 400 python bench.py --workload pomo100_train --steps 5 --warmup 3 --no-cpu-baseline > $OUT/bench_pomo100_train.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench_pomo100_train.json
EAMRL_TORCH_LINEAR=1 timeout -k 10 400 python bench.py --workload pomo100_train --steps 5 --warmup 3 --no-cpu-baseline > $OUT/bench_pomo100_train_torchlinear.json 2>> $OUT/bench.err; cut -c1-400 $OUT/bench_pomo100_train_torchlinear.json
timeout -k 10 300 python tools/kernel_bench.py train > $OUT/eam_steps.log 2>&1; grep "EAM" $OUT/eam_steps.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/train_stats -- python3 $R/bench.py --workload pomo100_train --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/train_stats.log 2>&1
cd $R
f=$(ls $OUT/train_stats/*/*kernel_stats.csv | head -1); cp $f $OUT/train_kernel_stats.csv; head -30 $f | cut -c1-150
