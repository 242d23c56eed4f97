#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_next.py -x -q -m gpu -k "ea_ or evolution or eam" > gpurun_out/ea_tests.log 2>&1; echo "rc=$?"
tail -15 gpurun_out/ea_tests.log
