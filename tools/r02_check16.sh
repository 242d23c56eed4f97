#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c16; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "entropy or reeval or train or eam" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -30 $OUT/pytest.log
