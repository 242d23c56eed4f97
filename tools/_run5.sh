set -o pipefail
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "ea_ or evolution" > gpurun_out/pytest_ea.log 2>&1; echo "exit=$?" >> gpurun_out/pytest_ea.log; tail -30 gpurun_out/pytest_ea.log
