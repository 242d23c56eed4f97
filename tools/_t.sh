set -o pipefail
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "attention or encoder or boundaries" 2>&1 | tail -5
timeout -k 10 400 python bench.py --workload cvrp500 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-420
