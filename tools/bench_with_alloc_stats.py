"""Development only: runs bench.py with the given arguments in this process and prints the caching allocator's counters
afterwards (device allocations / frees / retries): a steady-state step should make none."""
import runpy
import sys

import torch

sys.argv = ["bench.py"] + sys.argv[1:]
try:
    runpy.run_path(__file__.rsplit("/", 2)[0] + "/bench.py", run_name="__main__")
except SystemExit:
    pass
st = torch.cuda.memory_stats()
print({k: st[k] for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_ooms")},
      f"peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, reserved {torch.cuda.max_memory_reserved() / 2**30:.1f} GiB",
      file=sys.stderr)
