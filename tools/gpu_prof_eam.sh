#!/bin/bash
set -o pipefail
OUT=gpurun_out/eam
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/tools/prof_eam_step.py > $R/$OUT/stats.log 2>&1
cd $R
grep -v amdgpu.ids $OUT/stats.log | tail -2
f=$(ls $OUT/stats/*/*kernel_stats.csv | head -1); cp $f $OUT/eam_kernel_stats.csv
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/eam/stats/*/*kernel_stats.csv')[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print(f"kernel time total {tot/1e6:.1f} ms over 10 steps = {tot/1e7:.2f} ms per step, {calls/10:.0f} launches per step")
for r in rows[:28]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e7:8.3f} ms/step", f"{float(r['AverageNs'])/1e3:8.1f} us")
PY
