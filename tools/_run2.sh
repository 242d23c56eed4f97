set -o pipefail
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail > $R/gpurun_out/pmc/avail.txt 2>&1 || true
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INST_CYCLES_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $R/gpurun_out/pmc/$tag.log 2>&1 || echo "FAILED $tag"
done
ls -R $R/gpurun_out/pmc | head -40
