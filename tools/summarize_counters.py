#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc CSVs of tools/collect_mfma_counters.sh into one JSON: per kernel family the mean counter values
per launch, the MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES; both count cycles summed over the CUs) and
the achieved clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration, MI355X_MICROARCH.md 'DVFS give-back').

    python3 tools/summarize_counters.py gpurun_out/<tag>  > profiles/<tag>_counters.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

FAMILIES = ("k_encoder_fused", "k_rollout_ms_mfma", "k_rollout_resident", "k_rollout_stream", "k_reeval_bwd_logits",
            "k_reeval_bwd_glimpse", "k_reeval_bwd_gather", "k_linear_mfma", "k_linear_wgrad", "k_mha_encoder_bwd", "k_mha_encoder")


def family(name):
    for f in FAMILIES:
        if f + "<" in name or f + "(" in name or name.startswith(f):
            return f
    return None


def main(root):
    out = {"_how": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU "
                   "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --workload W --steps 3 "
                   "--warmup 1 --no-cpu-baseline [--no-graph] (tools/collect_mfma_counters.sh); means per launch; durations "
                   "from the kernel trace of the same pass; clock = GRBM_GUI_ACTIVE / 8 / duration"}
    for d in sorted(glob.glob(os.path.join(root, "pmc_mfma_*"))):
        wl = os.path.basename(d)[len("pmc_mfma_"):]
        cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if not cc:
            continue
        dur = {}
        for path in kt:
            with open(path) as f:
                for r in csv.DictReader(f):
                    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        acc = defaultdict(lambda: defaultdict(float))
        n = defaultdict(set)
        tdur = defaultdict(float)
        with open(cc[0]) as f:
            for r in csv.DictReader(f):
                fam = family(r["Kernel_Name"])
                if fam is None:
                    continue
                acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in n[fam]:
                    n[fam].add(r["Dispatch_Id"])
                    tdur[fam] += dur.get(r["Dispatch_Id"], 0.0)
        res = {}
        for fam, c in acc.items():
            k = len(n[fam])
            m = {name: v / k for name, v in c.items()}
            e = {"launches": k, "mean_duration_ms": round(tdur[fam] / k * 1e3, 4), "counters_per_launch": {a: round(b, 1) for a, b in m.items()}}
            if m.get("SQ_BUSY_CU_CYCLES"):
                e["mfma_busy_frac_of_cu_busy"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / m["SQ_BUSY_CU_CYCLES"], 4)
            if m.get("GRBM_GUI_ACTIVE") and tdur[fam]:
                e["clock_ghz"] = round(m["GRBM_GUI_ACTIVE"] / 8 / (tdur[fam] / k) / 1e9, 3)
            if m.get("SQ_WAVE_CYCLES"):
                e["valu_active_frac_of_wave_cycles"] = round(m.get("SQ_ACTIVE_INST_VALU", 0.0) / m["SQ_WAVE_CYCLES"], 4)
            res[fam] = e
        out[wl] = res
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
