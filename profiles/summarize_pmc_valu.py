#!/usr/bin/env python3
"""Summarise one rocprofv3 --pmc pass of SQ/GRBM counters into per-kernel VALU issue utilisation.

usage: summarize_pmc_valu.py <rocprofv3 output dir> <out.json>
SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over waves (MI355X_MICROARCH.md, PMC notes);
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  valu_issue_util = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).
"""
import csv
import glob
import json
import os
import re
import sys


def main():
    d, out = sys.argv[1:3]
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(path)):
            key = (r["Dispatch_Id"], re.sub(r"\(.*", "", r["Kernel_Name"]).strip(), r["Counter_Name"])
            per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
        for (_, name, ctr), v in per.items():
            a = acc.setdefault(name, {}).setdefault(ctr, [0, 0.0])
            a[0] += 1
            a[1] += v
    doc = {}
    for name, ctrs in sorted(acc.items()):
        avg = {c: s / n for c, (n, s) in ctrs.items()}
        row = {"dispatches": max(n for n, _ in ctrs.values()), "avg": {c: round(v, 1) for c, v in avg.items()}}
        if avg.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in avg:
            cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
            row["kernel_cycles"] = round(cyc)
            row["valu_issue_util"] = round(4.0 * avg["SQ_ACTIVE_INST_VALU"] / (cyc * 1024), 4)
            if "SQ_INSTS_VALU" in avg:
                row["cycles_per_valu_wave_instruction"] = round(4.0 * avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_INSTS_VALU"], 3) if avg.get("SQ_INSTS_VALU") else None
            if "SQ_WAVE_CYCLES" in avg:
                row["mean_resident_waves_per_simd"] = round(4.0 * avg["SQ_WAVE_CYCLES"] / (cyc * 1024), 3)
        doc[name] = row
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in doc.items():
        if "k_msm_flat" in k or "k_ntt" in k or "k_spmv" in k:
            print(k[:60], {x: v.get(x) for x in ("kernel_cycles", "valu_issue_util", "cycles_per_valu_wave_instruction", "mean_resident_waves_per_simd")})


if __name__ == "__main__":
    main()
