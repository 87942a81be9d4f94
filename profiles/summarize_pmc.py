#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, as MI355X_MICROARCH.md's HBM
section prescribes) into per-kernel HBM bytes per launch.

usage: summarize_pmc.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [circuit batch window_bits]
Correction applied (same guide, gfx950): the counters are in KB, and FETCH_SIZE counts half of the wide reads, so
HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; the NTT passes, which read and write the same volume, confirm it.
"""
import csv
import glob
import json
import os
import re
import sys


def per_kernel(d, counter):
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = {}
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            key = (r["Dispatch_Id"], r["Kernel_Name"])
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r["Counter_Value"])
        for (_, name), v in per_dispatch.items():
            name = re.sub(r"\(.*", "", name).strip()
            a = acc.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += v
    return {k: {"dispatches": n, "avg_kb": round(s / n, 1)} for k, (n, s) in acc.items()}


def main():
    fdir, wdir, out = sys.argv[1:4]
    f, w = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fe, wr = f.get(k, {"dispatches": 0, "avg_kb": 0.0}), w.get(k, {"dispatches": 0, "avg_kb": 0.0})
        kernels[k] = {"FETCH_SIZE": fe, "WRITE_SIZE": wr,
                      "hbm_bytes_per_launch_corrected": int((2 * fe["avg_kb"] + wr["avg_kb"]) * 1024)}
    doc = {"units": "KB per dispatch as reported; corrected HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024", "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    if len(sys.argv) >= 7 and os.environ.get("SPP_WRITE_LATEST"):   # pmc_hbm_latest.json: what bench.py reads for roofline.traffic
        import hashlib
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        lib = os.path.join(root, "shielded-pool-pinocchio-solana_amd", "libspp.so")
        pick = lambda kern, field: [v for k, v in kernels.items() if kern in k and field in k]
        g1, g2 = pick("k_msm_flat", "FqParams"), pick("k_msm_flat", "Fq2")
        if g1:
            latest = {"circuit": sys.argv[4], "batch": int(sys.argv[5]), "n_distinct_witnesses": int(sys.argv[5]), "window_bits": int(sys.argv[6]),
                      "libspp_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
                      "k_msm_flat_g1_hbm_bytes_per_launch": g1[0]["hbm_bytes_per_launch_corrected"],
                      "k_msm_flat_g2_hbm_bytes_per_launch": g2[0]["hbm_bytes_per_launch_corrected"] if g2 else None,
                      "source": "profiles/%s (separate FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes)" % os.path.basename(out)}
            valu = os.environ.get("SPP_VALU_JSON")
            if valu and os.path.exists(valu):
                vd = json.load(open(valu))
                for k, v in vd.items():
                    if "k_msm_flat" in k and "FqParams" in k:
                        latest["k_msm_flat_g1_valu_issue_util_serialised"] = v.get("valu_issue_util")
                        latest["k_msm_flat_g1_resident_waves_per_simd"] = v.get("mean_resident_waves_per_simd")
                    if "k_msm_flat" in k and "Fq2" in k:
                        latest["k_msm_flat_g2_valu_issue_util_serialised"] = v.get("valu_issue_util")
                latest["source"] += " and profiles/%s (SQ counters, SPP_SERIAL=1)" % os.path.basename(valu)
            json.dump(latest, open(os.path.join(os.path.dirname(out) or ".", "pmc_hbm_latest.json"), "w"), indent=1)
            print(latest)


if __name__ == "__main__":
    main()
