#!/usr/bin/env python
"""Condenses gpurun_out/prof_<tag>/ of tools/profile_f16x3.sh into profiles/r4_f16x3_kernel_stats.csv and profiles/r4_f16x3_pmc.json:
per conv kernel the MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE / 128: cycles per SIMD vs cycles per XCD) and the LDS
counters, with the hash of the kernel sources they were measured on.  Usage: python tools/summarize_f16x3_profile.py r4h [r4]"""
import collections
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402

tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "r4"      # profiles/<prefix>_f16x3_*
src = os.path.join(root, "gpurun_out", "prof_" + tag)


def newest(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(g, key=os.path.getmtime)


agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(newest("pmc_mfma/**/*counter_collection.csv"))):
    name = r["Kernel_Name"].split("(")[0]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[name] += r["Counter_Name"] == "GRBM_GUI_ACTIVE"
for r in csv.DictReader(open(newest("pmc_lds/**/*counter_collection.csv"))):
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for n, c in agg.items():
    if "conv" in n and c.get("GRBM_GUI_ACTIVE", 0) > 0:
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["GRBM_GUI_ACTIVE"]
        out[n] = {"dispatches": cnt[n], "GRBM_GUI_ACTIVE": c["GRBM_GUI_ACTIVE"], "SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"],
                  "mfma_busy_of_128_per_xcd_cycle": busy, "mfma_busy_frac": busy / 128.0,
                  "SQ_LDS_IDX_ACTIVE": c.get("SQ_LDS_IDX_ACTIVE"), "SQ_LDS_BANK_CONFLICT": c.get("SQ_LDS_BANK_CONFLICT"),
                  "SQ_WAIT_INST_LDS": c.get("SQ_WAIT_INST_LDS"), "SQ_WAVE_CYCLES": c.get("SQ_WAVE_CYCLES")}
json.dump({"command": "tools/profile_f16x3.sh: bench.py --steps 2 --warmup 1 --model frcnn101 --shape coco --augs FCDRG --precision f16x3 "
                      "(BASELINE configs[4], one GPU)", "csrc_sha1": bench.csrc_sha1(), "per_kernel": out},
          open(os.path.join(root, "profiles", prefix + "_f16x3_pmc.json"), "w"), indent=1)
for n, v in sorted(out.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:8]:
    print("%-55s n=%4d  MFMA busy %.1f / 128 = %.3f" % (n[:55], v["dispatches"], v["mfma_busy_of_128_per_xcd_cycle"], v["mfma_busy_frac"]))
rows = list(csv.DictReader(open(newest("stats/**/*kernel_stats.csv"))))
with open(os.path.join(root, "profiles", prefix + "_f16x3_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
