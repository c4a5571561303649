#!/usr/bin/env python
"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 CSVs) into profiles/<tag>_*.{csv,json} (tracked)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def find(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)      # several runs may have merged into one directory: the newest wins
    return max(g, key=os.path.getmtime) if g else None


stats = find("stats/**/*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    print("kernel stats ->", os.path.join(dst, "%s_kernel_stats.csv" % tag))

summary = {}
for key, pat in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write"), ("MFMA", "pmc_mfma")):
    f = find(pat + "/**/*counter_collection.csv")
    if not f:
        continue
    agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name][r["Counter_Name"]] += 1
    for name in agg:
        for c in agg[name]:
            summary.setdefault(name, {})[c] = {"sum": agg[name][c], "dispatches": cnt[name][c], "avg": agg[name][c] / cnt[name][c]}
def is_gemm(name):          # the exact mode's GEMM family; the split-fp16 look-ahead of rpn_prune.hip (conv_h3) is summarised apart
    return any(t in name for t in ("conv_mfma", "conv_p4", "conv_fused", "conv_stem"))


def is_lookahead(name):
    return any(t in name for t in ("conv_h3", "conv_h4"))


gemm_ns = gemm_calls = 0
if stats:
    for r in rows:
        if is_gemm(r["Name"]):
            gemm_ns += int(r["TotalDurationNs"]); gemm_calls += int(r["Calls"])
if summary:
    conv = {k: v for k, v in summary.items() if is_gemm(k)}
    tot_f = sum(v.get("FETCH_SIZE", {}).get("sum", 0) for v in conv.values())
    tot_w = sum(v.get("WRITE_SIZE", {}).get("sum", 0) for v in conv.values())
    n = sum(v.get("FETCH_SIZE", {}).get("dispatches", 0) for v in conv.values())
    mf = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("sum", 0) for v in conv.values())
    ga = sum(v.get("GRBM_GUI_ACTIVE", {}).get("sum", 0) for v in conv.values())
    bytes_per_launch = ((2.0 * tot_f + tot_w) * 1024.0 / n) if n else None
    avg_ns = gemm_ns / gemm_calls if gemm_calls else None
    sys.path.insert(0, root)
    import bench
    out = {"csrc_sha1": bench.csrc_sha1(),     # the kernel sources these counters were measured on: bench.py reports them only for this tree
           "per_kernel": summary,
           "conv_mfma": {"dispatches": n, "FETCH_SIZE_KB_sum": tot_f, "WRITE_SIZE_KB_sum": tot_w,
                         # MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads -> x2; unit KB
                         "hbm_bytes_per_launch": bytes_per_launch,
                         # SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, GRBM_GUI_ACTIVE per XCD: 32 CUs x 4 SIMDs = 128
                         "mfma_busy": (mf / ga / 128.0) if ga else None,
                         # counter bytes per launch / the --stats run's average launch duration of the same kernels
                         "avg_launch_ns": avg_ns,
                         "hbm_gbps": (bytes_per_launch / avg_ns) if (bytes_per_launch and avg_ns) else None}}
    look = {k: v for k, v in summary.items() if is_lookahead(k)}
    if look:
        lf = sum(v.get("FETCH_SIZE", {}).get("sum", 0) for v in look.values()); lw = sum(v.get("WRITE_SIZE", {}).get("sum", 0) for v in look.values())
        ln = sum(v.get("FETCH_SIZE", {}).get("dispatches", 0) for v in look.values())
        lm = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("sum", 0) for v in look.values()); lg = sum(v.get("GRBM_GUI_ACTIVE", {}).get("sum", 0) for v in look.values())
        out["rpn_prune_lookahead"] = {"dispatches": ln, "hbm_bytes_per_launch": ((2.0 * lf + lw) * 1024.0 / ln) if ln else None,
                                      "mfma_busy": (lm / lg / 128.0) if lg else None}
    json.dump(out, open(os.path.join(dst, "%s_pmc.json" % tag), "w"), indent=1)
    print("pmc ->", os.path.join(dst, "%s_pmc.json" % tag), out["conv_mfma"])
