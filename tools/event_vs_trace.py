#!/usr/bin/env python
"""Same-run comparison of the two clocks the roofline uses: bench.py's HIP-event time per GEMM launch (CALD_PROFILE_DUMP csv)
against rocprofv3's kernel-trace duration of the very same launches.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- env CALD_PROFILE_DUMP=launches.csv python bench.py --steps 2 \
        --warmup 1 --no-cpu-baseline --no-full-pool --no-f16x3 --no-train --no-cfg4
    python tools/event_vs_trace.py launches.csv DIR out.json        (tools/profile_event_vs_trace.sh runs both on the GPU box)

The timed region is the last thing that launches GEMM kernels in that command, so the last len(csv) GEMM dispatches of the
trace are the csv's rows, in order (a grouped launch is one dispatch)."""
import csv
import glob
import json
import os
import re
import sys
from collections import OrderedDict


def main():
    launches, trace_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    ev = list(csv.DictReader(open(launches)))
    f = glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True)[0]
    tr = [r for r in csv.DictReader(open(f)) if any(t in r["Kernel_Name"] for t in ("conv_p4", "conv_mfma", "conv_stem", "conv_fused"))]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    # an event region of a grouped launch holds ONE dispatch when the grouped kernel took it, else one dispatch per problem;
    # walk backwards from the end of the trace: the last event row ends at the last GEMM dispatch
    pos = len(tr)
    spans = []
    for e in reversed(ev):
        m = re.search(r"group=(\d+)", e["desc"])
        w = 1
        if m and "group_kernel" not in tr[pos - 1]["Kernel_Name"]:
            w = int(m.group(1))
        spans.append((pos - w, pos)); pos -= w
    assert pos >= 0, "fewer GEMM dispatches in the trace than the event log needs"
    spans.reverse()
    agg = OrderedDict()
    for e, (lo, hi) in zip(ev, spans):
        t = tr[lo]
        key = re.sub(r"mt=\d+,", "", e["desc"]) + " | " + t["Kernel_Name"].split("(")[0].replace("void ", "")
        dur = sum((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) for x in tr[lo:hi]) * 1e-6
        a = agg.setdefault(key, dict(n=0, gflop=0.0, event_ms=0.0, trace_ms=0.0))
        a["n"] += 1; a["gflop"] += float(e["gflop"]); a["event_ms"] += float(e["ms"]); a["trace_ms"] += dur
    tot = dict(launches=len(ev), gflop=sum(a["gflop"] for a in agg.values()), event_ms=sum(a["event_ms"] for a in agg.values()),
               trace_ms=sum(a["trace_ms"] for a in agg.values()))
    tot["event_tflops"] = tot["gflop"] / tot["event_ms"]; tot["trace_tflops"] = tot["gflop"] / tot["trace_ms"]
    tot["event_frac"] = tot["event_tflops"] / 157.3; tot["trace_frac"] = tot["trace_tflops"] / 157.3
    rows = []
    for k, a in agg.items():
        rows.append(dict(layer=k, n=a["n"], event_ms=a["event_ms"] / a["n"], trace_ms=a["trace_ms"] / a["n"],
                         event_tflops=a["gflop"] / a["event_ms"], trace_tflops=a["gflop"] / a["trace_ms"],
                         gap_to_peak_ms_per_launch=(a["trace_ms"] - a["gflop"] / 157.3) / a["n"], share_of_trace_time=a["trace_ms"] / tot["trace_ms"]))
    rows.sort(key=lambda r: -r["gap_to_peak_ms_per_launch"] * r["n"])
    json.dump(dict(total=tot, per_layer=rows), open(out, "w"), indent=1)
    print(json.dumps(tot))
    for r in rows[:28]:
        print("%-95s n=%3d event %.3f trace %.3f ms  %.1f / %.1f TF  share %.3f" % (r["layer"][:95], r["n"], r["event_ms"], r["trace_ms"],
                                                                             r["event_tflops"], r["trace_tflops"], r["share_of_trace_time"]))


if __name__ == "__main__":
    main()
