"""Aggregates a rocprofv3 kernel_trace.csv by (kernel, grid) for the training step: which launches carry the time."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:48]
    if len(sys.argv) > 3 and sys.argv[3] not in name:
        continue
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
    a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
print("total %.2f ms/step" % (tot / 1e3 / steps))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-48s grid %8s x %4s wg %4s  calls/step %5.1f  us/call %8.1f  ms/step %6.3f" % (k[0], k[1], k[2], k[3], a[0] / steps, a[1] / a[0], a[1] / 1e3 / steps))
