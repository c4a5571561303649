#!/usr/bin/env python
"""Where a training step's wall-clock goes on the GPU: from a rocprofv3 --kernel-trace CSV of tools/bench_train.py, per steady-state step
(delimited by sgd_kernel): wall time, time with at least one kernel running, idle time, busy time per queue, and the largest idle gaps
with the kernels on either side.     python tools/train_trace_gaps.py <kernel_trace.csv> [out.json]"""
import csv
import json
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "0")) for r in rows))
sgd = [i for i, e in enumerate(ev) if e[2].startswith("sgd_kernel")]
steps = []
for a, b in zip(sgd[-11:-1], sgd[-10:]):
    seg = ev[a + 1:b + 1]
    t0, t1 = ev[a][1], ev[b][1]
    # union of busy intervals
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    last_name = ev[a][2]
    for s, e, n, q in seg:
        if cur_e is None:
            cur_s, cur_e = s, e
            if s > t0: gaps.append((s - t0, last_name, n))
        elif s <= cur_e:
            if e > cur_e: cur_e = e
        else:
            gaps.append((s - cur_e, last_name, n)); busy += cur_e - cur_s; cur_s, cur_e = s, e
        last_name = n if e >= (cur_e or 0) else last_name
    busy += cur_e - cur_s
    perq = defaultdict(int)
    for s, e, n, q in seg: perq[q] += e - s
    steps.append(dict(wall_ms=(t1 - t0) / 1e6, busy_ms=busy / 1e6, idle_ms=(t1 - t0 - busy) / 1e6, per_queue_ms={k: v / 1e6 for k, v in perq.items()},
                      kernels=len(seg), top_gaps=[dict(us=g / 1e3, after=a_, before=b_) for g, a_, b_ in sorted(gaps, reverse=True)[:12]]))
avg = {k: sum(s[k] for s in steps) / len(steps) for k in ("wall_ms", "busy_ms", "idle_ms")}
out = dict(steps=len(steps), average=avg, per_queue_ms_last=steps[-1]["per_queue_ms"], top_gaps_last_step=steps[-1]["top_gaps"])
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
