"""Per-workgroup timeline of one conv launch on the chip (ConvArgs::trace, conv_p4.hip): where a workgroup's life goes -- prologue,
k-loop, epilogue, store drain -- and how busy each CU's three slots are.  Usage: python tools/conv_trace.py [layer substring ...]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cald_amd import _ffi, detector
from bench_conv import LAYERS


def analyse(path, name):
    t = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
    t = t[t[:, 0] != 0]
    t0, t1, t2, t3, t4 = [t[:, i].astype(np.int64) for i in range(5)]
    hw, xcc = t[:, 5].astype(np.int64), t[:, 6].astype(np.int64) & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    out = {"layer": name, "workgroups": int(len(t)), "distinct_cu_keys": int(len(np.unique(key)))}
    ph = {"prologue": t1 - t0, "kloop": t2 - t1, "epilogue_issue": t3 - t2, "store_drain": t4 - t3, "life": t4 - t0}
    for k, v in ph.items():
        out[k] = {"p10": float(np.percentile(v, 10)), "median": float(np.median(v)), "p90": float(np.percentile(v, 90)), "mean": float(v.mean())}
    # per CU: span, time with 0 / 1 / 2 / 3 workgroups inside their k-loop
    spans, kfrac, conc, res_conc = [], [], np.zeros(5), np.zeros(5)
    for k in np.unique(key):
        m = key == k
        a0, a4, k1, k2 = t0[m], t4[m], t1[m], t2[m]
        lo, hi = a0.min(), a4.max()
        spans.append(hi - lo)
        for (b, e, tgt) in ((k1, k2, conc), (a0, a4, res_conc)):
            ev = sorted([(int(x), 1) for x in b] + [(int(x), -1) for x in e])
            cur, last = 0, lo
            for x, d in ev:
                tgt[min(cur, 4)] += x - last; last = x; cur += d
            tgt[min(cur, 4)] += hi - last
        kfrac.append((k2 - k1).sum() / float(hi - lo))
    out["cu_span_cycles_median"] = float(np.median(spans))
    out["kloop_concurrency_share"] = [round(float(x), 4) for x in conc / conc.sum()]       # CU time with 0, 1, 2, 3, >= 4 workgroups inside the k-loop
    out["resident_concurrency_share"] = [round(float(x), 4) for x in res_conc / res_conc.sum()]  # ... with 0, 1, 2, 3, >= 4 workgroups resident
    out["kloop_sum_over_span_median"] = float(np.median(kfrac))
    out["launch_span_cycles"] = int(t4.max() - t0.min())
    return out


def main():
    want = sys.argv[1:] or ["fc7", "fc6", "256->1024", "128->512", "1024->256"]
    L, ctx = _ffi.lib(), detector.get_ctx(0)
    os.makedirs("gpurun_out", exist_ok=True)
    res = []
    for (name, V, H, W, Cin, Cout, K, s, p, resid, relu, grp) in LAYERS:
        if not any(w in name for w in want):
            continue
        path = "/tmp/conv_trace.bin"
        os.environ["CALD_CONV_TRACE"] = path
        ms, tf = C.c_double(), C.c_double()
        _ffi.check(L.cald_op_conv_bench(ctx, V, H, W, Cin, Cout, K, s, p, resid, relu, 3, grp, C.byref(ms), C.byref(tf)))
        r = analyse(path, name); r["ms"] = ms.value; r["tflops"] = tf.value
        res.append(r)
        print(json.dumps(r), flush=True)
        os.system("cp /tmp/conv_trace.bin gpurun_out/conv_trace_%s.bin" % name.split()[0].replace("/", "_")) if name.startswith("fc7") else None
    json.dump(res, open("gpurun_out/conv_trace_%s.json" % os.environ.get("TRACE_TAG", "run"), "w"), indent=1)


if __name__ == "__main__":
    main()
