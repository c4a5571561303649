"""north_star parity against a path that does NOT share the arithmetic contract: the MI355X sweep (exact fp32 mode and the
opt-in f16x3 mode) vs the torch-CPU fp32 port's stored scores (tools/torch_cpu_reference.py -> profiles/torch_cpu_reference_r2.npz)
on the configs[1] synthetic pool.  Reports, per mode: |d consistency| median / p99 / max, images beyond 1e-4, overlap of the
candidate cut (first int(1.2 * budget) of argsort) and of the final selection (argsort + cls_kldiv), and whether the selected
ORDER is identical.  Also exact vs f16x3.  Usage: python tools/parity_full_pool.py [ref.npz] [out.json]"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

AUGS = os.environ.get("CALD_PARITY_AUGS", "flip,cut_out,smaller_resize").split(",")     # configs[0]: CALD_PARITY_AUGS=flip
FULL_POOL, FULL_BUDGET = 5217, 500


def compare(name_a, a, name_b, b, labeled, budget):
    from cald_amd import sweep
    (ca, ka), (cb, kb) = a, b
    d = np.abs(ca - cb)
    kq = int(1.2 * budget)
    cand_a, cand_b = np.argsort(ca)[:kq], np.argsort(cb)[:kq]
    sel_a = np.asarray(sweep.select(list(ca), [ka[i] for i in range(len(ca))], labeled, budget=budget, mr=1.2))
    sel_b = np.asarray(sweep.select(list(cb), [kb[i] for i in range(len(cb))], labeled, budget=budget, mr=1.2))
    dk = np.abs(ka - kb)
    same_prefix = 0
    for x, y in zip(sel_a, sel_b):
        if x != y:
            break
        same_prefix += 1
    return {"a": name_a, "b": name_b, "images": int(len(ca)), "budget": int(budget),
            "consistency_abs_diff": {"median": float(np.median(d)), "p99": float(np.quantile(d, 0.99)), "max": float(d.max()),
                                     "images_beyond_1e-4": int((d > 1e-4).sum()), "frac_beyond_1e-4": float((d > 1e-4).mean())},
            "cls_corr_abs_diff": {"median_nonzero": float(np.median(dk[(ka > 0) | (kb > 0)])) if ((ka > 0) | (kb > 0)).any() else 0.0,
                                  "entries_beyond_1e-4": int((dk > 1e-4).sum()), "entries": int(dk.size)},
            "candidates_first_%d_same_set" % kq: int(len(set(cand_a.tolist()) & set(cand_b.tolist()))),
            "selected_same_set": int(len(set(sel_a.tolist()) & set(sel_b.tolist()))), "selected_total": int(len(sel_a)),
            "selected_identical_order": bool(len(sel_a) == len(sel_b) and np.array_equal(sel_a, sel_b)),
            "selected_identical_prefix": int(same_prefix),
            "selected_sha1_a": hashlib.sha1(sel_a.astype(np.int64).tobytes()).hexdigest(),
            "selected_sha1_b": hashlib.sha1(sel_b.astype(np.int64).tobytes()).hexdigest()}


def main():
    ref_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join("profiles", "torch_cpu_reference_r2.npz")
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    z = np.load(ref_path)
    done = z["done"]
    assert [str(a) for a in z["augs"]] == AUGS, "reference file was made with augs %s (set CALD_PARITY_AUGS)" % list(z["augs"])
    n = int(np.argmin(done)) if not done.all() else len(done)       # the completed prefix
    assert n >= 64, "torch-CPU reference holds only %d images" % n
    cpu = (z["consistency"][:n].astype(np.float64), z["cls_corr"][:n].astype(np.float64))
    from cald_amd import detector, synth, sweep
    sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
    pool = [torch.from_numpy(im).cuda() for im in synth.make_pool(n, "voc", 0)]
    res = {}
    for prec in ("fp32", "f16x3"):
        m = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=600, max_size=1000, precision=prec).to("cuda")
        m.load_state_dict(sd); m.eval()
        sweep.sweep_device_images(m, pool[:64], list(range(64)), AUGS)
        torch.cuda.synchronize(); t = time.time()
        res[prec] = sweep.sweep_device_images(m, pool, list(range(n)), AUGS, bp=1.3, base_seed=0)
        torch.cuda.synchronize(); res[prec + "_s"] = time.time() - t
        del m
    rs = np.random.RandomState(0)
    labeled = [(None, [{"labels": torch.from_numpy(rs.randint(1, 21, rs.randint(1, 6)))}]) for _ in range(500)]
    budget = max(1, int(round(FULL_BUDGET * n / float(FULL_POOL))))
    out = {"pool": "configs[1] synthetic VOC-shaped pool (cald_amd/synth.py), positions 0..%d" % (n - 1), "augs": AUGS,
           "independent_path": "oracle/torch_port.py on host cores (torch %s, %d threads): oneDNN conv/linear, torch exp/softmax/interpolate"
                               % (str(z["torch_version"]), int(z["threads"])),
           "gpu_seconds": {"fp32": res["fp32_s"], "f16x3": res["f16x3_s"]},
           "fp32_vs_torch_cpu": compare("mi355x exact fp32", res["fp32"], "torch-CPU fp32", cpu, labeled, budget),
           "f16x3_vs_torch_cpu": compare("mi355x f16x3", res["f16x3"], "torch-CPU fp32", cpu, labeled, budget),
           "f16x3_vs_fp32": compare("mi355x f16x3", res["f16x3"], "mi355x exact fp32", res["fp32"], labeled, budget)}
    s = lambda c: {"images": c["images"], "images_beyond_1e-4": c["consistency_abs_diff"]["images_beyond_1e-4"],
                   "median_abs_diff": c["consistency_abs_diff"]["median"], "selected_same_set": c["selected_same_set"],
                   "selected_total": c["selected_total"], "selected_identical_order": c["selected_identical_order"]}
    out["summary"] = {"fp32_vs_torch_cpu": s(out["fp32_vs_torch_cpu"]), "f16x3_vs_torch_cpu": s(out["f16x3_vs_torch_cpu"]),
                      "f16x3_vs_fp32": s(out["f16x3_vs_fp32"]), "source": "tools/parity_full_pool.py"}
    import bench
    out["csrc_sha1"] = bench.csrc_sha1()          # bench.py reports this file only beside the kernels it was measured on
    print(json.dumps(out["summary"]))
    if out_path:
        json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
