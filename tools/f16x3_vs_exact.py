"""Characterises the opt-in f16x3 mode against the exact mode on the full BASELINE configs[1] pool (5 217 VOC-shaped
images, 3 augs): consistency differences, how many images moved by more than 1e-4 (borderline NMS / threshold decisions
that flip), and whether the selection (argsort -> first 1.2 * budget candidates -> cls_kldiv, budget 500) changes.
Usage: python tools/f16x3_vs_exact.py [pool_size] [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cald_amd import detector, synth, sweep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5217
out_path = sys.argv[2] if len(sys.argv) > 2 else None
augs = ["flip", "cut_out", "smaller_resize"]
sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
pool = [torch.from_numpy(im).cuda() for im in synth.make_pool(n, "voc", 0)]
res, secs = {}, {}
for prec in ("fp32", "f16x3"):
    m = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=600, max_size=1000, precision=prec).to("cuda")
    m.load_state_dict(sd); m.eval()
    sweep.sweep_device_images(m, pool[:64], list(range(64)), augs)
    torch.cuda.synchronize(); t = time.time()
    res[prec] = sweep.sweep_device_images(m, pool, list(range(n)), augs, bp=1.3, base_seed=0)
    torch.cuda.synchronize(); secs[prec] = time.time() - t
    del m
(ce, le), (ch, lh) = res["fp32"], res["f16x3"]
d = np.abs(ce - ch)
rs = np.random.RandomState(0)
labeled = [(None, [{"labels": torch.from_numpy(rs.randint(1, 21, rs.randint(1, 6)))}]) for _ in range(500)]
budget = min(500, n // 4)
pe = sweep.select(list(ce), [le[i] for i in range(n)], labeled, budget=budget, mr=1.2)
ph = sweep.select(list(ch), [lh[i] for i in range(n)], labeled, budget=budget, mr=1.2)
k = int(1.2 * budget)
out = {"pool": n, "images_per_s": {p: n / secs[p] for p in secs},
       "consistency_abs_diff": {"median": float(np.median(d)), "p99": float(np.quantile(d, 0.99)), "max": float(d.max()),
                                "images_over_1e-4": int((d > 1e-4).sum())},
       "cls_corr_entries_over_1e-4": int((np.abs(le - lh) > 1e-4).sum()), "cls_corr_entries": int(le.size),
       "first_%d_candidates_identical_order" % k: bool(np.array_equal(np.argsort(ce, kind="stable")[:k], np.argsort(ch, kind="stable")[:k])),
       "first_%d_candidates_same_set" % k: int(len(set(np.argsort(ce, kind="stable")[:k]) & set(np.argsort(ch, kind="stable")[:k]))),
       "selected_%d_same" % budget: int(len(set(map(int, pe)) & set(map(int, ph)))), "selected_identical_order": bool(list(map(int, pe)) == list(map(int, ph)))}
print(json.dumps(out))
if out_path:
    json.dump(out, open(out_path, "w"), indent=1)
