"""BASELINE.json configs[1] end to end at full size: Faster R-CNN R50-FPN, VOC2012-shaped synthetic pool of 5 217 images
(cald_train.py:299-300: 5 717 train images - 500 initially labeled), augs flip / cut_out / smaller_resize, then the
selection stage (argsort + cls_kldiv, budget 500, mr 1.2).  Prints / writes wall times and a spot check of 6 random
pool positions against the oracle (bit for bit).  Usage: python tools/run_config2.py [pool_size] [out.json]"""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cald_amd import detector, synth, sweep
from oracle import oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5217
out_path = sys.argv[2] if len(sys.argv) > 2 else None
augs = ["flip", "cut_out", "smaller_resize"]
sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
model = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=600, max_size=1000).to("cuda")
model.load_state_dict(sd)
t = time.time(); pool = synth.make_pool(n, "voc", 0); t_gen = time.time() - t
t = time.time(); dev = [torch.from_numpy(im).cuda() for im in pool]; torch.cuda.synchronize(); t_up = time.time() - t
sweep.sweep_device_images(model, dev[:64], list(range(64)), augs)          # warm-up (arena, code objects)
torch.cuda.synchronize(); t = time.time()
cons, cls = sweep.sweep_device_images(model, dev, list(range(n)), augs, bp=1.3, base_seed=0)
torch.cuda.synchronize(); t_sweep = time.time() - t
rs = np.random.RandomState(0)
labeled = [(None, [{"labels": torch.from_numpy(rs.randint(1, 21, rs.randint(1, 6)))}]) for _ in range(500)]
t = time.time(); picked = sweep.select(list(cons), [cls[i] for i in range(n)], labeled, budget=min(500, n // 4), mr=1.2); t_sel = time.time() - t
P = orc.prepare_frcnn(sd, 21, 50)
spots = sorted(int(i) for i in rs.choice(n, size=min(6, n), replace=False))
wc, wk = orc.get_uncertainty(P, [pool[i] for i in spots], augs, 21, 1.3, 600, 1000, 0, positions=spots)
ok = all(cons[i] == wc[j] and np.array_equal(cls[i], wk[j]) for j, i in enumerate(spots))
srt = np.sort(cons); kq = int(1.2 * min(500, n // 4))
gaps = np.diff(srt[:kq + 1])
res = {"config": "BASELINE.json configs[1] full pool", "pool": n, "sweep_s": t_sweep, "images_per_s": n / t_sweep,
       "selection_s": t_sel, "images_per_s_incl_selection": n / (t_sweep + t_sel), "upload_s": t_up, "synth_gen_s": t_gen,
       "zero_score_images": int((cons == 0).sum()), "consistency_min_max": [float(cons.min()), float(cons.max())],
       "selected_sha1": hashlib.sha1(np.asarray(picked, np.int64).tobytes()).hexdigest(), "n_selected": int(len(picked)),
       "rank_gap_first_%d_candidates" % kq: {"min": float(gaps.min()), "median": float(np.median(gaps)), "exact_ties": int((gaps == 0).sum()),
                                              "gap_at_cut": float(srt[kq] - srt[kq - 1])},
       "oracle_spot_check_positions": spots, "oracle_spot_check_bit_exact": bool(ok)}
print(json.dumps(res))
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
