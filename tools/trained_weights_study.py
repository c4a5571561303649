"""trained_weights_study.py -- re-measure the round-5 distributional claims on a detector this repo TRAINED ITSELF.

Every figure DESIGN.md quotes about the pool -- the fraction of P2 / P3 the certified pruning recomputes, the worst bound ratio, the f16x3
flip rate, "the whole pool lies in [0.001, 0.055] so any flip can cross the cut" (section 6b) -- came from synth.pseudo_trained_frcnn: random
heads on a random body.  This script trains the Faster R-CNN of configs[1] with the repo's own training step (cald_amd/train.py, the
reference's cald_train.py:40-74) on a synthetic LABELLED set -- the shapes synth.synth_image draws, class = shape (rectangle / ellipse) x
colour bucket (10) = 20 classes, VOC-sized images -- and repeats the measurements on those weights:

    python tools/trained_weights_study.py --steps 2000 --pool 1024 --out profiles/r6_trained_weights.json

Reference lines: cald_train.py:40-74 (train_one_epoch), :349-356 (the cycle's model), :439-447 (selection).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cald_amd import _ffi, detector, synth, sweep, train


def labelled_image(index, H, W):
    """synth.synth_image(index, H, W) and its ground truth: one box per drawn shape that is still >= 40 % visible, label = 1 + 10 * shape
    + colour bucket (hue sextant of the fill colour x bright / dark; grey-ish fills share bucket 9 with the dark reds... 10 buckets)."""
    img = synth.synth_image(index, H, W)
    rs = np.random.RandomState(1000003 + index)
    gh, gw = H // 32 + 2, W // 32 + 2
    rs.rand(gh, gw, 3)
    yy, xx = np.mgrid[0:H, 0:W]
    shapes = []
    owner = np.full((H, W), -1, np.int32)
    for s in range(rs.randint(3, 9)):
        cy, cx = rs.rand() * H, rs.rand() * W
        hh, ww = (0.05 + 0.3 * rs.rand()) * H, (0.05 + 0.3 * rs.rand()) * W
        col = rs.rand(3).astype(np.float32)
        rect = rs.rand() < 0.5
        if rect:
            mask = (np.abs(yy - cy) < hh / 2) & (np.abs(xx - cx) < ww / 2)
        else:
            mask = ((yy - cy) / (hh / 2)) ** 2 + ((xx - cx) / (ww / 2)) ** 2 < 1.0
        owner[mask] = s
        shapes.append((mask.sum(), rect, col))
    boxes, labels = [], []
    for s, (area, rect, col) in enumerate(shapes):
        vis = owner == s
        if area == 0 or vis.sum() < 0.4 * area or vis.sum() < 64:
            continue
        ys, xs = np.nonzero(vis)
        mx, mn = float(col.max()), float(col.min())
        if mx - mn < 0.15:
            bucket = 9
        else:
            hue = int(np.argmax(col)) * 2 + int(col[(int(np.argmax(col)) + 1) % 3] > col[(int(np.argmax(col)) + 2) % 3])      # 0..5
            bucket = hue if mx > 0.5 else 6 + hue % 3
        boxes.append([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1]); labels.append(1 + 10 * (0 if rect else 1) + bucket)
    return img, np.array(boxes, np.float32).reshape(-1, 4), np.array(labels, np.int64)


def train_detector(steps, n_train, batch, seed=0, log=print):
    from torch.utils.data.sampler import SequentialSampler
    from cald_amd.group_by_aspect_ratio import GroupedBatchSampler, _quantize
    sd0 = synth.pseudo_trained_frcnn(21, 50, seed=seed)
    net = train.FasterRCNNTrainer(sd0, 21, depth=50, min_size=600, max_size=1000, generator=torch.Generator().manual_seed(seed))
    model = train.TrainableDetector(net)
    opt = train.SGD([p for p in model.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=1e-4, net=net)
    sizes = synth.pool_sizes(n_train, "voc", 100 + seed)
    t = time.time()
    data = [labelled_image(100000 + i, h, w) for i, (h, w) in enumerate(sizes)]
    log("labelled set: %d images, %.1f boxes / image, %.1f s to draw" % (n_train, np.mean([len(d[2]) for d in data]), time.time() - t))
    groups = _quantize([float(w) / float(h) for h, w in sizes], (2 ** np.linspace(-1, 1, 7)).tolist())
    dev = [(torch.from_numpy(im).cuda(), {"boxes": torch.from_numpy(b), "labels": torch.from_numpy(l)}) for im, b, l in data]
    hist, step, t0 = [], 0, time.time()
    warm = 200
    while step < steps:
        for idx in GroupedBatchSampler(SequentialSampler(sizes), groups, batch):
            if step >= steps: break
            lr = 0.01 * min(1.0, (step + 1) / warm) * (0.1 if step > 0.8 * steps else 1.0)            # linear warm-up (cald_train.py:45-50), one decay
            opt.param_groups[0]["lr"] = lr
            ims = [dev[i][0] for i in idx]; tgs = [dev[i][1] for i in idx]
            loss = model(ims, tgs); total = sum(loss.values())
            opt.zero_grad(); total.backward(); opt.step()
            if step % 100 == 0 or step == steps - 1:
                v = {k: float(x.detach()) for k, x in loss.items()}
                if not np.isfinite(sum(v.values())): raise RuntimeError("loss is not finite at step %d: %r" % (step, v))      # cald_train.py:62-65
                hist.append(dict(step=step, lr=lr, **v)); log("step %4d lr %.4f  %s" % (step, lr, "  ".join("%s %.4f" % kv for kv in v.items())))
            step += 1
    torch.cuda.synchronize()
    log("trained %d steps of batch %d in %.1f s" % (steps, batch, time.time() - t0))
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    return sd, hist, data


def detection_quality(model, data, n=64):
    """fraction of ground-truth boxes found (IoU >= 0.5, right label) and detections per image on the first n training images"""
    hit = tot = ndet = 0
    for im, b, l in data[:n]:
        out = model.forward_views([(torch.from_numpy(im).cuda(), False, None)])[0]
        db, dl, ds = out["boxes"].cpu().numpy(), out["labels"].cpu().numpy(), out["scores"].cpu().numpy()
        keep = ds >= 0.5; db, dl = db[keep], dl[keep]; ndet += len(db)
        for g, gl in zip(b, l):
            tot += 1
            if len(db) == 0: continue
            ix0 = np.maximum(db[:, 0], g[0]); iy0 = np.maximum(db[:, 1], g[1]); ix1 = np.minimum(db[:, 2], g[2]); iy1 = np.minimum(db[:, 3], g[3])
            inter = np.clip(ix1 - ix0, 0, None) * np.clip(iy1 - iy0, 0, None)
            iou = inter / ((db[:, 2] - db[:, 0]) * (db[:, 3] - db[:, 1]) + (g[2] - g[0]) * (g[3] - g[1]) - inter)
            hit += bool(((iou >= 0.5) & (dl == gl)).any())
    return dict(images=n, recall_at_iou50_score50=hit / max(tot, 1), detections_per_image_score50=ndet / n)


def p2_coverage(model, imgs, n=32):
    """could the FPN output conv of P2 (17 % of the GEMM time) be deferred like the RPN head?  Fraction of P2 pixels that
    (a) a RoIAlign sample of a level-2 proposal reads (bilinear footprint of the 14 x 14 sample grid), (b) lie in the 3 x 3 neighbourhood
    of a pixel whose RPN head the pruning recomputes exactly, (c) either."""
    fa = fb = fu = 0.0
    model.set_rpn_prune_capture(True)
    try:
        for im in imgs[:n]:
            model.forward_views([(im, False, None)])
            sel = model.debug_tensor("rpn0", 0)[:, :, 0] != -np.finfo(np.float32).max
            H, W = sel.shape
            props = model.debug_tensor("proposals", 0).reshape(-1, 4)
            props = props[(props[:, 2] > props[:, 0]) | (props[:, 3] > props[:, 1])]
            area = (props[:, 2] - props[:, 0]) * (props[:, 3] - props[:, 1])
            lvl = np.clip(np.floor(4 + np.log2(np.sqrt(np.maximum(area, 1e-12)) / 224.0) + 1e-6), 2, 5)
            need = np.zeros((H, W), bool)
            for x0, y0, x1, y1 in props[lvl == 2] * 0.25:
                rw, rh = max(x1 - x0, 1.0), max(y1 - y0, 1.0)
                sx = x0 + (np.arange(14) + 0.5) * rw / 14; sy = y0 + (np.arange(14) + 0.5) * rh / 14
                xs = np.unique(np.clip(np.concatenate([np.floor(sx), np.floor(sx) + 1]), 0, W - 1).astype(int))
                ys = np.unique(np.clip(np.concatenate([np.floor(sy), np.floor(sy) + 1]), 0, H - 1).astype(int))
                need[np.ix_(ys, xs)] = True
            dil = np.zeros_like(sel)
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    dil[max(0, dy):H + min(0, dy), max(0, dx):W + min(0, dx)] |= sel[max(0, -dy):H + min(0, -dy), max(0, -dx):W + min(0, -dx)]
            fa += need.mean(); fb += dil.mean(); fu += (need | dil).mean()
    finally:
        model.set_rpn_prune_capture(False)
    k = min(n, len(imgs))
    return dict(images=k, roi_align_level2_footprint=fa / k, rpn_selected_3x3=fb / k, union=fu / k)


def pool_study(sd, tag, n_pool, log=print):
    import ctypes as C
    L, ctx = _ffi.lib(), detector.get_ctx(0)
    augs = ["flip", "cut_out", "smaller_resize"]
    pool = [torch.from_numpy(im).cuda() for im in synth.make_pool(n_pool, "voc", 0)]
    pos = list(range(n_pool))
    out = {}
    m = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=600, max_size=1000).to("cuda"); m.load_state_dict(sd); m.eval()
    _ffi.check(L.cald_profile_enable(ctx, 1))
    t = time.time(); ce, ke, me = sweep.sweep_device_images(m, pool, pos, augs, bp=1.3, base_seed=0, batch_images=64, margins=False) + (None,)
    torch.cuda.synchronize(); dt = time.time() - t
    ms, fl, worst = C.c_double(), C.c_double(), C.c_double(); frac = (C.c_double * 2)()
    _ffi.check(L.cald_profile_prune(ctx, C.byref(ms), C.byref(fl), frac, C.byref(worst), None))
    _ffi.check(L.cald_profile_enable(ctx, 0))
    nfb = C.c_int64(); _ffi.check(L.cald_profile_prune_fallbacks(ctx, C.byref(nfb)))
    m.set_rpn_prune(False)
    c0, k0 = sweep.sweep_device_images(m, pool, pos, augs, bp=1.3, base_seed=0, batch_images=64)
    m.set_rpn_prune(True)
    out["rpn_prune"] = dict(pixels_recomputed_exactly=dict(P2=frac[0], P3=frac[1]), worst_observed_error_over_bound=worst.value,
                            bit_identical_to_dense_head=bool(ce.tobytes() == c0.tobytes() and ke.tobytes() == k0.tobytes()),
                            dense_fallbacks_so_far=int(nfb.value), images_per_s_incl_profile_events=n_pool / dt)
    out["p2_coverage"] = p2_coverage(m, pool)
    cm, km, mg = sweep.sweep_device_images(m, pool, pos, augs, bp=1.3, base_seed=0, batch_images=64, margins=True)
    del m; torch.cuda.empty_cache()
    mf = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=600, max_size=1000, precision="f16x3").to("cuda"); mf.load_state_dict(sd); mf.eval()
    ch, kh, mh = sweep.sweep_device_images(mf, pool, pos, augs, bp=1.3, base_seed=0, batch_images=64, margins=True)
    del mf; torch.cuda.empty_cache()
    d = np.abs(ce - ch); dk = np.abs(ke - kh).max(axis=1)
    budget = max(10, n_pool // 10)
    rs = np.random.RandomState(0)
    labeled = [(None, [{"labels": torch.from_numpy(rs.randint(1, 21, rs.randint(1, 6)))}]) for _ in range(100)]
    se = sweep.select(list(ce), [ke[i] for i in range(n_pool)], labeled, budget=budget, mr=1.2)
    sh = sweep.select(list(ch), [kh[i] for i in range(n_pool)], labeled, budget=budget, mr=1.2)
    srt = np.sort(ce); cut = srt[int(1.2 * budget) - 1]
    gaps = np.diff(srt[max(0, int(1.2 * budget) - 25):int(1.2 * budget) + 25])
    changed = (d > 1e-5) | (dk > 1e-5)
    noise = np.array(_ffi.MARGIN_NOISE_F16X3, np.float32)[None, :15]
    flagged = (mh[:, :15] < 2.0 * noise).any(axis=1)
    out["scores"] = dict(images=n_pool, consistency_quantiles={q: float(np.quantile(ce, float(q))) for q in ("0", "0.01", "0.1", "0.5", "0.9", "0.99", "1")},
                         candidate_cut=float(cut), budget=budget, median_gap_between_neighbours_at_the_cut=float(np.median(gaps)),
                         zero_score_images=int((ce == 0).sum()), mean_detections_proxy_cls_corr_nonzero=float((ke > 0).sum(axis=1).mean()))
    moved = d[d > 1e-5]
    out["f16x3_vs_exact"] = dict(median_abs_consistency_diff=float(np.median(d)), images_beyond_1e_5=int((d > 1e-5).sum()), images_beyond_1e_4=int((d > 1e-4).sum()),
                                 max_abs_consistency_diff=float(d.max()), cls_corr_changed_beyond_1e_5=int((dk > 1e-5).sum()),
                                 selected_in_common=len(set(map(int, se)) & set(map(int, sh))), selected_total=len(se),
                                 a_flip_moves_an_image_by={"min": float(moved.min()) if len(moved) else None, "median": float(np.median(moved)) if len(moved) else None, "max": float(moved.max()) if len(moved) else None},
                                 note="f16x3 is itself bit-identical to its CPU restatement since round 6 (tests: test_sweep_f16x3_matches_oracle); these are the decisions that flip between the two arithmetics")
    out["cascade"] = dict(images_changed=int(changed.sum()), flagged_by_the_margin_audit_at_2x_noise=int(flagged.sum()), changed_and_flagged=int((changed & flagged).sum()),
                          fraction_flagged=float(flagged.mean()),
                          changed_images_whose_move_can_cross_the_cut=int(((np.abs(ce - cut) < d) & changed).sum()),
                          images_within_a_typical_flip_of_the_cut=int((np.abs(ce - cut) < (np.median(moved) if len(moved) else 0.0)).sum()))
    log(json.dumps({tag: out}, indent=1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000); ap.add_argument("--train-images", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=4); ap.add_argument("--pool", type=int, default=1024)
    ap.add_argument("--out", default="gpurun_out/r6_trained_weights.json")
    a = ap.parse_args()
    res = dict(what="round-5 distributional claims re-measured on a detector trained by this repo's own training step (tools/trained_weights_study.py)",
               steps=a.steps, batch=a.batch, train_images=a.train_images, pool_images=a.pool, csrc_sha1=None)
    sd, hist, data = train_detector(a.steps, a.train_images, a.batch)
    res["loss_history"] = hist
    m = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=600, max_size=1000).to("cuda"); m.load_state_dict(sd); m.eval()
    res["detection_quality_trained"] = detection_quality(m, data)
    del m; torch.cuda.empty_cache()
    m = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=600, max_size=1000).to("cuda"); m.load_state_dict(synth.pseudo_trained_frcnn(21, 50, seed=0)); m.eval()
    res["detection_quality_pseudo_trained"] = detection_quality(m, data)
    del m; torch.cuda.empty_cache()
    res["trained"] = pool_study(sd, "trained", a.pool)
    res["pseudo_trained"] = pool_study(synth.pseudo_trained_frcnn(21, 50, seed=0), "pseudo_trained", a.pool)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
