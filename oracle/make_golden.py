"""Generate tests/golden/*.npz from the IMPORTED reference (TEST INFRASTRUCTURE ONLY).

Run in the build container only (needs /root/reference):
    python oracle/make_golden.py
The reference's get_uncertainty / cls_kldiv / cald_helper functions are executed as they lie
in /root/reference (through oracle/ref_harness.py) on a deterministic fake detector; only the
resulting input/output vectors are committed.  No reference source is copied.
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def image_seed(base_seed, pool_pos):
    return (int(base_seed) * 1000003 + int(pool_pos)) & 0xFFFFFFFFFFFFFFFF


def synth_image(rs, H, W):
    img = (rs.rand(H, W, 3) * 255).astype(np.uint8)
    # a few flat rectangles so that PIL resampling sees edges
    for _ in range(4):
        y0, x0 = rs.randint(0, H - 8), rs.randint(0, W - 8)
        img[y0:y0 + rs.randint(4, H // 2), x0:x0 + rs.randint(4, W // 2)] = rs.randint(0, 256, 3)
    return img


def fake_dets(rs, n, H, W, C, kind, sorted_scores=True):
    """kind 'softmax' -> FRCNN-like (labels 1..C-1, rows sum to 1); 'sigmoid' -> RetinaNet-like
    (labels 0..C-1, independent sigmoids, grouped by class)."""
    x0 = rs.rand(n) * (W * 0.7); y0 = rs.rand(n) * (H * 0.7)
    bw = 4 + rs.rand(n) * (W * 0.5); bh = 4 + rs.rand(n) * (H * 0.5)
    boxes = np.stack([x0, y0, np.minimum(x0 + bw, W), np.minimum(y0 + bh, H)], 1).astype(np.float32)
    logits = rs.randn(n, C).astype(np.float32) * 2.0
    if kind == "softmax":
        e = np.exp(logits - logits.max(1, keepdims=True)); sc = (e / e.sum(1, keepdims=True)).astype(np.float32)
        labels = (1 + np.argmax(sc[:, 1:], 1)).astype(np.int64) if n else np.zeros(0, np.int64)
        scores = sc[np.arange(n), labels] if n else np.zeros(0, np.float32)
        pm = sc[:, 1:].max(1) if n else np.zeros(0, np.float32)
        if sorted_scores and n:
            o = np.argsort(-scores, kind="stable")
            boxes, sc, labels, scores, pm = boxes[o], sc[o], labels[o], scores[o], pm[o]
    else:
        sc = (1.0 / (1.0 + np.exp(-logits))).astype(np.float32)
        labels = np.sort(rs.randint(0, C, n)).astype(np.int64)
        scores = sc[np.arange(n), labels] if n else np.zeros(0, np.float32)
        pm = sc.max(1) if n else np.zeros(0, np.float32)
    return dict(boxes=boxes, labels=labels, scores=scores.astype(np.float32), prob_max=pm.astype(np.float32),
                scores_cls=sc)


class FakeModel:
    """Returns pre-generated detections in call order and records the image tensors it is given."""

    def __init__(self, outputs):
        self.outputs = outputs
        self.calls = 0
        self.seen = []
        self.seen_float = []

    def eval(self):
        return self

    def __call__(self, imgs):
        self.seen.append((imgs[0].detach().clone() * 255).round().to(torch.uint8).permute(1, 2, 0).numpy())
        self.seen_float.append(imgs[0].detach().clone().permute(1, 2, 0).numpy())
        o = self.outputs[self.calls]
        self.calls += 1
        return [{k: torch.from_numpy(v.copy()) for k, v in o.items()}]


class SeededLoader:
    """Re-seeds Python's `random` per pool position (SURVEY section 7 'RNG-dependent augmentations')."""

    def __init__(self, images, base_seed):
        self.images, self.base_seed = images, base_seed

    def __iter__(self):
        from PIL import Image
        for pos, img in enumerate(self.images):
            random.seed(image_seed(self.base_seed, pos))
            torch.manual_seed(image_seed(self.base_seed, pos))
            yield (Image.fromarray(img),), (None,)


def gen_scoring(ct, name, kind, C, augs, ref_counts, aug_counts, seed, n_views=None, sizes=((96, 128), (120, 90), (75, 100)),
                seen_images=3, float_views=()):
    rs = np.random.RandomState(seed)
    images, outputs, per_image = [], [], []
    n_views = len(augs) if n_views is None else n_views
    for i, nref in enumerate(ref_counts):
        H, W = sizes[i % len(sizes)]
        img = synth_image(rs, H, W)
        images.append(img)
        ref = fake_dets(rs, nref, H, W, C, kind)
        outs = [ref]
        if nref > 0:
            for a in range(n_views):
                m = aug_counts[(i + a) % len(aug_counts)]
                d = fake_dets(rs, m, H, W, C, kind)
                if m and (i + a) % 4 == 1:           # an all-zero IoU row: push detections far away
                    d["boxes"] = d["boxes"] + np.float32(10000.0)
                if m > 2 and (i + a) % 4 == 2:       # duplicate best box -> first index must win
                    d["boxes"][1] = d["boxes"][0]
                outs.append(d)
        outputs.extend(outs)
        per_image.append(len(outs))
    model = FakeModel(outputs)
    ct.args.bp = 1.3
    cons, cls = ct.get_uncertainty(model, SeededLoader(images, 7), list(augs), C)
    assert model.calls == len(outputs)
    blob = {"augs": np.array(augs), "C": C, "kind": kind, "bp": 1.3, "base_seed": 7, "n_images": len(images),
            "per_image": np.array(per_image), "consistency": np.array(cons, np.float64),
            "cls_all": np.stack([np.asarray(c, np.float64) for c in cls])}
    k = 0
    for i, img in enumerate(images):
        blob["img%d" % i] = img
        for v in range(per_image[i]):
            for key, val in outputs[k].items():
                blob["det%d_%d_%s" % (i, v, key)] = val
            if i < seen_images:
                blob["seen%d_%d" % (i, v)] = model.seen[k]
                if v in float_views:
                    blob["seenf%d_%d" % (i, v)] = model.seen_float[k]
            k += 1
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **blob)
    print(name, "consistency", np.round(cons, 4))


def gen_helpers(ch):
    from PIL import Image
    rs = np.random.RandomState(3)
    blob = {}
    for i, (H, W) in enumerate([(150, 200), (133, 200), (120, 90), (64, 64)]):
        img = synth_image(rs, H, W)
        boxes = fake_dets(rs, 6, H, W, 21, "softmax")["boxes"]
        blob["img%d" % i] = img; blob["boxes%d" % i] = boxes
        fi, fb = ch.HorizontalFlip(Image.fromarray(img), torch.from_numpy(boxes))
        blob["flip_img%d" % i] = (fi * 255).round().to(torch.uint8).permute(1, 2, 0).numpy(); blob["flip_boxes%d" % i] = fb.numpy()
        for r in (0.8, 1.2, 0.7):
            ri, rb = ch.resize(Image.fromarray(img), torch.from_numpy(boxes), r)
            blob["resize%d_%d_img" % (i, int(r * 10))] = (ri * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
            blob["resize%d_%d_boxes" % (i, int(r * 10))] = rb.numpy()
        for s in (11, 12, 13):
            random.seed(s)
            ci = ch.cutout(Image.fromarray(img), torch.from_numpy(boxes), None, 2)
            blob["cutout%d_%d_img" % (i, s)] = (ci * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
        ri, rb = ch.rotate(Image.fromarray(img), torch.from_numpy(boxes), 5)
        blob["rotate%d_img" % i] = (ri * 255).round().to(torch.uint8).permute(1, 2, 0).numpy(); blob["rotate%d_boxes" % i] = rb.numpy()
        if i >= 2:
            torch.manual_seed(31 + i)
            blob["ga%d_img" % i] = ch.GaussianNoise(Image.fromarray(img), 16).permute(1, 2, 0).numpy()    # float32 HWC
        for s in (21, 22):
            torch.manual_seed(s)
            si = ch.SaltPepperNoise(Image.fromarray(img), 0.1)
            blob["sp%d_%d_img" % (i, s)] = (si * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
        b2 = fake_dets(rs, 4, H, W, 21, "softmax")["boxes"]
        blob["boxes_b%d" % i] = b2
        blob["intersect%d" % i] = ch.intersect(torch.from_numpy(boxes), torch.from_numpy(b2)).numpy()
    for s in (0, 1, 12345, (1 << 40) + 17):
        random.seed(s)
        blob["pyrandom_%d" % s] = np.array([random.random() for _ in range(8)])
    np.savez_compressed(os.path.join(OUT, "helpers.npz"), **blob)
    print("helpers ok")


def gen_js():
    import scipy.stats
    rs = np.random.RandomState(5)
    P, Q, J = [], [], []
    for C in (21, 91):
        for t in range(40):
            if t % 2 == 0:
                p = rs.rand(C).astype(np.float32); q = rs.rand(C).astype(np.float32)       # sigmoid-like, un-normalised
            else:
                a = rs.randn(C).astype(np.float32) * 3; b = rs.randn(C).astype(np.float32) * 3
                p = np.exp(a - a.max()); p = (p / p.sum()).astype(np.float32)
                q = np.exp(b - b.max()); q = (q / q.sum()).astype(np.float32)
            if t % 10 == 3:
                p[2] = 0.0
            m = (p + q) / 2
            js = 0.5 * scipy.stats.entropy(p, m) + 0.5 * scipy.stats.entropy(q, m)
            P.append(np.pad(p, (0, 91 - C))); Q.append(np.pad(q, (0, 91 - C))); J.append((C, float(js)))
    np.savez_compressed(os.path.join(OUT, "js.npz"), p=np.stack(P), q=np.stack(Q), cj=np.array(J))
    print("js ok")


def gen_selection(ct):
    rs = np.random.RandomState(9)
    blob = {}
    for case, (uniform, nzero) in enumerate([(False, 0), (False, 3), (True, 0), (True, 2)]):
        n, Cm1, budget = 60, 20, 12
        cls_corrs = rs.rand(n, Cm1) * (rs.rand(n, Cm1) > 0.6)
        for z in range(nzero):
            cls_corrs[5 + 7 * z] = 0
        labeled = [[{"labels": torch.from_numpy(rs.randint(1, Cm1 + 1, rs.randint(1, 6)))}] for _ in range(15)]
        loader = [(None, t) for t in labeled]
        ct.args.uniform = uniform
        sel = ct.cls_kldiv(loader, list(cls_corrs), budget, 0)
        blob["cls_corrs%d" % case] = cls_corrs
        blob["uniform%d" % case] = uniform
        blob["budget%d" % case] = budget
        blob["labels%d" % case] = np.array([np.pad(t[0]["labels"].numpy(), (0, 8 - len(t[0]["labels"])), constant_values=-1) for t in labeled])
        blob["sel%d" % case] = np.array([int(s) for s in sel])
    ct.args.uniform = False
    u = np.round(rs.rand(200), 1); u[::7] = 0.0
    blob["argsort_in"] = u
    blob["argsort_out"] = np.argsort(u)
    np.savez_compressed(os.path.join(OUT, "selection.npz"), **blob)
    print("selection ok")


def gen_voc_results():
    """detection/voc_eval.py:188-222 _write_voc_results_file on synthetic detections -> text fixture."""
    import importlib, glob
    ve = importlib.import_module("detection.voc_eval")
    rs = np.random.RandomState(4)
    classes = ('__background__', 'aeroplane', 'bicycle', 'bird')
    names = ["2008_%06d" % i for i in (12, 3, 7, 3)]          # one repeated index
    all_boxes = [[] for _ in classes]
    blob = {"names": np.array(names), "classes": np.array(classes)}
    for ii, n in enumerate(names):
        for c in range(len(classes)):
            k = rs.randint(0, 3)
            if k:
                d = torch.from_numpy((rs.rand(k, 5) * 300).astype(np.float32)); d[:, 4] = torch.from_numpy(rs.rand(k).astype(np.float32))
                all_boxes[c].append([d]); blob["d%d_%d" % (ii, c)] = d.numpy()
            else:
                all_boxes[c].append([])
    ve._write_voc_results_file([list(b) for b in all_boxes], list(names), "cald_golden_voc", classes)
    for f in sorted(glob.glob("/tmp/cald_golden_voc/det_test_*.txt")):
        blob["file_" + os.path.basename(f)] = np.array(open(f).read())
    np.savez_compressed(os.path.join(OUT, "voc_results.npz"), **blob)
    print("voc_results ok")


def gen_coco_results():
    """detection/coco_eval.py:76-98 CocoEvaluator.prepare_for_coco_detection on synthetic predictions."""
    import importlib, json
    ce = importlib.import_module("detection.coco_eval")
    rs = np.random.RandomState(6)
    preds, blob = {}, {}
    for image_id, n in ((139, 3), (285, 0), (632, 7)):
        d = {"boxes": torch.from_numpy((rs.rand(n, 4) * 300).astype(np.float32)), "scores": torch.from_numpy(rs.rand(n).astype(np.float32)),
             "labels": torch.from_numpy(rs.randint(1, 91, n).astype(np.int64))}
        d["boxes"][:, 2:] += d["boxes"][:, :2]
        preds[image_id] = d
        for k, v in d.items():
            blob["p%d_%s" % (image_id, k)] = v.numpy()
    ev = ce.CocoEvaluator.__new__(ce.CocoEvaluator)
    blob["ids"] = np.array(list(preds.keys()))
    blob["json"] = np.array(json.dumps(ev.prepare_for_coco_detection(preds)))
    np.savez_compressed(os.path.join(OUT, "coco_results.npz"), **blob)
    print("coco_results ok")


def gen_baselines():
    lt, ls = ref_harness.load_baselines()
    rs = np.random.RandomState(8)
    blob = {}
    # ---- lt_c: detections with props ----
    lt_outs = []
    for i, n in enumerate([0, 1, 7, 100]):
        d = fake_dets(rs, n, 120, 160, 21, "softmax")
        d["props"] = (d["boxes"] + rs.randn(n, 4).astype(np.float32) * 6).astype(np.float32)
        if n > 3:
            d["props"][2] = d["props"][2] + np.float32(5000)      # disjoint -> calcu_iou returns 0
        lt_outs.append(d)
        for k, v in d.items():
            blob["lt%d_%s" % (i, k)] = v

    class LtModel:
        def eval(self): return self
        def __call__(self, images): return [{k: torch.from_numpy(v.copy()) for k, v in lt_outs[self.i].items()}]
    m = LtModel(); res = []
    for i in range(len(lt_outs)):
        m.i = i
        res += lt.get_uncertainty(m, [((torch.zeros(3, 8, 8),), (None,))])
    blob["lt_unc"] = np.array(res, np.float64)
    # ---- ls_c: reference view + six noisy views ----
    images, outputs, per = [], [], []
    for i, nref in enumerate([4, 0, 35, 12]):
        H, W = [(60, 80), (75, 50)][i % 2]
        images.append(synth_image(rs, H, W))
        outs = [fake_dets(rs, nref, H, W, 21, "softmax")]
        if nref:
            for k in range(6):
                outs.append(fake_dets(rs, [9, 0, 40, 3, 17, 1][(i + k) % 6], H, W, 21, "softmax"))
        outputs.extend(outs); per.append(len(outs))
    model = FakeModel(outputs)
    model_seen_float = []
    orig_call = model.__call__

    class Rec(FakeModel):
        def __call__(self, imgs):
            model_seen_float.append(imgs[0].detach().clone().permute(1, 2, 0).numpy())
            return FakeModel.__call__(self, imgs)
    model = Rec(outputs)
    stab = ls.get_uncertainty(model, SeededLoader(images, 5))
    blob["ls_unc"] = np.array(stab, np.float64); blob["ls_per"] = np.array(per); blob["ls_n"] = len(images)
    k = 0
    for i, img in enumerate(images):
        blob["ls_img%d" % i] = img
        for v in range(per[i]):
            for key in ("boxes", "prob_max", "labels"):
                blob["ls%d_%d_%s" % (i, v, key)] = outputs[k][key]
            if i == 0:
                blob["ls_seen%d_%d" % (i, v)] = model_seen_float[k]
            k += 1
    np.savez_compressed(os.path.join(OUT, "baselines.npz"), **blob)
    print("baselines ok", res, stab)


def main():
    os.makedirs(OUT, exist_ok=True)
    ct, ch = ref_harness.load_reference()
    gen_scoring(ct, "scoring_frcnn_F", "softmax", 21, ["flip"], [3, 0, 41, 49, 50, 51, 100, 1], [5, 0, 100, 1, 17], 1)
    gen_scoring(ct, "scoring_frcnn_FCD", "softmax", 21, ["flip", "cut_out", "smaller_resize"],
                [3, 0, 41, 49, 50, 51, 100, 1, 12, 7], [5, 0, 100, 1, 17, 33], 2)
    gen_scoring(ct, "scoring_retina_FCD", "sigmoid", 21, ["flip", "cut_out", "smaller_resize"],
                [6, 0, 45, 300, 2, 80], [9, 0, 600, 3, 40], 3)
    gen_scoring(ct, "scoring_frcnn_FSCDR", "softmax", 21, ["flip", "sp", "cut_out", "smaller_resize", "rotation"],
                [5, 0, 44, 9], [7, 0, 100, 2, 30], 6)
    # every augmentation branch of get_uncertainty that runs in the reference (multi_color_adjust raises NameError there):
    # 28 views per image, in the reference's order; views 1..7 are GaussianNoise (kept as float for image 0)
    all_augs = ["flip", "ga", "multi_ga", "color_adjust", "color_swap", "sp", "multi_sp", "cut_out", "multi_cut_out",
                "multi_resize", "larger_resize", "smaller_resize", "rotation"]
    gen_scoring(ct, "scoring_frcnn_ALL", "softmax", 21, all_augs, [6, 0, 43], [7, 0, 60, 2, 25], 11, n_views=28,
                sizes=((48, 64), (60, 44)), seen_images=1, float_views=tuple(range(1, 8)))
    gen_scoring(ct, "scoring_frcnn_coco_FD", "softmax", 91, ["flip", "smaller_resize"], [10, 60, 0, 2], [20, 3, 100], 4)
    gen_helpers(ch)
    gen_js()
    gen_selection(ct)
    gen_voc_results()
    gen_coco_results()
    gen_baselines()


if __name__ == "__main__":
    main()
