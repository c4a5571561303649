"""Golden vectors for the part of the TRAINING step whose code lives in the reference repo itself: TEST
INFRASTRUCTURE, run only in the build container:
    python oracle/make_golden_train_losses.py

1. RetinaNet's losses.  The bodies of ``RetinaNet.compute_loss`` (detection/retinanet_cal.py:389-400, the matcher driver),
   ``RetinaNetClassificationHead.compute_loss`` (:100-133, focal loss / max(1, #fg), ignore band) and
   ``RetinaNetRegressionHead.compute_loss`` (:185-223, L1 / max(1, #fg) on encoded targets) are executed as they lie in
   /root/reference under the stub harness.  The four torchvision 0.8.2 primitives they call -- ``sigmoid_focal_loss``,
   ``box_ops.box_iou``, ``det_utils.Matcher`` and ``BoxCoder.encode_single`` -- are not installed and are bound to the plain
   restatements below; everything else (which anchors are foreground / ignored, label scatter, per-image
   normalisers, the batch mean) is reference code.  Stored: inputs, the matched indices the reference's driver produced, both
   losses in float32 (the reference's dtype) and in float64 (same code on double inputs: the checker's dtype).
2. The training loop.  ``cald_train.train_one_epoch`` (cald_train.py:40-74) with ``detection/utils.warmup_lr_scheduler``
   (utils.py:239-247) is run on a small differentiable stand-in model with torch.optim.SGD: the learning rate seen by every
   iteration, the summed loss and the parameter trajectory pin the loop's order of operations (warm-up only in epoch 0,
   ``min(1000, len - 1)`` iterations, zero_grad / backward / step / scheduler.step).
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


# ---- torchvision 0.8.2 primitives (Appendix A), dtype-generic torch ops ----
def sigmoid_focal_loss(inputs, targets, alpha=0.25, gamma=2, reduction="none"):
    p = torch.sigmoid(inputs)
    ce_loss = torch.nn.functional.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce_loss * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.sum() if reduction == "sum" else (loss.mean() if reduction == "mean" else loss)


def box_iou(boxes1, boxes2):
    area1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    area2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2]); rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area1[:, None] + area2 - inter)


class Matcher:
    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold, low_threshold, allow_low_quality_matches=False):
        self.high_threshold, self.low_threshold, self.allow_low_quality_matches = high_threshold, low_threshold, allow_low_quality_matches

    def __call__(self, match_quality_matrix):
        matched_vals, matches = match_quality_matrix.max(dim=0)
        all_matches = matches.clone() if self.allow_low_quality_matches else None
        below = matched_vals < self.low_threshold
        between = (matched_vals >= self.low_threshold) & (matched_vals < self.high_threshold)
        matches[below] = self.BELOW_LOW_THRESHOLD
        matches[between] = self.BETWEEN_THRESHOLDS
        if self.allow_low_quality_matches:
            highest_quality_foreach_gt, _ = match_quality_matrix.max(dim=1)
            gt_pred_pairs = torch.where(match_quality_matrix == highest_quality_foreach_gt[:, None])
            pred_inds = gt_pred_pairs[1]
            matches[pred_inds] = all_matches[pred_inds]
        return matches


class BoxCoder:
    def __init__(self, weights, bbox_xform_clip=math.log(1000. / 16)):
        self.weights, self.bbox_xform_clip = weights, bbox_xform_clip

    def encode_single(self, reference_boxes, proposals):
        dtype, device = reference_boxes.dtype, reference_boxes.device
        w = torch.as_tensor(self.weights, dtype=dtype, device=device)
        px1, py1, px2, py2 = [proposals[:, i].unsqueeze(1) for i in range(4)]
        rx1, ry1, rx2, ry2 = [reference_boxes[:, i].unsqueeze(1) for i in range(4)]
        ex_w, ex_h = px2 - px1, py2 - py1
        ex_cx, ex_cy = px1 + 0.5 * ex_w, py1 + 0.5 * ex_h
        gt_w, gt_h = rx2 - rx1, ry2 - ry1
        gt_cx, gt_cy = rx1 + 0.5 * gt_w, ry1 + 0.5 * gt_h
        return torch.cat((w[0] * (gt_cx - ex_cx) / ex_w, w[1] * (gt_cy - ex_cy) / ex_h,
                          w[2] * torch.log(gt_w / ex_w), w[3] * torch.log(gt_h / ex_h)), dim=1)


def retina_anchors(Hp, Wp):
    """RetinaNet anchors of a padded Hp x Wp batch (retinanet_cal.py:346-351 sizes; level-major, pixel, 9 per location)."""
    from oracle import oracle as orc
    base = np.stack([orc.base_anchors(list(s), [0.5, 1.0, 2.0]) for s in orc.retina_anchor_sizes()])
    out, level_hw = [], []
    h, w = (Hp + 7) // 8, (Wp + 7) // 8
    for l in range(5):
        level_hw.append((h, w))
        sh, sw = Hp // h, Wp // w
        ys, xs = np.meshgrid(np.arange(h) * sh, np.arange(w) * sw, indexing="ij")
        shifts = np.stack([xs, ys, xs, ys], -1).reshape(-1, 1, 4).astype(np.float32)
        out.append((shifts + base[l][None]).reshape(-1, 4))
        h, w = (h + 1) // 2, (w + 1) // 2
    return np.concatenate(out).astype(np.float32), np.array(level_hw)


def build_model(rc, K):
    rn = rc.RetinaNet.__new__(rc.RetinaNet)
    torch.nn.Module.__init__(rn)
    rn.proposal_matcher = Matcher(0.5, 0.4, allow_low_quality_matches=True)      # retinanet_cal.py:366 fg 0.5 / bg 0.4
    head = rc.RetinaNetHead.__new__(rc.RetinaNetHead); torch.nn.Module.__init__(head)
    ch = rc.RetinaNetClassificationHead.__new__(rc.RetinaNetClassificationHead); torch.nn.Module.__init__(ch)
    ch.num_classes, ch.num_anchors, ch.BETWEEN_THRESHOLDS = K, 9, Matcher.BETWEEN_THRESHOLDS
    rh = rc.RetinaNetRegressionHead.__new__(rc.RetinaNetRegressionHead); torch.nn.Module.__init__(rh)
    rh.box_coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
    head.classification_head, head.regression_head = ch, rh
    rn.head = head
    return rn


def loss_cases(rc):
    rs = np.random.RandomState(41)
    blob = {}
    # (K, Hp, Wp, per-image list of gt boxes [x1 y1 x2 y2 label], logit gain, logit bias)
    def rnd_boxes(n, Hp, Wp, lo, hi, K):
        x0 = rs.rand(n) * (Wp - lo); y0 = rs.rand(n) * (Hp - lo)
        w = lo + rs.rand(n) * (hi - lo); h = lo + rs.rand(n) * (hi - lo)
        return np.stack([x0, y0, np.minimum(x0 + w, Wp), np.minimum(y0 + h, Hp), rs.randint(0, K, n)], 1)
    anchors128, _ = retina_anchors(96, 128)
    specs = [
        ("normal", 21, 96, 128, [rnd_boxes(3, 96, 128, 24, 70, 21)], 2.0, -4.0),
        ("batch2_tiny_gt_low_quality_only", 21, 96, 128, [np.array([[50.2, 40.1, 54.0, 43.7, 5]]), rnd_boxes(2, 96, 128, 40, 90, 21)], 2.0, -4.0),
        ("many_overlapping_gts", 4, 128, 160, [rnd_boxes(9, 128, 160, 30, 60, 4), rnd_boxes(7, 128, 160, 16, 120, 4)], 1.0, -1.0),
        # a gt that IS an anchor (IoU 1), the same box twice with two labels (tie -> first), a box reaching outside the image
        ("gt_equals_anchor_duplicate_and_outside", 21, 96, 128,
         [np.concatenate([np.c_[anchors128[[1200, 1200, 2000]], [[3], [9], [0]]], np.array([[100.0, 60.0, 150.0, 120.0, 20]])])], 2.0, -4.0),
        ("extreme_logits", 7, 64, 64, [rnd_boxes(2, 64, 64, 20, 40, 7), rnd_boxes(1, 64, 64, 30, 50, 7), rnd_boxes(4, 64, 64, 10, 30, 7)], 12.0, 0.0),
        ("coco_classes", 91, 64, 96, [rnd_boxes(4, 64, 96, 16, 60, 91), rnd_boxes(1, 64, 96, 40, 60, 91)], 2.0, -4.6),
    ]
    for k, (name, K, Hp, Wp, gts, gain, bias) in enumerate(specs):
        anchors, level_hw = retina_anchors(Hp, Wp)
        A = anchors.shape[0]
        N = len(gts)
        cls = (rs.randn(N, A, K) * gain + bias).astype(np.float32)
        reg = (rs.randn(N, A, 4) * 0.6).astype(np.float32)
        rn = build_model(rc, K)
        seen = {}
        inner = rn.head.compute_loss

        def spy(targets, head_outputs, anchors_, matched_idxs, _inner=inner, _seen=seen):
            _seen["matched"] = [m.clone() for m in matched_idxs]
            return _inner(targets, head_outputs, anchors_, matched_idxs)
        rn.head.compute_loss = spy
        res = {}
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            targets = [{"boxes": torch.from_numpy(g[:, :4].astype(np.float32)).to(dt), "labels": torch.from_numpy(g[:, 4].astype(np.int64))} for g in gts]
            head = {"cls_logits": torch.from_numpy(cls).to(dt), "bbox_regression": torch.from_numpy(reg).to(dt)}
            out = rn.compute_loss(targets, head, [torch.from_numpy(anchors).to(dt)] * N)
            res[tag] = out
            blob["l%d_cls_%s" % (k, tag)] = np.array(out["classification"].item(), np.float64)
            blob["l%d_reg_%s" % (k, tag)] = np.array(out["bbox_regression"].item(), np.float64)
            if tag == "f32":
                m32 = [m.numpy().astype(np.int64) for m in seen["matched"]]
            else:
                for a, b in zip(m32, seen["matched"]):
                    assert np.array_equal(a, b.numpy()), "matcher decisions differ between float32 and float64 inputs: move the boxes"
        blob.update({"l%d_K" % k: K, "l%d_hw" % k: np.array([Hp, Wp]), "l%d_level_hw" % k: level_hw, "l%d_anchors" % k: anchors,
                     "l%d_cls_logits" % k: cls, "l%d_bbox_regression" % k: reg, "l%d_N" % k: N, "l%d_matched" % k: np.stack(m32)})
        for i, g in enumerate(gts):
            blob["l%d_gt%d" % (k, i)] = g[:, :4].astype(np.float32); blob["l%d_labels%d" % (k, i)] = g[:, 4].astype(np.int64)
        m = np.stack(m32)
        print("loss case %d %-40s N=%d anchors=%d fg=%s ignored=%s cls=%.6f reg=%.6f" % (
            k, name, N, A, (m >= 0).sum(1), (m == -2).sum(1), blob["l%d_cls_f32" % k], blob["l%d_reg_f32" % k]))
        assert (m == -2).sum() > 0 and (m >= 0).sum() > 0
    blob["l_n"] = len(specs)
    return blob


class _Toy(torch.nn.Module):
    """Differentiable stand-in for the detector inside the reference's loop: two 'losses' of the parameters and the batch."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([0.5, -1.25, 2.0], dtype=torch.float64))
        self.b = torch.nn.Parameter(torch.tensor([0.1], dtype=torch.float64))

    def forward(self, images, targets):
        x = torch.stack([im.double().mean() for im in images])
        t = torch.stack([tg["boxes"].double().sum() for tg in targets])
        pred = x[:, None] * self.w[None, :] + self.b
        return {"loss_a": ((pred.sum(1) - t) ** 2).mean() * 0.1, "loss_b": (self.w ** 2).sum() * 0.01 + self.b.abs().sum()}


def loop_cases(ct):
    blob = {}
    rs = np.random.RandomState(5)
    for k, (n_iter, epochs) in enumerate([(6, 2), (1, 1), (3, 1), (12, 2)]):
        data = [([torch.from_numpy(rs.rand(3, 4, 5)) for _ in range(2)],
                 [{"boxes": torch.from_numpy(rs.rand(2, 4))} for _ in range(2)]) for _ in range(n_iter)]
        model = _Toy()
        lrs, params = [], []

        class SpySGD(torch.optim.SGD):
            def step(self, closure=None):
                lrs.append(self.param_groups[0]["lr"])          # the rate the update is taken with
                r = super().step(closure)
                params.append(np.concatenate([model.w.detach().numpy().copy(), model.b.detach().numpy().copy()]))
                return r
        opt = SpySGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3)
        for ep in range(epochs):
            ct.train_one_epoch(model, opt, data, torch.device("cpu"), 0, ep, 1000)
        blob["t%d_iters" % k] = n_iter; blob["t%d_epochs" % k] = epochs
        blob["t%d_lrs" % k] = np.array(lrs, np.float64); blob["t%d_params" % k] = np.stack(params)
        for i, (ims, tgs) in enumerate(data):
            blob["t%d_im%d" % (k, i)] = np.stack([im.numpy() for im in ims]); blob["t%d_bx%d" % (k, i)] = np.stack([t["boxes"].numpy() for t in tgs])
        print("loop case", k, "lrs", np.round(lrs, 6)[:8], "final params", params[-1])
    blob["t_n"] = 4
    return blob


if __name__ == "__main__":
    ct, _ = ref_harness.load_reference()
    rc = sys.modules["detection.retinanet_cal"]
    rc.sigmoid_focal_loss = sigmoid_focal_loss
    rc.box_ops.box_iou = box_iou
    blob = loss_cases(rc)
    blob.update(loop_cases(ct))
    path = os.path.join(OUT, "train_losses.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path))
