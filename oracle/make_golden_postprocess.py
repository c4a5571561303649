"""Golden vectors for the detection-bookkeeping rows A19 / A22 (SURVEY 8c "optional extra pin"): TEST INFRASTRUCTURE.

The BODIES of the reference's own ``RoIHeads.postprocess_detections`` (detection/frcnn_la.py:32-87) and
``RetinaNet.postprocess_detections`` (detection/retinanet_cal.py:402-490) are executed, as they lie in /root/reference,
under the stub harness.  The torchvision primitives they call (BoxCoder.decode, clip_boxes_to_image, nms, batched_nms,
remove_small_boxes -- torchvision 0.8.2 is not installed) are bound to the plain restatements below.
What this pins against reference code is the expand / flatten / gather / concatenate index bookkeeping:
which (proposal, class) pairs become candidates, what scores_cls / prob_max / props rows they carry, the class-major
order of RetinaNet's output, the label conventions.  Run in the build container:
    python oracle/make_golden_postprocess.py
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


# ---- torchvision 0.8.2 primitives (Appendix A), float32 torch ops ----
class BoxCoder:
    def __init__(self, weights, bbox_xform_clip=math.log(1000. / 16)):
        self.weights, self.clip = weights, bbox_xform_clip

    def decode_single(self, rel_codes, boxes):
        boxes = boxes.to(rel_codes.dtype)
        widths = boxes[:, 2] - boxes[:, 0]; heights = boxes[:, 3] - boxes[:, 1]
        ctr_x = boxes[:, 0] + 0.5 * widths; ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = self.weights
        dx = rel_codes[:, 0::4] / wx; dy = rel_codes[:, 1::4] / wy
        dw = torch.clamp(rel_codes[:, 2::4] / ww, max=self.clip); dh = torch.clamp(rel_codes[:, 3::4] / wh, max=self.clip)
        pcx = dx * widths[:, None] + ctr_x[:, None]; pcy = dy * heights[:, None] + ctr_y[:, None]
        pw = torch.exp(dw) * widths[:, None]; ph = torch.exp(dh) * heights[:, None]
        x1 = pcx - torch.tensor(0.5) * pw; y1 = pcy - torch.tensor(0.5) * ph
        x2 = pcx + torch.tensor(0.5) * pw; y2 = pcy + torch.tensor(0.5) * ph
        return torch.stack((x1, y1, x2, y2), dim=2).flatten(1)

    def decode(self, rel_codes, boxes):
        concat = torch.cat(list(boxes), dim=0)
        n = concat.shape[0]
        return self.decode_single(rel_codes.reshape(n, -1), concat).reshape(n, -1, 4)


def clip_boxes_to_image(boxes, size):
    h, w = size
    bx = boxes[..., 0::2].clamp(min=0, max=w); by = boxes[..., 1::2].clamp(min=0, max=h)
    return torch.stack((bx, by), dim=boxes.dim()).reshape(boxes.shape)


def nms(boxes, scores, thr):
    b = boxes.numpy().astype(np.float32); s = scores.numpy()
    order = sorted(range(len(s)), key=lambda i: (-s[i], i))
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    keep, dead = [], np.zeros(len(s), bool)
    for ii, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        for j in order[ii + 1:]:
            if dead[j]:
                continue
            w = np.float32(max(np.float32(0), min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0])))
            h = np.float32(max(np.float32(0), min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1])))
            inter = np.float32(w * h)
            if inter / np.float32(np.float32(areas[i] + areas[j]) - inter) > np.float32(thr):
                dead[j] = True
    return torch.tensor(keep, dtype=torch.int64)


def batched_nms(boxes, scores, idxs, thr):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    offsets = idxs.to(boxes) * (boxes.max() + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, thr)


def remove_small_boxes(boxes, min_size):
    ws, hs = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    return torch.where((ws >= min_size) & (hs >= min_size))[0]


def bind(box_ops):
    box_ops.clip_boxes_to_image = clip_boxes_to_image
    box_ops.nms = nms
    box_ops.batched_nms = batched_nms
    box_ops.remove_small_boxes = remove_small_boxes


def frcnn_cases(fl):
    rs = np.random.RandomState(17)
    rh = fl.RoIHeads()
    rh.box_coder = BoxCoder((10., 10., 5., 5.)); rh.score_thresh = 0.05; rh.nms_thresh = 0.5; rh.detections_per_img = 100
    blob = {}
    specs = [(300, 21, 240, 320, 3.0), (250, 91, 400, 600, 4.0), (5, 21, 100, 100, 3.0), (60, 21, 150, 200, 0.0), (500, 21, 300, 500, 6.0)]
    for k, (R, C, H, W, gain) in enumerate(specs):
        x0 = rs.rand(R) * W * 0.8; y0 = rs.rand(R) * H * 0.8
        props = np.stack([x0, y0, np.minimum(x0 + 8 + rs.rand(R) * W * 0.5, W), np.minimum(y0 + 8 + rs.rand(R) * H * 0.5, H)], 1).astype(np.float32)
        logits = (rs.randn(R, C) * gain).astype(np.float32)
        deltas = (rs.randn(R, 4 * C) * np.array([1.5, 1.5, 0.8, 0.8] * C)).astype(np.float32)
        deltas[::17, 2::4] = 30.0                      # exercises the log(1000/16) clamp
        outs = rh.postprocess_detections(torch.from_numpy(logits), torch.from_numpy(deltas), [torch.from_numpy(props)], [(H, W)])
        names = ("boxes", "scores", "labels", "props", "prob_max", "scores_cls")
        blob.update({"f%d_logits" % k: logits, "f%d_deltas" % k: deltas, "f%d_props" % k: props, "f%d_hw" % k: np.array([H, W])})
        for n, o in zip(names, outs):
            blob["f%d_out_%s" % (k, n)] = o[0].numpy()
        print("frcnn case", k, "dets", outs[0][0].shape[0])
    blob["f_n"] = len(specs)
    return blob


def retina_cases(rc):
    from oracle import oracle as orc
    rs = np.random.RandomState(23)
    rn = rc.RetinaNet.__new__(rc.RetinaNet)
    torch.nn.Module.__init__(rn)
    rn.box_coder = BoxCoder((1., 1., 1., 1.)); rn.score_thresh = 0.05; rn.nms_thresh = 0.5; rn.detections_per_img = 300
    base = np.stack([orc.base_anchors(list(s), [0.5, 1.0, 2.0]) for s in orc.retina_anchor_sizes()])   # [5][9][4]
    blob = {"r_base": base}
    specs = [(21, 64, 96, 60, 90, -4.5), (21, 64, 64, 64, 64, -6.5), (4, 96, 128, 96, 120, -0.5)]
    for k, (K, Hp, Wp, Hr, Wr, bias) in enumerate(specs):
        shapes = []
        h, w = Hp // 8, Wp // 8
        for _ in range(5):
            shapes.append((max(h, 1), max(w, 1))); h, w = (h + 1) // 2, (w + 1) // 2
        anchors, cls, reg = [], [], []
        for l, (fh, fw) in enumerate(shapes):
            sh, sw = Hp // fh, Wp // fw
            ys, xs = np.meshgrid(np.arange(fh) * sh, np.arange(fw) * sw, indexing="ij")
            shifts = np.stack([xs, ys, xs, ys], -1).reshape(-1, 1, 4).astype(np.float32)
            anchors.append((shifts + base[l][None]).reshape(-1, 4))
            c = (rs.randn(fh, fw, 9 * K) * 2.0 + bias).astype(np.float32)
            r = (rs.randn(fh, fw, 36) * 0.5).astype(np.float32)
            cls.append(c); reg.append(r)
            blob["r%d_cls%d" % (k, l)] = c; blob["r%d_reg%d" % (k, l)] = r
        anchors = np.concatenate(anchors).astype(np.float32)
        cl = np.concatenate([c.reshape(-1, K) for c in cls]); rg = np.concatenate([r.reshape(-1, 4) for r in reg])
        head = {"cls_logits": torch.from_numpy(cl)[None], "bbox_regression": torch.from_numpy(rg)[None]}
        det = rn.postprocess_detections(head, [torch.from_numpy(anchors)], [(Hr, Wr)])[0]
        blob.update({"r%d_K" % k: K, "r%d_sizes" % k: np.array([Hp, Wp, Hr, Wr]), "r%d_anchors" % k: anchors})
        for n in ("boxes", "scores", "labels", "scores_cls", "prob_max"):
            blob["r%d_out_%s" % (k, n)] = det[n].numpy()
        print("retina case", k, "dets", det["boxes"].shape[0], "labels", np.unique(det["labels"].numpy())[:8])
    blob["r_n"] = len(specs)
    return blob


if __name__ == "__main__":
    ref_harness.load_reference()
    fl = sys.modules["detection.frcnn_la"]; rc = sys.modules["detection.retinanet_cal"]
    bind(fl.box_ops); bind(rc.box_ops)
    blob = frcnn_cases(fl)
    blob.update(retina_cases(rc))
    path = os.path.join(OUT, "postprocess.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path))
