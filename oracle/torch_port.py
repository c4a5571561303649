"""PyTorch-CPU port of the reference sweep (TEST INFRASTRUCTURE: cpu_baseline + float cross-check).

Structurally identical to the reference's path on a CPU-only box: batch-1 sequential forwards of the
reference view and every augmented view (cald_train.py:107, :185-186) with float32 torch CPU ops for
everything that carries FLOPs (conv / linear / pooling / interpolate -- the oneDNN kernels the reference
would run), and the per-reference-box Python scoring loop with scipy.stats.entropy
(cald_train.py:202-222).  The integer-heavy glue that torchvision implements in C++ (top-k/NMS,
RoIAlign, post-processing) is taken from the C oracle, which is what pins its results.

It is (a) timed by bench.py as `cpu_baseline` (kind "port") and (b) used by tests/ as the plain
torch fp32 reference the oracle's convolution chain is cross-checked against.
"""
import numpy as np
import scipy.stats
import torch
import torch.nn.functional as F

from . import oracle as orc

MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class TorchFRCNN:
    def __init__(self, sd, num_classes, depth=50, min_size=600, max_size=1000):
        self.sd = {k: _t(v.detach().cpu().numpy() if hasattr(v, "detach") else v) for k, v in sd.items()}
        self.C = num_classes
        self.depth = depth
        self.min_size, self.max_size = min_size, max_size
        self.anchors = np.stack([orc.base_anchors([s], [0.5, 1.0, 2.0]) for s in (32, 64, 128, 256, 512)])

    def _bn(self, x, p):
        sd = self.sd
        scale = sd[p + ".weight"] * (sd[p + ".running_var"] + 1e-5).rsqrt()
        shift = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
        return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)

    def transform(self, img_u8, flip=False, rects=None):
        x = torch.from_numpy(np.ascontiguousarray(img_u8)).permute(2, 0, 1).float().div(255)   # to_tensor
        if flip:
            x = x.flip(-1)
        if rects is not None:
            for (l, t, r, b) in rects:
                x[:, t:b, l:r] = 0.0
        x = (x - MEAN) / STD
        H, W = x.shape[-2:]
        Hr, Wr, Hp, Wp = orc.transform_size(H, W, self.min_size, self.max_size)
        x = F.interpolate(x[None], size=(Hr, Wr), mode="bilinear", align_corners=False)
        out = torch.zeros((1, 3, Hp, Wp))
        out[:, :, :Hr, :Wr] = x
        return out, (Hr, Wr, Hp, Wp)

    def backbone(self, x):
        sd = self.sd
        x = F.relu(self._bn(F.conv2d(x, sd["backbone.body.conv1.weight"], stride=2, padding=3), "backbone.body.bn1"))
        x = F.max_pool2d(x, 3, 2, 1)
        feats = []
        for li, nb in enumerate(orc.RESNET_LAYERS[self.depth]):
            for bi in range(nb):
                p = "backbone.body.layer%d.%d" % (li + 1, bi)
                stride = 2 if (bi == 0 and li > 0) else 1
                idn = x
                if p + ".downsample.0.weight" in sd:
                    idn = self._bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), p + ".downsample.1")
                o = F.relu(self._bn(F.conv2d(x, sd[p + ".conv1.weight"]), p + ".bn1"))
                o = F.relu(self._bn(F.conv2d(o, sd[p + ".conv2.weight"], stride=stride, padding=1), p + ".bn2"))
                x = F.relu(self._bn(F.conv2d(o, sd[p + ".conv3.weight"]), p + ".bn3") + idn)
            feats.append(x)
        inner = [None] * 4
        inner[3] = F.conv2d(feats[3], sd["backbone.fpn.inner_blocks.3.weight"], sd["backbone.fpn.inner_blocks.3.bias"])
        for i in (2, 1, 0):
            lat = F.conv2d(feats[i], sd["backbone.fpn.inner_blocks.%d.weight" % i], sd["backbone.fpn.inner_blocks.%d.bias" % i])
            inner[i] = lat + F.interpolate(inner[i + 1], size=lat.shape[-2:], mode="nearest")
        outs = [F.conv2d(inner[i], sd["backbone.fpn.layer_blocks.%d.weight" % i], sd["backbone.fpn.layer_blocks.%d.bias" % i], padding=1)
                for i in range(4)]
        outs.append(F.max_pool2d(outs[3], 1, 2, 0))
        return outs, feats

    def forward(self, img_u8, flip=False, rects=None, keep=None):
        sd = self.sd
        H, W, _ = img_u8.shape
        with torch.no_grad():
            x, (Hr, Wr, Hp, Wp) = self.transform(img_u8, flip, rects)
            feats, cfeats = self.backbone(x)
            heads = []
            for f in feats:
                t = F.relu(F.conv2d(f, sd["rpn.head.conv.weight"], sd["rpn.head.conv.bias"], padding=1))
                cl = F.conv2d(t, sd["rpn.head.cls_logits.weight"], sd["rpn.head.cls_logits.bias"])
                bb = F.conv2d(t, sd["rpn.head.bbox_pred.weight"], sd["rpn.head.bbox_pred.bias"])
                heads.append(torch.cat([cl, bb], 1)[0].permute(1, 2, 0).contiguous().numpy())
            if keep is not None:
                keep["input"] = x[0].permute(1, 2, 0).numpy(); keep["fpn"] = [f[0].permute(1, 2, 0).contiguous().numpy() for f in feats]
                keep["C"] = [f[0].permute(1, 2, 0).contiguous().numpy() for f in cfeats]; keep["rpn_head"] = heads
            props, _ = orc.rpn_proposals(heads, self.anchors, Hp, Wp, Hr, Wr)
            Cn = self.C
            if props.shape[0] == 0:
                z = np.zeros
                return dict(boxes=z((0, 4), np.float32), scores=z(0, np.float32), labels=z(0, np.int64), props=z((0, 4), np.float32),
                            prob_max=z(0, np.float32), scores_cls=z((0, Cn), np.float32))
            roi = orc.roi_align([f[0].permute(1, 2, 0).contiguous().numpy() for f in feats[:4]], props)     # [R][49][256]
            r = torch.from_numpy(roi).permute(0, 2, 1).reshape(roi.shape[0], -1)                          # (c, bin) flatten
            h = F.relu(F.linear(r, sd["roi_heads.box_head.fc6.weight"], sd["roi_heads.box_head.fc6.bias"]))
            h = F.relu(F.linear(h, sd["roi_heads.box_head.fc7.weight"], sd["roi_heads.box_head.fc7.bias"]))
            logits = F.linear(h, sd["roi_heads.box_predictor.cls_score.weight"], sd["roi_heads.box_predictor.cls_score.bias"]).numpy()
            deltas = F.linear(h, sd["roi_heads.box_predictor.bbox_pred.weight"], sd["roi_heads.box_predictor.bbox_pred.bias"]).numpy()
        return orc.frcnn_postprocess(logits, deltas, props, Hr, Wr, H, W)


def score_image_python(ref, aug_outs, aug_boxes, num_cls, bp):
    """The reference's own scoring loop shape (cald_train.py:187-228): python loop over reference boxes,
    torch ops per box, scipy.stats.entropy on float32 numpy vectors."""
    def cls_corr_of(o):
        cc = [0] * (num_cls - 1)
        for s, l in zip(o["scores"], o["labels"]):
            cc[int(l) - 1] = max(cc[int(l) - 1], float(s))
        return cc
    cls_corrs = [cls_corr_of(ref)]
    if ref["boxes"].shape[0] == 0:
        return 0.0, np.mean(cls_corrs, axis=0)
    consistency_aug = []
    for out, aug_box in zip(aug_outs, aug_boxes):
        cls_corrs.append(cls_corr_of(out))
        boxes = torch.from_numpy(out["boxes"]); scores_cls = out["scores_cls"]; pm = torch.from_numpy(out["prob_max"])
        if len(boxes) == 0:
            consistency_aug.append(0.0)
            continue
        consistency_img = 1.0
        for ab, ref_score_cls, ref_pm in zip(torch.from_numpy(np.asarray(aug_box, np.float32)), ref["scores_cls"], torch.from_numpy(ref["prob_max"])):
            width = torch.min(ab[2], boxes[:, 2]) - torch.max(ab[0], boxes[:, 0])
            height = torch.min(ab[3], boxes[:, 3]) - torch.max(ab[1], boxes[:, 1])
            Aarea = (ab[2] - ab[0]) * (ab[3] - ab[1])
            Barea = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
            iner = width * height
            iou = iner / (Aarea + Barea - iner)
            iou[width < 0] = 0.0
            iou[height < 0] = 0.0
            j = int(torch.argmax(iou))
            p, q = ref_score_cls, scores_cls[j]
            m = (p + q) / 2
            js = 0.5 * scipy.stats.entropy(p, m) + 0.5 * scipy.stats.entropy(q, m)
            if js < 0:
                js = 0
            consistency_img = min(consistency_img, torch.abs(torch.max(iou) + 0.5 * (1 - js) * (ref_pm + pm[j]) - bp).item())
        consistency_aug.append(np.mean(consistency_img))
    return float(np.mean(consistency_aug)), np.mean(np.array(cls_corrs), axis=0)


def get_uncertainty(model, images, augs, num_cls, bp=1.3, base_seed=0, positions=None):
    cons, cls = [], []
    for pos, img in enumerate(images):
        gpos = pos if positions is None else positions[pos]
        ref = orc.subsample_ref(model.forward(img))
        if ref["boxes"].shape[0] == 0:
            c, cc = score_image_python(ref, [], [], num_cls, bp)
        else:
            views = orc.build_views(img, augs, ref, orc.image_seed(base_seed, gpos))
            outs = [model.forward(v[0], v[1], v[2]) for v in views]   # (the port is only timed on flip / cut_out / resize)
            c, cc = score_image_python(ref, outs, [v[3] for v in views], num_cls, bp)
        cons.append(c); cls.append(cc)
    return cons, cls
