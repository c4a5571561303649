"""Training-step oracle (TEST INFRASTRUCTURE ONLY): the Faster R-CNN training forward of torchvision 0.8.2 restated with plain
torch CPU ops in float64 under torch autograd -- the checker for cald_amd/train.py.

What is restated, with the reference call sites (the arithmetic itself lives in torchvision 0.8.2, absent from the image:
"parity unpinned" for those pieces, as for the inference detector):
  * GeneralizedRCNN.forward in training mode            detection/frcnn_la.py:237-275
  * transform: normalize, resize (image and boxes), batch detection/frcnn_la.py:230-234 (GeneralizedRCNNTransform)
  * resnet_fpn_backbone(trainable_layers=3), FrozenBN    detection/frcnn_la.py:283
  * RegionProposalNetwork: assign_targets_to_anchors, BalancedPositiveNegativeSampler(256, 0.5), compute_loss
                                                         detection/frcnn_la.py:185-203 (in-repo copy detection/frcnn_ll.py:323-374)
  * RoIHeads.select_training_samples / fastrcnn_loss     detection/frcnn_la.py:104-128, :160-222
Proposals are an INPUT here (the region-proposal arithmetic is index work pinned by the inference oracle, cald_oracle.c); the
samplers' random choices are either drawn here (torch.randperm from the generator handed in) or handed in and validated.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as orc

MEAN = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float64).view(3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float64).view(3, 1, 1)


def box_iou(a, b):
    area1 = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); area2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[:, :2]); rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area1[:, None] + area2 - inter)


def matcher(iou, hi, lo, allow_low):
    """det_utils.Matcher.__call__ on a [gt, candidates] quality matrix (float32 IoUs)."""
    vals, matches = iou.max(dim=0)
    all_matches = matches.clone()
    matches[vals < lo] = -1
    matches[(vals >= lo) & (vals < hi)] = -2
    if allow_low:
        best_per_gt, _ = iou.max(dim=1)
        upd = torch.nonzero(iou == best_per_gt[:, None])[:, 1]
        matches[upd] = all_matches[upd]
    return matches


def encode(reference, proposals, w):
    ex_w = proposals[:, 2] - proposals[:, 0]; ex_h = proposals[:, 3] - proposals[:, 1]
    ex_cx = proposals[:, 0] + 0.5 * ex_w; ex_cy = proposals[:, 1] + 0.5 * ex_h
    gt_w = reference[:, 2] - reference[:, 0]; gt_h = reference[:, 3] - reference[:, 1]
    gt_cx = reference[:, 0] + 0.5 * gt_w; gt_cy = reference[:, 1] + 0.5 * gt_h
    return torch.stack([w[0] * (gt_cx - ex_cx) / ex_w, w[1] * (gt_cy - ex_cy) / ex_h, w[2] * torch.log(gt_w / ex_w), w[3] * torch.log(gt_h / ex_h)], dim=1)


def smooth_l1_sum(x, t, beta):
    n = (x - t).abs()
    return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta).sum()


def sample(labels_pos, labels_neg, batch, frac, gen):
    num_pos = min(int(batch * frac), labels_pos.numel())
    num_neg = min(batch - num_pos, labels_neg.numel())
    p = labels_pos[torch.randperm(labels_pos.numel(), generator=gen)[:num_pos]]
    n = labels_neg[torch.randperm(labels_neg.numel(), generator=gen)[:num_neg]]
    return p, n


def roi_align(feats, rois_img, rois_box):
    """MultiScaleRoIAlign(['0','1','2','3'], 7, 2), aligned=False, differentiable wrt feats (pure torch gathers)."""
    out = []
    for n, box in zip(rois_img.tolist(), rois_box):
        b32 = box.float()
        area = (b32[2] - b32[0]) * (b32[3] - b32[1])
        k = math.floor(4.0 + math.log2(max(float(torch.sqrt(area)), 1e-30) / 224.0) + 1e-6)
        l = int(min(max(k, 2), 5)) - 2
        f = feats[l][n]                                       # [C, H, W]
        Hf, Wf = f.shape[1], f.shape[2]
        s = 1.0 / (4 << l)
        x1, y1, x2, y2 = [float(v) * s for v in b32]
        rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
        bw, bh = rw / 7.0, rh / 7.0
        def samples(start, bin_, size):
            t = start + torch.arange(7, dtype=torch.float64)[:, None] * bin_ + (torch.arange(2, dtype=torch.float64)[None, :] + 0.5) * bin_ / 2.0
            t = t.reshape(-1)
            valid = ~((t < -1.0) | (t > size))
            tt = t.clamp(min=0)
            lo = tt.floor().long()
            over = lo >= size - 1
            lo = torch.where(over, torch.full_like(lo, size - 1), lo)
            hi = torch.where(over, lo, lo + 1)
            tt = torch.where(over, lo.double(), tt)
            frac = tt - lo.double()
            return lo, hi, frac, valid
        ylo, yhi, ly, vy = samples(y1, bh, Hf)
        xlo, xhi, lx, vx = samples(x1, bw, Wf)
        hy, hx = 1 - ly, 1 - lx
        v = (f[:, ylo][:, :, xlo] * (hy[:, None] * hx[None, :]) + f[:, ylo][:, :, xhi] * (hy[:, None] * lx[None, :])
             + f[:, yhi][:, :, xlo] * (ly[:, None] * hx[None, :]) + f[:, yhi][:, :, xhi] * (ly[:, None] * lx[None, :]))
        v = v * (vy[:, None] & vx[None, :]).double()
        out.append(v.reshape(-1, 7, 2, 7, 2).mean(dim=(2, 4)))  # [C, 7, 7]
    return torch.stack(out)


class TorchTrainFRCNN(object):
    def __init__(self, sd, num_classes, depth=50, min_size=600, max_size=1000, trainable_layers=3):
        self.C, self.min_size, self.max_size, self.depth = num_classes, min_size, max_size, depth
        frozen_layers = ["layer4", "layer3", "layer2", "layer1", "conv1"][trainable_layers:]
        self.p = {}
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                continue
            t = (v.detach().cpu() if hasattr(v, "detach") else torch.from_numpy(np.asarray(v))).double().clone()
            is_bn = ".bn" in k or "downsample.1" in k or k.startswith("backbone.body.bn1")
            frozen = is_bn or (k.startswith("backbone.body.") and any(k.startswith("backbone.body." + f) for f in frozen_layers))
            self.p[k] = t.requires_grad_(not frozen)

    masks = None   # optional {name: bool tensor}: ReLU decisions taken from the implementation under test (see relu())

    def relu(self, z, key):
        """ReLU.  The gradient of a float32 implementation and of this float64 restatement can only be compared where both took the
        same branch at every ReLU; a pre-activation within float32 noise of zero may fall on either side.  With ``masks`` set, the
        decision is the checked implementation's (z * mask: forward values move by float32 noise, the graph becomes smooth)."""
        if self.masks is not None and key in self.masks:
            return z * self.masks[key].to(z.dtype)
        return F.relu(z)

    def trainable(self):
        return {k: v for k, v in self.p.items() if v.requires_grad}

    def _bn(self, x, prefix):
        p = self.p
        scale = p[prefix + ".weight"] * (p[prefix + ".running_var"] + 1e-5).rsqrt()
        shift = p[prefix + ".bias"] - p[prefix + ".running_mean"] * scale
        return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)

    def body(self, x):
        p = self.p
        x = F.relu(self._bn(F.conv2d(x, p["backbone.body.conv1.weight"], stride=2, padding=3), "backbone.body.bn1"))
        x = F.max_pool2d(x, 3, 2, 1)
        feats = []
        for li, nb in enumerate({50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[self.depth]):
            for b in range(nb):
                pre = "backbone.body.layer%d.%d" % (li + 1, b)
                s = 2 if (b == 0 and li > 0) else 1
                idt = x
                key = "layer%d.%d" % (li + 1, b)
                o = self.relu(self._bn(F.conv2d(x, p[pre + ".conv1.weight"]), pre + ".bn1"), key + ".a1")
                o = self.relu(self._bn(F.conv2d(o, p[pre + ".conv2.weight"], stride=s, padding=1), pre + ".bn2"), key + ".a2")
                o = self._bn(F.conv2d(o, p[pre + ".conv3.weight"]), pre + ".bn3")
                if b == 0:
                    idt = self._bn(F.conv2d(x, p[pre + ".downsample.0.weight"], stride=s), pre + ".downsample.1")
                x = self.relu(o + idt, key + ".out")
            feats.append(x)
        return feats

    def backbone(self, x):
        p = self.p
        feats = self.body(x)
        inner = [None] * 4
        inner[3] = F.conv2d(feats[3], p["backbone.fpn.inner_blocks.3.weight"], p["backbone.fpn.inner_blocks.3.bias"])
        for i in (2, 1, 0):
            lat = F.conv2d(feats[i], p["backbone.fpn.inner_blocks.%d.weight" % i], p["backbone.fpn.inner_blocks.%d.bias" % i])
            inner[i] = lat + F.interpolate(inner[i + 1], size=lat.shape[-2:], mode="nearest")
        P = [F.conv2d(inner[i], p["backbone.fpn.layer_blocks.%d.weight" % i], p["backbone.fpn.layer_blocks.%d.bias" % i], padding=1) for i in range(4)]
        P.append(F.max_pool2d(P[3], 1, 2, 0))
        return P

    def batch(self, images, targets):
        """GeneralizedRCNNTransform: normalized, resized, zero-padded batch (float64) and the resized ground-truth boxes (float32)."""
        N = len(images)
        sizes = [orc.transform_size(int(im.shape[1]), int(im.shape[2]), self.min_size, self.max_size) for im in images]
        Hp, Wp = max(s[2] for s in sizes), max(s[3] for s in sizes)
        batch = torch.zeros(N, 3, Hp, Wp, dtype=torch.float64)
        gts = []
        for i, (im, s, t) in enumerate(zip(images, sizes, targets)):
            x = (im.double() - MEAN) / STD
            x = F.interpolate(x[None], size=(s[0], s[1]), mode="bilinear", align_corners=False)[0]
            batch[i, :, :s[0], :s[1]] = x
            b = t["boxes"].float().reshape(-1, 4)
            rh = torch.tensor(s[0], dtype=torch.float32) / torch.tensor(im.shape[1], dtype=torch.float32)
            rw = torch.tensor(s[1], dtype=torch.float32) / torch.tensor(im.shape[2], dtype=torch.float32)
            gts.append(torch.stack([b[:, 0] * rw, b[:, 1] * rh, b[:, 2] * rw, b[:, 3] * rh], dim=1))
        return batch, gts, Hp, Wp

    def losses(self, images, targets, proposals, gen, cfg=None, samples=None):
        """images: list of float CHW in [0, 1]; targets: dicts with boxes (original image coordinates) / labels; proposals: list
        of [n_i, 4] tensors in resized-image coordinates (what the RPN produced).  samples: optional {"rpn": [(pos, neg) index
        arrays per image], "box": [...]} -- the samplers' random choices as made by the implementation under test; each is checked
        to be a legal draw of BalancedPositiveNegativeSampler (right sizes, drawn from this restatement's own positive / negative
        sets).  Without it the draws are torch.randperm(n, generator=gen)[:k].  Returns (loss dict, records)."""
        def draw(kind, i, pos, neg, batch, frac):
            if samples is None:
                return sample(pos, neg, batch, frac, gen)
            sp, sn = [torch.as_tensor(np.asarray(v), dtype=torch.int64) for v in samples[kind][i]]
            num_pos = min(int(batch * frac), pos.numel())
            assert len(sp) == num_pos and len(sn) == min(batch - num_pos, neg.numel()), "sampler sizes"
            assert len(set(sp.tolist())) == len(sp) and len(set(sn.tolist())) == len(sn), "sampled twice"
            assert set(sp.tolist()) <= set(pos.tolist()) and set(sn.tolist()) <= set(neg.tolist()), "sampled outside the candidate sets"
            return sp, sn
        cfg = dict(dict(rpn_fg=0.7, rpn_bg=0.3, rpn_batch=256, rpn_pos=0.5, box_fg=0.5, box_bg=0.5, box_batch=512, box_pos=0.25, w=(10.0, 10.0, 5.0, 5.0)),
                   **(cfg or {}))
        p, N = self.p, len(images)
        batch, gts, Hp, Wp = self.batch(images, targets)
        P = self.backbone(batch)
        # RPN head, flattened in torchvision's order (image, level, y, x, anchor)
        obj, deltas = [], []
        for l, f in enumerate(P):
            t = self.relu(F.conv2d(f, p["rpn.head.conv.weight"], p["rpn.head.conv.bias"], padding=1), "rpn.%d" % l)
            o = F.conv2d(t, p["rpn.head.cls_logits.weight"], p["rpn.head.cls_logits.bias"])
            d = F.conv2d(t, p["rpn.head.bbox_pred.weight"], p["rpn.head.bbox_pred.bias"])
            obj.append(o.permute(0, 2, 3, 1).reshape(N, -1))
            deltas.append(d.permute(0, 2, 3, 1).reshape(N, -1, 4))
        obj = torch.cat(obj, dim=1); deltas = torch.cat(deltas, dim=1)
        # anchors (float32, as AnchorGenerator builds them)
        anchors = []
        for l, f in enumerate(P):
            Hl, Wl = f.shape[-2:]
            base = torch.from_numpy(orc.base_anchors([32.0 * 2 ** l], [0.5, 1.0, 2.0])).float().reshape(-1, 4)
            sy, sx = Hp // Hl, Wp // Wl
            ys, xs = torch.meshgrid(torch.arange(Hl) * sy, torch.arange(Wl) * sx, indexing="ij")
            shifts = torch.stack([xs, ys, xs, ys], dim=-1).reshape(-1, 1, 4).float()
            anchors.append((shifts + base[None]).reshape(-1, 4))
        anchors = torch.cat(anchors)
        rec = dict(anchors=anchors)
        # ---- RPN loss ----
        pos_all, neg_all, tgt_all = [], [], []
        A = anchors.shape[0]
        for i in range(N):
            if gts[i].shape[0] == 0:
                m = torch.full((A,), -1, dtype=torch.int64)
            else:
                m = matcher(box_iou(gts[i], anchors), cfg["rpn_fg"], cfg["rpn_bg"], True)
            pos, neg = torch.nonzero(m >= 0).squeeze(1), torch.nonzero(m == -1).squeeze(1)
            sp, sn = draw("rpn", i, pos, neg, cfg["rpn_batch"], cfg["rpn_pos"])
            sp, sn = sp.sort().values, sn.sort().values
            pos_all.append(i * A + sp); neg_all.append(i * A + sn)
            tgt_all.append(encode(gts[i][m[sp]].double(), anchors[sp].double(), (1.0, 1.0, 1.0, 1.0)) if len(sp) else torch.zeros(0, 4, dtype=torch.float64))
        pos_all, neg_all = torch.cat(pos_all), torch.cat(neg_all)
        sampled = torch.cat([pos_all, neg_all])
        lab = torch.cat([torch.ones(len(pos_all)), torch.zeros(len(neg_all))]).double()
        loss_obj = F.binary_cross_entropy_with_logits(obj.reshape(-1)[sampled], lab)
        loss_rpn_box = smooth_l1_sum(deltas.reshape(-1, 4)[pos_all], torch.cat(tgt_all), 1.0 / 9) / sampled.numel()
        rec.update(rpn_pos=pos_all, rpn_neg=neg_all)
        # ---- RoI heads ----
        r_img, r_box, r_lab, r_tgt = [], [], [], []
        for i in range(N):
            pr = torch.cat([proposals[i].float(), gts[i]]) if gts[i].shape[0] else proposals[i].float()
            if gts[i].shape[0] == 0:
                m = torch.full((pr.shape[0],), -1, dtype=torch.int64); labels = torch.zeros(pr.shape[0], dtype=torch.int64)
            else:
                m = matcher(box_iou(gts[i], pr), cfg["box_fg"], cfg["box_bg"], False)
                labels = targets[i]["labels"].long()[m.clamp(min=0)].clone()
                labels[m == -1] = 0
                labels[m == -2] = -1
            pos, neg = torch.nonzero(labels >= 1).squeeze(1), torch.nonzero(labels == 0).squeeze(1)
            sp, sn = draw("box", i, pos, neg, cfg["box_batch"], cfg["box_pos"])
            keep = torch.cat([sp, sn]).sort().values
            r_img.append(torch.full((len(keep),), i, dtype=torch.int64)); r_box.append(pr[keep]); r_lab.append(labels[keep])
            mg = gts[i][m[keep].clamp(min=0)] if gts[i].shape[0] else torch.zeros(len(keep), 4)
            r_tgt.append(encode(mg.double(), pr[keep].double(), cfg["w"]))
        r_img, r_box, r_lab, r_tgt = torch.cat(r_img), torch.cat(r_box), torch.cat(r_lab), torch.cat(r_tgt)
        feat = roi_align(P[:4], r_img, r_box)
        h = self.relu(F.linear(feat.flatten(1), p["roi_heads.box_head.fc6.weight"], p["roi_heads.box_head.fc6.bias"]), "fc6")
        h = self.relu(F.linear(h, p["roi_heads.box_head.fc7.weight"], p["roi_heads.box_head.fc7.bias"]), "fc7")
        logits = F.linear(h, p["roi_heads.box_predictor.cls_score.weight"], p["roi_heads.box_predictor.cls_score.bias"])
        breg = F.linear(h, p["roi_heads.box_predictor.bbox_pred.weight"], p["roi_heads.box_predictor.bbox_pred.bias"])
        loss_cls = F.cross_entropy(logits, r_lab)
        posr = torch.nonzero(r_lab > 0).squeeze(1)
        loss_box = smooth_l1_sum(breg.reshape(len(r_lab), -1, 4)[posr, r_lab[posr]], r_tgt[posr], 1.0 / 9) / r_lab.numel()
        rec.update(roi_img=r_img, roi_box=r_box, roi_labels=r_lab, P=P, logits=logits)
        return {"loss_classifier": loss_cls, "loss_box_reg": loss_box, "loss_objectness": loss_obj, "loss_rpn_box_reg": loss_rpn_box}, rec


def sigmoid_focal_loss_sum(x, t, alpha=0.25, gamma=2.0):
    """torchvision.ops.sigmoid_focal_loss(reduction='sum')."""
    p = torch.sigmoid(x)
    ce = F.binary_cross_entropy_with_logits(x, t, reduction="none")
    p_t = p * t + (1 - p) * (1 - t)
    loss = ce * (1 - p_t) ** gamma
    return ((alpha * t + (1 - alpha) * (1 - t)) * loss).sum()


class TorchTrainRetinaNet(TorchTrainFRCNN):
    """detection/retinanet_cal.py RetinaNet in training mode (forward :545-564, compute_loss :389-400, head losses :100-133 and
    :185-221), float64 torch autograd.  The loss code is the reference's own (in-repo); backbone / anchors / matcher are torchvision's."""

    def losses(self, images, targets, cfg=None):
        p, N, K = self.p, len(images), self.C
        batch, gts, Hp, Wp = self.batch(images, targets)
        feats = self.body(batch)
        inner = [None] * 3
        inner[2] = F.conv2d(feats[3], p["backbone.fpn.inner_blocks.2.weight"], p["backbone.fpn.inner_blocks.2.bias"])
        for i in (1, 0):
            lat = F.conv2d(feats[i + 1], p["backbone.fpn.inner_blocks.%d.weight" % i], p["backbone.fpn.inner_blocks.%d.bias" % i])
            inner[i] = lat + F.interpolate(inner[i + 1], size=lat.shape[-2:], mode="nearest")
        P = [F.conv2d(inner[i], p["backbone.fpn.layer_blocks.%d.weight" % i], p["backbone.fpn.layer_blocks.%d.bias" % i], padding=1) for i in range(3)]
        p6 = F.conv2d(P[2], p["backbone.fpn.extra_blocks.p6.weight"], p["backbone.fpn.extra_blocks.p6.bias"], stride=2, padding=1)
        p7 = F.conv2d(self.relu(p6, "p6"), p["backbone.fpn.extra_blocks.p7.weight"], p["backbone.fpn.extra_blocks.p7.bias"], stride=2, padding=1)
        P += [p6, p7]
        cls, reg = [], []
        for l, f in enumerate(P):
            for name, head, outs, last in (("cls", "classification_head", cls, "cls_logits"), ("reg", "regression_head", reg, "bbox_reg")):
                t = f
                for j in range(4):
                    t = self.relu(F.conv2d(t, p["head.%s.conv.%d.weight" % (head, 2 * j)], p["head.%s.conv.%d.bias" % (head, 2 * j)], padding=1), "%s.%d.%d" % (name, l, j))
                o = F.conv2d(t, p["head.%s.%s.weight" % (head, last)], p["head.%s.%s.bias" % (head, last)], padding=1)
                Nn, _, H, W = o.shape
                c = K if name == "cls" else 4
                outs.append(o.view(Nn, -1, c, H, W).permute(0, 3, 4, 1, 2).reshape(Nn, -1, c))
        cls, reg = torch.cat(cls, dim=1), torch.cat(reg, dim=1)
        anchors = []
        for l, f in enumerate(P):
            Hl, Wl = f.shape[-2:]
            x = 32 * 2 ** l
            base = torch.from_numpy(orc.base_anchors([float(x), float(int(x * 2 ** (1.0 / 3))), float(int(x * 2 ** (2.0 / 3)))], [0.5, 1.0, 2.0])).float().reshape(-1, 4)
            ys, xs = torch.meshgrid(torch.arange(Hl) * (Hp // Hl), torch.arange(Wl) * (Wp // Wl), indexing="ij")
            anchors.append((torch.stack([xs, ys, xs, ys], dim=-1).reshape(-1, 1, 4).float() + base[None]).reshape(-1, 4))
        anchors = torch.cat(anchors)
        cfg = dict(dict(fg=0.5, bg=0.4), **(cfg or {}))
        lc, lr, matched = retina_losses(cls, reg, anchors, gts, [t["labels"] for t in targets], cfg["fg"], cfg["bg"])
        rec = dict(anchors=anchors, matched=torch.stack(matched))
        return {"classification": lc, "bbox_regression": lr}, rec


def retina_losses(cls, reg, anchors, gts, labels, fg_thr=0.5, bg_thr=0.4):
    """detection/retinanet_cal.py:389-400 (matcher driver), :100-133 (classification) and :185-223 (box regression), restated.
    cls [N][A][K], reg [N][A][4], anchors [A][4], per-image gt boxes / labels.  Pinned to the reference's own code by
    tests/golden/train_losses.npz (oracle/make_golden_train_losses.py).  Returns (classification, bbox_regression, matched)."""
    N = len(gts)
    lc, lr, matched = [], [], []
    for i in range(N):
        m = matcher(box_iou(gts[i], anchors), fg_thr, bg_thr, True)
        matched.append(m)
        fg = m >= 0
        nfg = max(1, int(fg.sum()))
        tgt = torch.zeros_like(cls[i])
        tgt[fg, labels[i].long()[m[fg]]] = 1.0
        valid = m != -2
        lc.append(sigmoid_focal_loss_sum(cls[i][valid], tgt[valid]) / nfg)
        t_reg = encode(gts[i][m.clamp(min=0)][fg].double(), anchors[fg].double(), (1.0, 1.0, 1.0, 1.0))
        lr.append((reg[i][fg] - t_reg).abs().sum() / nfg)
    return sum(lc) / N, sum(lr) / max(1, N), matched
