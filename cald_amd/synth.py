"""Synthetic VOC/COCO-shaped pool images and seeded "pseudo-trained" detector weights.

There is no network here (no datasets, no checkpoints), and untrained heads give softmax ~ 1/21 <
box_score_thresh, i.e. no detections and a degenerate ranking.  This module
generates (a) uint8 RGB images with the size mix of VOC2012 / COCO2017 and (b) a torchvision-layout
state dict (SURVEY.md section 8b key layout) whose head statistics yield tens of detections per
view.  Only the generator is committed, never the weights.
"""
import numpy as np

VOC_SIZES = [((375, 500), 0.55), ((500, 375), 0.20), ((333, 500), 0.15), ((500, 334), 0.10)]   # (H, W), prob
COCO_SIZES = [((480, 640), 0.60), ((640, 480), 0.20), ((427, 640), 0.20)]

RESNET_LAYERS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}


def pool_sizes(n, kind="voc", seed=0):
    table = VOC_SIZES if kind == "voc" else COCO_SIZES
    rs = np.random.RandomState(seed)
    probs = np.array([p for _, p in table])
    idx = rs.choice(len(table), size=n, p=probs / probs.sum())
    return [table[i][0] for i in idx]


def synth_image(index, H, W):
    """Low-frequency noise + 3..8 filled rectangles / ellipses; seed = image index."""
    rs = np.random.RandomState(1000003 + index)
    gh, gw = H // 32 + 2, W // 32 + 2
    coarse = rs.rand(gh, gw, 3).astype(np.float32)
    ys = np.linspace(0, gh - 1.001, H); xs = np.linspace(0, gw - 1.001, W)
    y0 = ys.astype(int); x0 = xs.astype(int)
    fy = (ys - y0)[:, None, None].astype(np.float32); fx = (xs - x0)[None, :, None].astype(np.float32)
    img = (coarse[y0][:, x0] * (1 - fy) * (1 - fx) + coarse[y0 + 1][:, x0] * fy * (1 - fx) +
           coarse[y0][:, x0 + 1] * (1 - fy) * fx + coarse[y0 + 1][:, x0 + 1] * fy * fx)
    img = 0.25 + 0.5 * img
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(rs.randint(3, 9)):
        cy, cx = rs.rand() * H, rs.rand() * W
        hh, ww = (0.05 + 0.3 * rs.rand()) * H, (0.05 + 0.3 * rs.rand()) * W
        col = rs.rand(3).astype(np.float32)
        if rs.rand() < 0.5:
            mask = (np.abs(yy - cy) < hh / 2) & (np.abs(xx - cx) < ww / 2)
        else:
            mask = ((yy - cy) / (hh / 2)) ** 2 + ((xx - cx) / (ww / 2)) ** 2 < 1.0
        img[mask] = col
    img += (rs.rand(H, W, 3).astype(np.float32) - 0.5) * 0.04
    return np.ascontiguousarray(np.clip(img * 255.0, 0, 255).astype(np.uint8))


def make_pool(n, kind="voc", seed=0, scale=1.0):
    """List of uint8 HWC images.  scale < 1 shrinks the images (fast CPU tests)."""
    out = []
    for i, (H, W) in enumerate(pool_sizes(n, kind, seed)):
        out.append(synth_image(i, max(32, int(H * scale)), max(32, int(W * scale))))
    return out


def _he(rs, shape, gain=2.0):
    fan_in = int(np.prod(shape[1:]))
    return (rs.randn(*shape) * np.sqrt(gain / fan_in)).astype(np.float32)


def _bn(rs, sd, prefix, c, gamma=1.0):
    sd[prefix + ".weight"] = (gamma * (0.8 + 0.4 * rs.rand(c))).astype(np.float32)
    sd[prefix + ".bias"] = (0.1 * rs.randn(c)).astype(np.float32)
    sd[prefix + ".running_mean"] = (0.1 * rs.randn(c)).astype(np.float32)
    sd[prefix + ".running_var"] = (0.7 + 0.6 * rs.rand(c)).astype(np.float32)


def pseudo_trained_frcnn(num_classes=21, depth=50, seed=0, cls_gain=1.0, rpn_gain=1.0):
    """torchvision-layout Faster R-CNN ResNet-FPN state dict (numpy float32), seeded."""
    rs = np.random.RandomState(seed)
    sd = {}
    sd["backbone.body.conv1.weight"] = _he(rs, (64, 3, 7, 7))
    _bn(rs, sd, "backbone.body.bn1", 64)
    inplanes = 64
    for li, nb in enumerate(RESNET_LAYERS[depth]):
        planes = 64 * 2 ** li
        for bi in range(nb):
            p = "backbone.body.layer%d.%d" % (li + 1, bi)
            sd[p + ".conv1.weight"] = _he(rs, (planes, inplanes, 1, 1)); _bn(rs, sd, p + ".bn1", planes)
            sd[p + ".conv2.weight"] = _he(rs, (planes, planes, 3, 3)); _bn(rs, sd, p + ".bn2", planes)
            # A frozen BN with a synthetic running_var does not normalise: each residual branch adds variance PROPORTIONAL to its input's,
            # so the activation scale grows geometrically with the number of blocks of a stage.  gamma 0.5 keeps ResNet-50 at the O(10)
            # scale of a trained network; ResNet-101's 23-block stage would reach |x| ~ 600 and RPN logits of +-800 (round 4's
            # generator did), which no trained detector has -- the branch gain is therefore scaled so that a stage's total growth is
            # ResNet-50's whatever its depth (gamma^2 x blocks constant).  ResNet-50 weights are unchanged.
            g3 = 0.5 * float(np.sqrt(RESNET_LAYERS[50][li] / float(nb)))
            sd[p + ".conv3.weight"] = _he(rs, (planes * 4, planes, 1, 1)); _bn(rs, sd, p + ".bn3", planes * 4, gamma=g3)
            if bi == 0:
                sd[p + ".downsample.0.weight"] = _he(rs, (planes * 4, inplanes, 1, 1), gain=1.0)
                _bn(rs, sd, p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    for i, cin in enumerate([256, 512, 1024, 2048]):
        sd["backbone.fpn.inner_blocks.%d.weight" % i] = _he(rs, (256, cin, 1, 1), gain=1.0)
        sd["backbone.fpn.inner_blocks.%d.bias" % i] = (0.05 * rs.randn(256)).astype(np.float32)
        sd["backbone.fpn.layer_blocks.%d.weight" % i] = _he(rs, (256, 256, 3, 3), gain=1.0)
        sd["backbone.fpn.layer_blocks.%d.bias" % i] = (0.05 * rs.randn(256)).astype(np.float32)
    sd["rpn.head.conv.weight"] = _he(rs, (256, 256, 3, 3))
    sd["rpn.head.conv.bias"] = (0.05 * rs.randn(256)).astype(np.float32)
    wr = _he(rs, (3, 256, 1, 1), gain=1.0) * 1.5 * rpn_gain
    sd["rpn.head.cls_logits.weight"] = (wr - wr.mean(axis=1, keepdims=True)).astype(np.float32)
    sd["rpn.head.cls_logits.bias"] = (0.1 * rs.randn(3)).astype(np.float32)
    wb = _he(rs, (12, 256, 1, 1), gain=1.0) * 0.04
    sd["rpn.head.bbox_pred.weight"] = (wb - wb.mean(axis=1, keepdims=True)).astype(np.float32)
    sd["rpn.head.bbox_pred.bias"] = (0.02 * rs.randn(12)).astype(np.float32)
    sd["roi_heads.box_head.fc6.weight"] = _he(rs, (1024, 256 * 49))
    sd["roi_heads.box_head.fc6.bias"] = (0.05 * rs.randn(1024)).astype(np.float32)
    sd["roi_heads.box_head.fc7.weight"] = _he(rs, (1024, 1024))
    sd["roi_heads.box_head.fc7.bias"] = (0.05 * rs.randn(1024)).astype(np.float32)
    wc = _he(rs, (num_classes, 1024), gain=1.0) * 0.9 * cls_gain
    sd["roi_heads.box_predictor.cls_score.weight"] = (wc - wc.mean(axis=1, keepdims=True)).astype(np.float32)
    bc = (0.3 * rs.randn(num_classes)).astype(np.float32); bc[0] += 3.0
    sd["roi_heads.box_predictor.cls_score.bias"] = bc
    wd = _he(rs, (4 * num_classes, 1024), gain=1.0) * 0.25
    sd["roi_heads.box_predictor.bbox_pred.weight"] = (wd - wd.mean(axis=1, keepdims=True)).astype(np.float32)
    sd["roi_heads.box_predictor.bbox_pred.bias"] = (0.05 * rs.randn(4 * num_classes)).astype(np.float32)
    return sd


def pseudo_trained_retinanet(num_classes=21, depth=50, seed=0):
    """torchvision-layout RetinaNet ResNet-FPN state dict (detection/retinanet_cal.py key layout), seeded."""
    rs = np.random.RandomState(seed)
    full = pseudo_trained_frcnn(num_classes, depth, seed)
    sd = {k: v for k, v in full.items() if k.startswith("backbone.body.")}
    for i, cin in enumerate([512, 1024, 2048]):
        sd["backbone.fpn.inner_blocks.%d.weight" % i] = _he(rs, (256, cin, 1, 1), gain=1.0)
        sd["backbone.fpn.inner_blocks.%d.bias" % i] = (0.05 * rs.randn(256)).astype(np.float32)
        sd["backbone.fpn.layer_blocks.%d.weight" % i] = _he(rs, (256, 256, 3, 3), gain=1.0)
        sd["backbone.fpn.layer_blocks.%d.bias" % i] = (0.05 * rs.randn(256)).astype(np.float32)
    for p in ("p6", "p7"):
        sd["backbone.fpn.extra_blocks.%s.weight" % p] = _he(rs, (256, 256, 3, 3), gain=1.0)
        sd["backbone.fpn.extra_blocks.%s.bias" % p] = (0.05 * rs.randn(256)).astype(np.float32)
    for head in ("classification_head", "regression_head"):
        for i in range(4):
            sd["head.%s.conv.%d.weight" % (head, 2 * i)] = _he(rs, (256, 256, 3, 3))
            sd["head.%s.conv.%d.bias" % (head, 2 * i)] = (0.02 * rs.randn(256)).astype(np.float32)
    wc = _he(rs, (9 * num_classes, 256, 3, 3), gain=1.0) * 0.17
    sd["head.classification_head.cls_logits.weight"] = (wc - wc.mean(axis=(1, 2, 3), keepdims=True)).astype(np.float32)
    sd["head.classification_head.cls_logits.bias"] = (-5.0 + 0.3 * rs.randn(9 * num_classes)).astype(np.float32)
    wb = _he(rs, (36, 256, 3, 3), gain=1.0) * 0.05
    sd["head.regression_head.bbox_reg.weight"] = (wb - wb.mean(axis=(1, 2, 3), keepdims=True)).astype(np.float32)
    sd["head.regression_head.bbox_reg.bias"] = (0.02 * rs.randn(36)).astype(np.float32)
    return sd
