"""Golden vectors for the in-repo detector code the oracle so far only "followed" (VERDICT r3 item 2): TEST INFRASTRUCTURE.

Executed from /root/reference, as they lie, under the stub harness (torchvision 0.8.2 is not installed; the torchvision
PRIMITIVES these bodies call are bound to plain restatements -- SURVEY Appendix A -- exactly as oracle/make_golden_postprocess.py
does):

  * detection/frcnn_ll.py:207-238  concat_box_prediction_layers     (the (N, A*C, H, W) -> (N*HWA, C) layout)
  * detection/frcnn_ll.py:284-321  RegionProposalNetwork.filter_proposals
  * detection/frcnn_ll.py:323-374  RegionProposalNetwork.forward    (eval mode: head -> anchors -> concat -> decode -> filter)
        bound restatements: permute_and_flatten, _get_top_n_idx (a stable per-level top-k: torch.topk leaves the order of equal
        logits unspecified), BoxCoder.decode, AnchorGenerator, clip_boxes_to_image, remove_small_boxes, batched_nms.
        frcnn_ll.py:314-316 zero-pads the output when fewer than post_nms_top_n boxes survive -- a modification of the learning-loss
        baseline's copy that stock torchvision (what frcnn_la.py:199-203, the hot path, instantiates) does not have; case `pad`
        records it and the tests treat it as the documented deviation.
  * detection/retinanet_cal.py:57-62, :135-151, :225-241  RetinaNetHead / both sub-heads' forward -- pure torch.nn, run AS IS on
        random weights: pins the tower order and the (N, A*K, H, W) -> (N, HWA, K) layout
  * detection/retinanet_cal.py:323-374  RetinaNet.__init__ up to the head: the anchor sizes int(x * 2^(1/3)), int(x * 2^(2/3)) and
        nine anchors per location (:346-351)
  * detection/frcnn_la.py:292-315  GeneralizedRCNNTransform.postprocess / resize_boxes with resized != original sizes
        (python-float ratios applied to float32 tensors)

Writes tests/golden/{rpn_filter,retina_heads,resize_boxes}.npz.  Run in the build container:  python oracle/make_golden_rpn.py
"""
import math
import os
import sys
from argparse import Namespace
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness  # noqa: E402
import make_golden_postprocess as mgp  # noqa: E402  (BoxCoder, nms, batched_nms, clip, remove_small: the restated primitives)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


# ---- torchvision 0.8.2 pieces the RPN bodies call (Appendix A) ----
def permute_and_flatten(layer, N, A, C, H, W):
    layer = layer.view(N, -1, C, H, W)
    layer = layer.permute(0, 3, 4, 1, 2)
    return layer.reshape(N, -1, C)


def stable_top_n_idx(self, objectness, num_anchors_per_level):
    """RegionProposalNetwork._get_top_n_idx with a defined order among equal logits (value descending, index ascending)."""
    r, offset = [], 0
    for ob in objectness.split(num_anchors_per_level, 1):
        n = ob.shape[1]
        k = min(self.pre_nms_top_n(), n)
        idx = torch.stack([torch.from_numpy(np.argsort(-row.numpy().astype(np.float64), kind="stable")[:k].copy()) for row in ob])
        r.append(idx + offset)
        offset += n
    return torch.cat(r, dim=1)


def base_anchors(scales, ratios):
    scales = torch.as_tensor(scales, dtype=torch.float32); ratios = torch.as_tensor(ratios, dtype=torch.float32)
    h_ratios = torch.sqrt(ratios); w_ratios = 1 / h_ratios
    ws = (w_ratios[:, None] * scales[None, :]).view(-1); hs = (h_ratios[:, None] * scales[None, :]).view(-1)
    return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()


def grid_anchors(bases, level_hw, Hp, Wp):
    out = []
    for base, (fh, fw) in zip(bases, level_hw):
        sh, sw = Hp // fh, Wp // fw
        ys, xs = torch.meshgrid(torch.arange(fh, dtype=torch.float32) * sh, torch.arange(fw, dtype=torch.float32) * sw, indexing="ij")
        shifts = torch.stack((xs.reshape(-1), ys.reshape(-1), xs.reshape(-1), ys.reshape(-1)), dim=1)
        out.append((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4))
    return torch.cat(out)


def rpn_cases(fl):
    fl.permute_and_flatten = permute_and_flatten
    mgp.bind(fl.box_ops)
    rs = np.random.RandomState(31)
    A = 3
    sizes, ratios = [32, 64, 128, 256, 512], [0.5, 1.0, 2.0]
    bases = [base_anchors([s], ratios) for s in sizes]
    #        name      Hp   Wp   Hr   Wr   pre   post  logits                      delta gain
    specs = [("plain", 128, 160, 120, 150, 1000, 1000, lambda n: rs.randn(n) * 2.0, 0.5),
             ("ties", 128, 160, 128, 160, 1000, 300, lambda n: np.round(rs.randn(n) * 1.5) * 0.5, 0.5),
             ("clip", 96, 128, 40, 50, 1000, 200, lambda n: rs.randn(n) * 2.0, 0.3),        # most anchors end outside the 40 x 50 image: zero-size after the clip
             ("pre", 256, 320, 250, 300, 1000, 1000, lambda n: rs.randn(n) * 2.0, 0.6),     # level 0 holds 64 x 80 x 3 = 15 360 > 1000 anchors
             ("tiny", 64, 64, 64, 64, 1000, 50, lambda n: rs.randn(n) * 2.0, 0.4),          # coarsest levels are 1 x 1
             ("wide", 96, 352, 90, 345, 200, 100, lambda n: rs.randn(n) * 3.0, 1.2),        # large deltas: the log(1000 / 16) clamp
             ("pad", 64, 96, 64, 96, 1000, 5000, lambda n: rs.randn(n) * 2.0, 0.4)]         # fewer than post_n survive: frcnn_ll's zero padding
    blob = {"names": np.array([s[0] for s in specs]), "base": torch.stack(bases).numpy()}
    for name, Hp, Wp, Hr, Wr, pre, post, logit_fn, gain in specs:
        level_hw, h, w = [], Hp // 4, Wp // 4
        for _ in range(5):
            level_hw.append((max(h, 1), max(w, 1))); h, w = (h + 1) // 2, (w + 1) // 2
        obj, dl = [], []
        for fh, fw in level_hw:
            obj.append(torch.from_numpy(logit_fn(A * fh * fw).astype(np.float32).reshape(1, A, fh, fw)))
            d = (rs.randn(1, 4 * A, fh, fw) * gain).astype(np.float32)
            if name == "wide":
                d[0, 2::4] += 3.5
            dl.append(torch.from_numpy(d))
        anchors = grid_anchors(bases, level_hw, Hp, Wp)
        rpn = fl.RegionProposalNetwork.__new__(fl.RegionProposalNetwork)
        torch.nn.Module.__init__(rpn)
        rpn.training = False
        rpn._pre_nms_top_n = {"training": pre, "testing": pre}; rpn._post_nms_top_n = {"training": post, "testing": post}
        rpn.pre_nms_top_n = lambda: pre
        rpn.nms_thresh = 0.7; rpn.min_size = 1e-3
        rpn.box_coder = mgp.BoxCoder((1.0, 1.0, 1.0, 1.0))
        rpn._get_top_n_idx = lambda o, n, _r=rpn: stable_top_n_idx(_r, o, n)
        rpn.head = lambda feats: (obj, dl)
        rpn.anchor_generator = lambda images, feats: [anchors]
        # (1) the layout function on its own
        _bc, _br, bc, br = fl.concat_box_prediction_layers(obj, dl)
        # (2) forward (eval): head -> anchors -> concat -> decode -> filter_proposals
        images = Namespace(image_sizes=[(Hr, Wr)])
        boxes, losses = rpn.forward(images, OrderedDict((str(i), o) for i, o in enumerate(obj)))
        assert losses == {}
        # (3) filter_proposals directly, for the scores it returns beside the boxes
        props = rpn.box_coder.decode(br.detach(), [anchors]).view(1, -1, 4)
        fb, fs = rpn.filter_proposals(props, bc, images.image_sizes, [A * fh * fw for fh, fw in level_hw])
        assert torch.equal(fb[0], boxes[0])
        for l, (o, d) in enumerate(zip(obj, dl)):
            fh, fw = level_hw[l]
            head = torch.cat([o[0].permute(1, 2, 0), d[0].permute(1, 2, 0)], dim=2)        # NHWC: 3 logits then 12 deltas (a * 4 + j)
            blob["%s_head%d" % (name, l)] = head.numpy()
        blob["%s_cfg" % name] = np.array([Hp, Wp, Hr, Wr, pre, post])
        blob["%s_flat_logits" % name] = bc.numpy(); blob["%s_flat_deltas" % name] = br.numpy()
        blob["%s_boxes" % name] = boxes[0].numpy(); blob["%s_scores" % name] = fs[0].numpy()
        print("rpn case %-6s levels %s -> %d boxes%s" % (name, level_hw, boxes[0].shape[0], " (zero padded)" if float(boxes[0].abs().sum()) == 0 else ""))
    return blob


def retina_head_cases(rc):
    torch.manual_seed(5)
    blob = {}
    #        Cin  K   level sizes
    specs = [(16, 5, [(9, 12), (5, 6), (3, 3)]), (32, 21, [(6, 7), (3, 4), (2, 2), (1, 1)])]
    for k, (cin, K, hw) in enumerate(specs):
        head = rc.RetinaNetHead(cin, 9, K)
        for p in head.parameters():                                   # the reference initialises biases to constants: make every tensor informative
            with torch.no_grad():
                p.copy_(torch.randn_like(p) * (0.2 if p.dim() > 1 else 0.5))
        head.eval()
        feats = [torch.randn(2, cin, h, w) for h, w in hw]
        with torch.no_grad():
            out = head(feats)
        for n, t in head.state_dict().items():
            blob["h%d_w_%s" % (k, n)] = t.numpy()
        for l, f in enumerate(feats):
            blob["h%d_feat%d" % (k, l)] = f.numpy()
        blob["h%d_cls_logits" % k] = out["cls_logits"].numpy(); blob["h%d_bbox_regression" % k] = out["bbox_regression"].numpy()
        blob["h%d_cfg" % k] = np.array([cin, K, len(hw)])
        print("retina head case", k, "cls", tuple(out["cls_logits"].shape), "reg", tuple(out["bbox_regression"].shape))
    blob["n"] = len(specs)

    # RetinaNet.__init__: the default anchor generator's arguments and the head it builds (retinanet_cal.py:346-355)
    seen = {}

    class CapturingAnchorGenerator(torch.nn.Module):
        def __init__(self, sizes, aspect_ratios):
            super().__init__()
            seen["sizes"], seen["ratios"] = sizes, aspect_ratios

        def num_anchors_per_location(self):
            return [len(s) * len(a) for s, a in zip(seen["sizes"], seen["ratios"])]

    rc.AnchorGenerator = CapturingAnchorGenerator
    backbone = torch.nn.Identity(); backbone.out_channels = 8
    net = rc.RetinaNet(backbone, num_classes=4)
    blob["anchor_sizes"] = np.array(seen["sizes"], np.int64); blob["aspect_ratios"] = np.array(seen["ratios"], np.float64)
    blob["anchors_per_location"] = np.array(net.anchor_generator.num_anchors_per_location())
    blob["init_cls_out_channels"] = np.array(net.head.classification_head.cls_logits.out_channels)
    blob["init_thresholds"] = np.array([net.score_thresh, net.nms_thresh, net.detections_per_img])
    print("anchor sizes", seen["sizes"])
    return blob


def resize_cases(fl):
    rs = np.random.RandomState(41)
    tr = fl.GeneralizedRCNNTransform.__new__(fl.GeneralizedRCNNTransform)
    torch.nn.Module.__init__(tr)
    tr.training = False
    sizes = [((600, 800), (375, 500)), ((600, 901), (333, 500)), ((800, 1066), (480, 640)), ((602, 1000), (301, 500)), ((37, 53), (111, 160))]
    blob = {"n": len(sizes)}
    for k, (im_s, o_im_s) in enumerate(sizes):
        n = 40
        x0 = rs.rand(n) * im_s[1]; y0 = rs.rand(n) * im_s[0]
        b = np.stack([x0, y0, x0 + rs.rand(n) * 200, y0 + rs.rand(n) * 200], 1).astype(np.float32)
        p = (b + rs.randn(n, 4) * 3).astype(np.float32)
        res = tr.postprocess([{"boxes": torch.from_numpy(b), "props": torch.from_numpy(p), "scores": torch.zeros(n)}], [im_s], [o_im_s])
        blob["r%d_sizes" % k] = np.array(list(im_s) + list(o_im_s))
        blob["r%d_boxes" % k] = b; blob["r%d_props" % k] = p
        blob["r%d_out_boxes" % k] = res[0]["boxes"].numpy(); blob["r%d_out_props" % k] = res[0]["props"].numpy()
        direct = fl.resize_boxes(torch.from_numpy(b), im_s, o_im_s).numpy()
        assert np.array_equal(direct, blob["r%d_out_boxes" % k])
    print("resize_boxes cases", len(sizes))
    return blob


if __name__ == "__main__":
    ref_harness.load_reference()
    import importlib
    ll = importlib.import_module("detection.frcnn_ll")
    fl = sys.modules["detection.frcnn_la"]; rc = sys.modules["detection.retinanet_cal"]
    for name, blob in (("rpn_filter", rpn_cases(ll)), ("retina_heads", retina_head_cases(rc)), ("resize_boxes", resize_cases(fl))):
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print("wrote", path, os.path.getsize(path))
