"""Golden vectors for the Faster R-CNN training losses whose code lives in the reference repo: TEST INFRASTRUCTURE.

The hot path's detector (detection/frcnn_la.py) takes its training losses from torchvision 0.8.2 (`fastrcnn_loss`, the RPN's
`compute_loss`), which is not installed.  The reference tree does hold COPIES of both, made for its learning-loss baseline:
  * detection/frcnn_ll.py:28-63    _fastrcnn_loss                         (per image: cross_entropy, class-specific box deltas
                                                                           br[pos, label_pos], sum / label.numel())
  * detection/frcnn_ll.py:245-281  RegionProposalNetwork._compute_loss    (per image: sum over sampled positives / #sampled,
                                                                           binary_cross_entropy_with_logits over sampled)
They differ from stock torchvision in two documented ways: they return one loss PER IMAGE (stock: one over the batch -- identical
for a single image, which is what every case below is), and they use F.smooth_l1_loss's default beta = 1 / F.l1_loss where stock
uses smooth L1 with beta = 1/9.  What they pin, executed as they lie: the gather of class-specific deltas, the positive-only box
loss, the normalisers, the objectness labels.  The device kernels take beta as a parameter (1/9 on the hot path, 0 = L1 for
RetinaNet), so the tests run them at the copies' beta.  Writes tests/golden/frcnn_losses.npz:  python oracle/make_golden_frcnn_losses.py
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    ref_harness.load_reference()
    ll = importlib.import_module("detection.frcnn_ll")
    rs = np.random.RandomState(77)
    blob = {}
    # ---- box head: (R rows, C classes, positives) ----
    specs = [(64, 21, 16), (512, 21, 128), (128, 91, 5), (32, 21, 0), (96, 4, 96)]
    for k, (R, C, npos) in enumerate(specs):
        logits = (rs.randn(R, C) * 2.0).astype(np.float32)
        deltas = (rs.randn(R, 4 * C) * 0.7).astype(np.float32)
        deltas[::7] *= 6.0                                           # |x| > 1: the linear branch of smooth L1
        labels = np.zeros(R, np.int64)
        labels[rs.permutation(R)[:npos]] = rs.randint(1, C, npos)
        tgt = (rs.randn(R, 4) * 0.8).astype(np.float32)
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            cl, bl = ll._fastrcnn_loss(torch.from_numpy(logits).to(dt), torch.from_numpy(deltas).to(dt), [torch.from_numpy(labels)], [torch.from_numpy(tgt).to(dt)])
            assert cl.shape == (1,) and bl.shape == (1,)
            blob["b%d_cls_%s" % (k, tag)] = cl[0].numpy(); blob["b%d_box_%s" % (k, tag)] = bl[0].numpy()
        blob.update({"b%d_logits" % k: logits, "b%d_deltas" % k: deltas, "b%d_labels" % k: labels, "b%d_targets" % k: tgt})
        print("box head case", k, "R", R, "C", C, "positives", npos, "->", float(blob["b%d_cls_f64" % k]), float(blob["b%d_box_f64" % k]))
    blob["b_n"] = len(specs)
    # ---- RPN: (anchors, sampled positives, sampled negatives) ----
    rpn = ll.RegionProposalNetwork.__new__(ll.RegionProposalNetwork)
    torch.nn.Module.__init__(rpn)
    specs = [(3000, 40, 216), (12000, 128, 128), (900, 0, 256), (500, 3, 50)]
    for k, (A, npos, nneg) in enumerate(specs):
        obj = (rs.randn(A, 1) * 2.5).astype(np.float32)
        deltas = (rs.randn(A, 4) * 0.6).astype(np.float32)
        tgt = (rs.randn(A, 4) * 0.6).astype(np.float32)
        perm = rs.permutation(A)
        pos, neg = np.sort(perm[:npos]), np.sort(perm[npos:npos + nneg])
        labels = np.full(A, -1.0, np.float32)                        # -1: neither (between thresholds / not sampled anyway)
        labels[perm[:npos + 50]] = 1.0; labels[perm[npos + 50:]] = 0.0
        labels[pos] = 1.0; labels[neg] = 0.0
        pm, nm = np.zeros(A, np.uint8), np.zeros(A, np.uint8)
        pm[pos] = 1; nm[neg] = 1
        rpn.fg_bg_sampler = lambda lab, _pm=pm, _nm=nm: ([torch.from_numpy(_pm)], [torch.from_numpy(_nm)])
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            ol, bl = rpn._compute_loss([torch.from_numpy(obj).to(dt)], [torch.from_numpy(deltas).to(dt)], [torch.from_numpy(labels).to(dt)], [torch.from_numpy(tgt).to(dt)])
            blob["r%d_obj_%s" % (k, tag)] = ol[0].numpy(); blob["r%d_box_%s" % (k, tag)] = bl[0].numpy()
        blob.update({"r%d_obj" % k: obj, "r%d_deltas" % k: deltas, "r%d_targets" % k: tgt, "r%d_labels" % k: labels, "r%d_pos" % k: pos, "r%d_neg" % k: neg})
        print("rpn case", k, "anchors", A, "sampled", npos, "+", nneg, "->", float(blob["r%d_obj_f64" % k]), float(blob["r%d_box_f64" % k]))
    blob["r_n"] = len(specs)
    path = os.path.join(OUT, "frcnn_losses.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
