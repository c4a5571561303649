"""Data for the cascade's thresholds: the configs[1] pool (first N images) scored in exact fp32 and in f16x3 WITH the decision-margin
audit; writes consistency / cls_corr of both modes and the per-image margin records to an .npz and prints, per margin kind, how the
smallest margins of the images that DID change (|d consistency| > 1e-5 or a changed cls_corr) compare with those of the rest.
    python tools/cascade_margins.py N out.npz [model: frcnn|frcnn101coco]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cald_amd import _ffi, detector, synth, sweep


def main():
    n = int(sys.argv[1]); out = sys.argv[2]
    which = sys.argv[3] if len(sys.argv) > 3 else "frcnn"
    if which == "frcnn":
        sd = synth.pseudo_trained_frcnn(21, 50, seed=0); make = detector.fasterrcnn_resnet50_fpn_feature
        kw = dict(num_classes=21, min_size=600, max_size=1000); shape = "voc"; augs = ["flip", "cut_out", "smaller_resize"]
    else:
        sd = synth.pseudo_trained_frcnn(91, 101, seed=1); make = detector.fasterrcnn_resnet101_fpn_feature
        kw = dict(num_classes=91, min_size=800, max_size=1333); shape = "coco"; augs = ["flip", "ga", "cut_out", "smaller_resize", "rotation"]
    pool = [torch.from_numpy(im).cuda() for im in synth.make_pool(n, shape, 0)]
    pos = list(range(n))
    res = {}
    for prec in ("fp32", "f16x3"):
        m = make(precision=prec, **kw).to("cuda"); m.load_state_dict(sd); m.eval()
        sweep.sweep_device_images(m, pool[:96], pos[:96], augs, batch_images=96)
        torch.cuda.synchronize(); t = time.time()
        plain = sweep.sweep_device_images(m, pool, pos, augs, bp=1.3, base_seed=0, batch_images=96)
        torch.cuda.synchronize(); t1 = time.time() - t; t = time.time()
        c, k, mg = sweep.sweep_device_images(m, pool, pos, augs, bp=1.3, base_seed=0, batch_images=96, margins=True)
        torch.cuda.synchronize(); t2 = time.time() - t
        assert np.array_equal(plain[0], c) and np.array_equal(plain[1], k), "the audit changed the scores"
        print("%s: %.1f images/s plain, %.1f with the audit" % (prec, n / t1, n / t2))
        res[prec] = (c, k, mg)
        del m
        torch.cuda.empty_cache()
    (ce, ke, me), (ch, kh, mh) = res["fp32"], res["f16x3"]
    d = np.abs(ce - ch); dk = np.abs(ke - kh).max(axis=1)
    np.savez(out, cons_exact=ce, cls_exact=ke, margins_exact=me, cons_fast=ch, cls_fast=kh, margins_fast=mh)
    changed = (d > 1e-5) | (dk > 1e-5)
    print("images: %d; |d consistency| > 1e-5: %d, > 1e-4: %d; cls_corr changed > 1e-5: %d; either: %d" %
          (n, int((d > 1e-5).sum()), int((d > 1e-4).sum()), int((dk > 1e-5).sum()), int(changed.sum())))
    print("unchanged images: max |d consistency| %.3g, max |d cls_corr| %.3g" % (d[~changed].max(), dk[~changed].max()))
    print("%-14s %12s %12s %12s | quantiles of the fast mode's margin over ALL images: 1%% 5%% 25%%" % ("kind", "min(changed)", "med(changed)", "min(rest)"))
    for q, name in enumerate(_ffi.MARGIN_NAMES[:15]):
        a, b = mh[changed, q], mh[~changed, q]
        fin = mh[:, q][np.isfinite(mh[:, q])]
        qs = np.quantile(fin, [0.01, 0.05, 0.25]) if len(fin) else [np.inf] * 3
        print("%-14s %12.3g %12.3g %12.3g | %10.3g %10.3g %10.3g" % (name, a.min() if len(a) else np.inf, np.median(a) if len(a) else np.inf,
                                                                    b.min() if len(b) else np.inf, qs[0], qs[1], qs[2]))


if __name__ == "__main__":
    main()
