"""Where does precision="f16x3" lose accuracy against the exact fp32 mode?  One forward of a few views in both modes, every stage
tensor (cald_debug_tensor) compared: max |x|, max |d|, max |d| / max |x|, rms(d) / rms(x).  Then short sweeps of the same pool
with different augmentation sets (which view kind carries the error?).
    python tools/f16x3_stage_error.py DEPTH SHAPE CLASSES SEED [n_sweep_images]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cald_amd import detector, synth, sweep


def main():
    depth, shape, ncls, seed = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    nsw = int(sys.argv[5]) if len(sys.argv) > 5 else 32
    mn, mx = (600, 1000) if shape == "voc" else (800, 1333)
    sd = synth.pseudo_trained_frcnn(ncls, depth, seed=seed)
    make = detector.fasterrcnn_resnet101_fpn_feature if depth == 101 else detector.fasterrcnn_resnet50_fpn_feature
    pool = synth.make_pool(max(nsw, 2), shape, 0)
    dev = [torch.from_numpy(im).cuda() for im in pool]
    models = {}
    for prec in ("fp32", "f16x3"):
        m = make(num_classes=ncls, min_size=mn, max_size=mx, precision=prec).to("cuda")
        m.load_state_dict(sd); m.eval()
        models[prec] = m
    names = ["input", "conv1", "pool1", "C2", "C3", "C4", "C5", "P2", "P3", "P4", "P5", "P6", "rpn0", "rpn1", "rpn2", "rpn3", "rpn4"]
    views = [(dev[0], False, None), (dev[1], True, None)]
    T = {}
    for prec, m in models.items():
        m.forward_views(views)
        T[prec] = {(n, v): m.debug_tensor(n, v) for n in names for v in range(2)}
        T[prec].update({("proposals", v): m.debug_tensor("proposals", v) for v in range(2)})
    print("stage        view    max|x|      max|d|    max|d|/max|x|   rms(d)/rms(x)")
    for n in names:
        for v in range(2):
            a, b = T["fp32"][(n, v)].astype(np.float64), T["f16x3"][(n, v)].astype(np.float64)
            d = np.abs(a - b)
            print("%-12s %d   %10.4g  %10.3g  %12.3g  %12.3g" % (n, v, np.abs(a).max(), d.max(), d.max() / max(np.abs(a).max(), 1e-30),
                                                               np.sqrt((d ** 2).mean()) / max(np.sqrt((a ** 2).mean()), 1e-30)))
    for v in range(2):
        a, b = T["fp32"][("proposals", v)].reshape(-1, 4), T["f16x3"][("proposals", v)].reshape(-1, 4)
        same = (np.abs(a - b).max(axis=1) < 1e-2)
        print("proposals view %d: rows equal within 1e-2 px: %d / %d; first differing row %s; max |d| on equal rows %.3g"
              % (v, int(same.sum()), len(a), (int(np.argmin(same)) if not same.all() else None), float(np.abs(a - b)[same].max())))
    pos = list(range(nsw))
    for augs in (["flip"], ["ga"], ["cut_out"], ["smaller_resize"], ["rotation"], ["flip", "cut_out", "smaller_resize"]):
        ce, _ = sweep.sweep_device_images(models["fp32"], dev[:nsw], pos, augs, bp=1.3, base_seed=4, batch_images=32)
        ch, _ = sweep.sweep_device_images(models["f16x3"], dev[:nsw], pos, augs, bp=1.3, base_seed=4, batch_images=32)
        d = np.abs(ce - ch)
        print("augs %-40s median |d| %.3g  max %.3g  beyond 1e-4: %d / %d  beyond 1e-5: %d" % (augs, np.median(d), d.max(), int((d > 1e-4).sum()), nsw, int((d > 1e-5).sum())))


if __name__ == "__main__":
    main()
