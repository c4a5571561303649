"""Certified RPN pruning on / off on configurations other than the headline: scores must be bit-identical; prints images/s of both.
    python tools/prune_ab.py N"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cald_amd import detector, synth, sweep


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for name, depth, shape, ncls, mn, mx, augs in (("configs[3] R50 COCO-shaped, FCD", 50, "coco", 91, 800, 1333, ["flip", "cut_out", "smaller_resize"]),
                                                   ("configs[4] R101 COCO-shaped, FCDRG (exact mode)", 101, "coco", 91, 800, 1333, ["flip", "ga", "cut_out", "smaller_resize", "rotation"]),
                                                   ("reference default --augs FCDR, R50 VOC", 50, "voc", 21, 600, 1000, ["flip", "cut_out", "smaller_resize", "rotation"]),
                                                   ("all 13 augmentation names, R50 VOC", 50, "voc", 21, 600, 1000, [a for a in sweep.SUPPORTED_AUGS])):
        sd = synth.pseudo_trained_frcnn(ncls, depth, seed=3)
        make = detector.fasterrcnn_resnet101_fpn_feature if depth == 101 else detector.fasterrcnn_resnet50_fpn_feature
        m = make(num_classes=ncls, min_size=mn, max_size=mx).to("cuda"); m.load_state_dict(sd); m.eval()
        k = n if len(augs) < 8 else max(32, n // 8)
        dev = [torch.from_numpy(im).cuda() for im in synth.make_pool(k, shape, 7)]
        pos = list(range(k))
        B = 64 if len(augs) < 8 else 16
        res = {}
        for on in (True, False):
            m.set_rpn_prune(on)
            sweep.sweep_device_images(m, dev[:B], pos[:B], augs, batch_images=B)
            torch.cuda.synchronize(); t = time.time()
            c, kk = sweep.sweep_device_images(m, dev, pos, augs, bp=1.3, base_seed=1, batch_images=B)
            torch.cuda.synchronize(); res[on] = (c, kk, k / (time.time() - t))
        same = res[True][0].tobytes() == res[False][0].tobytes() and res[True][1].tobytes() == res[False][1].tobytes()
        print("%-50s %4d images: pruned %.1f img/s, dense %.1f img/s (%+.1f %%), scores bit-identical: %s, sha1 %s"
              % (name, k, res[True][2], res[False][2], 100.0 * (res[True][2] / res[False][2] - 1.0), same, hashlib.sha1(res[True][0].tobytes()).hexdigest()[:10]), flush=True)
        assert same
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
