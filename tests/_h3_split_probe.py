"""Helper of the tests that compare two code paths selected by an environment variable the library reads once per process
(CALD_H3_S16, CALD_ROI_ROWS): one small sweep, scores to an .npz.  argv: out.npz repo_root arch [precision] [min_size max_size]"""
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[2])
from cald_amd import detector, synth, sweep
arch = sys.argv[3]
prec = sys.argv[4] if len(sys.argv) > 4 else "f16x3"
mn, mx = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (300, 500)
if arch == "retinanet":
    sd = synth.pseudo_trained_retinanet(21, 50, seed=0)
    model = detector.retinanet_resnet50_fpn_cal(num_classes=21, min_size=mn, max_size=mx, precision=prec).to("cuda")
else:
    sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
    model = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=mn, max_size=mx, precision=prec).to("cuda")
model.load_state_dict(sd)
model.eval()
pool = synth.make_pool(20, "voc", 0, scale=mn / 600.0)
cons, cls = sweep.sweep_device_images(model, [torch.from_numpy(im).cuda() for im in pool], list(range(20)), ["flip", "cut_out", "smaller_resize"],
                                      bp=1.3, base_seed=4, batch_images=16)
np.savez(sys.argv[1], cons=cons, cls=cls)
