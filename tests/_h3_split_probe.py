"""Helper of test_f16x3_split_producers_equal_the_in_kernel_split: one small f16x3 sweep, scores to an .npz (run in a subprocess
because CALD_H3_S16 is read once per process)."""
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[2])
from cald_amd import detector, synth, sweep
arch = sys.argv[3]
if arch == "retinanet":
    sd = synth.pseudo_trained_retinanet(21, 50, seed=0)
    model = detector.retinanet_resnet50_fpn_cal(num_classes=21, min_size=300, max_size=500, precision="f16x3").to("cuda")
else:
    sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
    model = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=300, max_size=500, precision="f16x3").to("cuda")
model.load_state_dict(sd)
model.eval()
pool = synth.make_pool(20, "voc", 0, scale=0.5)
cons, cls = sweep.sweep_device_images(model, [torch.from_numpy(im).cuda() for im in pool], list(range(20)), ["flip", "cut_out", "smaller_resize"],
                                      bp=1.3, base_seed=4, batch_images=16)
np.savez(sys.argv[1], cons=cons, cls=cls)
