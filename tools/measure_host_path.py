"""PCIe-inclusive rate of the Python boundary: get_uncertainty() fed host-side images (numpy uint8), i.e. including
the per-image H2D upload that bench.py's `value` excludes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cald_amd import detector, synth, sweep
sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
model = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=600, max_size=1000).to("cuda")
model.load_state_dict(sd)
pool = synth.make_pool(192, "voc", 0)
loader = [((torch.from_numpy(im),), (None,)) for im in pool]
augs = ["flip", "cut_out", "smaller_resize"]
sweep.get_uncertainty(model, loader[:64], augs, 21)
torch.cuda.synchronize(); t = time.time()
sweep.get_uncertainty(model, loader, augs, 21)
torch.cuda.synchronize(); dt = time.time() - t
print("get_uncertainty from host images: %.1f images/s (192 images, %.2f s)" % (192 / dt, dt))
