"""make_golden_f16x3.py -- tests/golden/f16x3_gpu_small.npz: what the MI355X itself computes in CALD_PRECISION_F16X3 on a tiny case.

Run on a GPU box (python tools/make_golden_f16x3.py gpurun_out/f16x3_gpu_small.npz); the file is then committed under tests/golden/ so that
the CPU-only suite can hold the oracle's f16x3 restatement (oracle/f16x3_oracle.c) against the hardware's bits without a GPU
(tests/test_oracle_golden.py::test_f16x3_oracle_reproduces_the_gpu_golden).  Inputs are regenerated from seeds on both sides
(synth.pseudo_trained_frcnn(21, 50, 0), synth.make_pool(2, "voc", 0, scale=0.4)); the file holds outputs only."""
import hashlib
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from cald_amd import detector, synth, sweep


def main():
    sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
    m = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=240, max_size=400, precision="f16x3").to("cuda")
    m.load_state_dict(sd); m.eval()
    pool = synth.make_pool(2, "voc", 0, scale=0.4)
    augs = ["flip"]
    cons, cls = sweep.sweep_device_images(m, [torch.from_numpy(im).cuda() for im in pool], [0, 1], augs, bp=1.3, base_seed=3, batch_images=2)
    out = m.forward_views([(torch.from_numpy(pool[0]).cuda(), False, None)])[0]
    stages = {}
    for name in ("conv1", "P2", "P3", "P4", "P5", "rpn0", "rpn4"):
        stages["sha1_" + name] = np.frombuffer(hashlib.sha1(np.ascontiguousarray(m.debug_tensor(name, 0)).tobytes()).digest(), np.uint8)
    np.savez_compressed(sys.argv[1], consistency=cons, cls_corr=cls, boxes=out["boxes"].cpu().numpy(), scores=out["scores"].cpu().numpy(),
                        labels=out["labels"].cpu().numpy(), **stages)
    print("wrote", sys.argv[1], cons)


if __name__ == "__main__":
    main()
