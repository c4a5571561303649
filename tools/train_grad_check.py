"""Per-tensor gradient error of one HIP training step against the float64 torch-CPU autograd oracle (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_train import _train_case
from cald_amd import train
from oracle import torch_train as tt

sd, images, targets = _train_case(torch)
net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, box_batch=64, generator=torch.Generator().manual_seed(7))
losses = net.forward(images, targets)
props = [p.cpu() for p in net.last["proposals"]]
grads = {k: v.clone() for k, v in net.backward().items()}
ref = tt.TorchTrainFRCNN(sd, 21, min_size=160, max_size=256)
ref.masks = net.relu_decisions()
want, rec = ref.losses(images, targets, props, None, cfg=dict(box_batch=64), samples=net.last["samples"])
sum(want.values()).backward()
tr = ref.trainable()
for k in net.names:
    w = tr[k].grad; g = grads[k].double().cpu()
    d = (g - w).abs()
    print("%-55s max|ref| %.3e  err/max %.3e  rms err/rms ref %.3e" % (k, float(w.abs().max()), float(d.max() / w.abs().max()), float(d.pow(2).mean().sqrt() / w.pow(2).mean().sqrt())))

if len(sys.argv) > 1:                 # artifact: losses and the per-tensor errors as JSON
    import json
    rows = {}
    for k in net.names:
        w = tr[k].grad; g = grads[k].double().cpu(); d = (g - w).abs()
        rows[k] = {"max_abs_ref": float(w.abs().max()), "max_err_over_max_ref": float(d.max() / w.abs().max())}
    out = {"what": "one Faster R-CNN R50-FPN training step (2 images, min_size 160 / max_size 256, 64 RoIs / image): HIP fp32 vs the float64 "
                   "torch-CPU autograd restatement (oracle/torch_train.py) given the HIP run's proposals, sampler draws and ReLU decisions",
           "losses_hip": {k: float(v) for k, v in losses.items()}, "losses_ref": {k: float(v) for k, v in want.items()},
           "tensors": len(rows), "worst_tensor_error": max(r["max_err_over_max_ref"] for r in rows.values()),
           "tolerance_in_tests": 1e-4, "per_tensor": rows}
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print("worst", out["worst_tensor_error"])
