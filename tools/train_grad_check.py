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
