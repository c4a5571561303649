"""Isolated timing of the fc6 weight gradient (linear layer on the RoIAlign output, 49 taps): kernel and reduce, by torch events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cald_amd import train_ops as ops

R, K, Cout, taps = 2048, 12544, 1024, 49
x = torch.randn(R, K, device="cuda")
g = torch.randn(R, Cout, device="cuda")
dw = torch.empty(Cout, K // taps, 7, 7, device="cuda")
db = torch.empty(Cout, device="cuda")
for _ in range(3):
    ops.linear_wgrad(x, g, Cout, dw, db, taps=taps)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.linear_wgrad(x, g, Cout, dw, db, taps=taps)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("fc6 wgrad %.3f ms  (%.1f TFLOP/s incl. reduce)" % (ms, 2.0 * R * K * Cout / ms / 1e9))
