import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cald_amd import train_ops as ops
g = torch.Generator(device="cuda").manual_seed(0)
N, H, W, Cin, Cout, K = 4, 200, 200, 256, 256, 3
x = torch.randn(N, H, W, Cin, device="cuda", generator=g); w = torch.randn(Cout, Cin, K, K, device="cuda", generator=g) / 48
gy = torch.randn(N, H, W, Cout, device="cuda", generator=g); dw = torch.empty_like(w)
for _ in range(6):
    ops.conv_wgrad(x, gy, Cin, Cout, K, K, 1, 1, dw)
torch.cuda.synchronize()
