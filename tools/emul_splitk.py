"""What a 2- / 4-way split of the reduction would buy the small body layers at batch 4: the same FLOPs as (N, Cin) run as (2N, Cin / 2)
and (4N, Cin / 4) -- twice / four times the workgroups, half / a quarter of the k-loop (the partial-sum pass is not included)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cald_amd import train_ops as ops
from bench_wgrad import timed

SHAPES = [("layer3 3x3 256", 4, 50, 50, 256, 256, 3), ("layer4 3x3 512", 4, 25, 25, 512, 512, 3),
          ("layer3 1x1 1024->256", 4, 50, 50, 1024, 256, 1), ("layer4 1x1 2048->512", 4, 25, 25, 2048, 512, 1),
          ("layer3 1x1 256->1024", 4, 50, 50, 256, 1024, 1)]
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, H, W, Cin, Cout, K in SHAPES:
    row = []
    for s in (1, 2, 4):
        x = torch.randn(N * s, H, W, Cin // s, device="cuda", generator=g)
        w = torch.randn(Cout, Cin // s, K, K, device="cuda", generator=g) * 0.05
        pk = ops.PackedConv(w)
        ms = timed(lambda: ops.conv(x, pk, stride=1, pad=K // 2, relu=True))
        row.append(ms * 1e3)
    print("%-24s  1 split %6.1f us   2 splits %6.1f us   4 splits %6.1f us" % (name, *row))
