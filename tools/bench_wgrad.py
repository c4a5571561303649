"""Times the training GEMM kernels on single layer shapes (torch.cuda.Event on the context stream): forward conv, data gradient, weight
gradient.  TFLOP/s = algorithmic 2 * pixels * Cout * Cin * taps / time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cald_amd import train_ops as ops

SHAPES = [   # name, N, H, W, Cin, Cout, K, stride
    ("FPN/RPN 3x3 @P2 (batch 4, 800x800)", 4, 200, 200, 256, 256, 3, 1),
    ("FPN/RPN 3x3 @P3", 4, 100, 100, 256, 256, 3, 1),
    ("FPN/RPN 3x3 @P4", 4, 50, 50, 256, 256, 3, 1),
    ("layer2 3x3 128", 4, 100, 100, 128, 128, 3, 1),
    ("layer3 3x3 256", 4, 50, 50, 256, 256, 3, 1),
    ("layer4 3x3 512", 4, 25, 25, 512, 512, 3, 1),
    ("layer3 1x1 256->1024", 4, 50, 50, 256, 1024, 1, 1),
    ("layer3 1x1 1024->256", 4, 50, 50, 1024, 256, 1, 1),
    ("layer4 1x1 2048->512", 4, 25, 25, 2048, 512, 1, 1),
    ("lateral 1x1 256->256 @C2", 4, 200, 200, 256, 256, 1, 1),
]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    print("%-38s %10s %10s %10s   (TFLOP/s; fp32 MFMA peak 157.3)" % ("layer", "forward", "dgrad", "wgrad"))
    for name, N, H, W, Cin, Cout, K, s in SHAPES:
        x = torch.randn(N, H, W, Cin, device="cuda", generator=g)
        w = torch.randn(Cout, Cin, K, K, device="cuda", generator=g) / (Cin * K * K) ** 0.5
        pk = ops.PackedConv(w); pkd = ops.PackedConv(w, CinK=Cout, mode=1)
        pad = K // 2
        y = ops.conv(x, pk, stride=s, pad=pad)
        gy = torch.randn_like(y)
        dw = torch.empty_like(w)
        flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * Cin * K * K
        tf = timed(lambda: ops.conv(x, pk, stride=s, pad=pad, out=y))
        td = timed(lambda: ops.conv_dgrad(gy, pkd, H, W, s, pad))
        tw = timed(lambda: ops.conv_wgrad(x, gy, Cin, Cout, K, K, s, pad, dw))
        print("%-38s %10.1f %10.1f %10.1f   ms %.3f / %.3f / %.3f" % (name, flops / tf / 1e9, flops / td / 1e9, flops / tw / 1e9, tf, td, tw))


if __name__ == "__main__":
    main()
