"""Times individual conv / linear layer shapes of the configs[1] forward on the MI355X (cald_op_conv_bench): the tuning loop
for the GEMM kernels.  Usage: python tools/bench_conv.py [tag]   (env vars select kernel variants, see conv_p4.hip)"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cald_amd import _ffi, detector

# name, V, H, W, Cin, Cout, K, stride, pad, residual, relu, group      (64 views of 608 x 800 -> level sizes)
LAYERS = [
    ("fpn/rpn 3x3 256->256 @P2 (x1)", 64, 152, 200, 256, 256, 3, 1, 1, 0, 1, 1),
    ("l1 conv3 1x1 64->256 +res @L2", 64, 152, 200, 64, 256, 1, 1, 0, 1, 1, 1),
    ("l1 conv1 1x1 256->64 @L2", 64, 152, 200, 256, 64, 1, 1, 0, 0, 1, 1),
    ("l1 conv2 3x3 64->64 @L2", 64, 152, 200, 64, 64, 3, 1, 1, 0, 1, 1),
    ("l2 conv3 1x1 128->512 +res @L3", 64, 76, 100, 128, 512, 1, 1, 0, 1, 1, 1),
    ("l2 conv2 3x3 128->128 @L3", 64, 76, 100, 128, 128, 3, 1, 1, 0, 1, 1),
    ("l3 conv2 3x3 256->256 @L4", 64, 38, 50, 256, 256, 3, 1, 1, 0, 1, 1),
    ("l3 conv3 1x1 256->1024 +res @L4", 64, 38, 50, 256, 1024, 1, 1, 0, 1, 1, 1),
    ("l3 conv1 1x1 1024->256 @L4", 64, 38, 50, 1024, 256, 1, 1, 0, 0, 1, 1),
    ("l4 conv2 3x3 512->512 @L5", 64, 19, 25, 512, 512, 3, 1, 1, 0, 1, 1),
    ("l4 conv3 1x1 512->2048 +res @L5", 64, 19, 25, 512, 2048, 1, 1, 0, 1, 1, 1),
    ("fc6 12544->1024 (1000 rows/view)", 64, 1, 1000, 12544, 1024, 1, 1, 0, 0, 1, 1),
    ("fc7 1024->1024", 64, 1, 1000, 1024, 1024, 1, 1, 0, 0, 1, 1),
    ("stem 7x7 s2 4->64", 64, 608, 800, 4, 64, 7, 2, 3, 0, 1, 1),
]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "base"
    only = sys.argv[2] if len(sys.argv) > 2 else None
    L, ctx = _ffi.lib(), detector.get_ctx(0)
    res = {}
    for (name, V, H, W, Cin, Cout, K, s, p, resid, relu, grp) in LAYERS:
        if only and only not in name:
            continue
        ms, tf = C.c_double(), C.c_double()
        iters = 3 if (H * W * Cin * Cout * K * K > 2e10) else 8
        _ffi.check(L.cald_op_conv_bench(ctx, V, H, W, Cin, Cout, K, s, p, resid, relu, iters, grp, C.byref(ms), C.byref(tf)))
        res[name] = {"ms": ms.value, "tflops": tf.value}
        print("%-40s %8.3f ms  %6.1f TF" % (name, ms.value, tf.value), flush=True)
    print(json.dumps({"tag": tag, "env": {k: v for k, v in os.environ.items() if k.startswith("CALD_")}, "layers": res}))


if __name__ == "__main__":
    main()
