"""Does keeping a layer's tensors inside the 256 MiB Infinity Cache help the HBM-bound 1x1 layers?  Same layer, V views per launch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cald_amd import _ffi, detector
L, ctx = _ffi.lib(), detector.get_ctx(0)
CASES = [("l1 conv3 64->256 +res @L2", 152, 200, 64, 256, 1, 1, 0, 1), ("l1 conv1 256->64 @L2", 152, 200, 256, 64, 1, 1, 0, 0),
         ("l1 conv2 3x3 64->64 @L2", 152, 200, 64, 64, 3, 1, 1, 0), ("l2 conv3 128->512 +res @L3", 76, 100, 128, 512, 1, 1, 0, 1),
         ("stem 7x7", 608, 800, 4, 64, 7, 2, 3, 0)]
for (name, H, W, Cin, Cout, K, s, p, res) in CASES:
    row = []
    for V in (1, 2, 4, 8, 16, 64):
        ms, tf = C.c_double(), C.c_double()
        _ffi.check(L.cald_op_conv_bench(ctx, V, H, W, Cin, Cout, K, s, p, res, 1, 20, 1, C.byref(ms), C.byref(tf)))
        row.append("V=%d: %.1f TF (%.3f ms)" % (V, tf.value, ms.value))
    print("%-30s %s" % (name, "  ".join(row)), flush=True)
