"""Sustained-load check: the same layer timed over short and long back-to-back runs (does the rate sag under sustained load?)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cald_amd import _ffi, detector
L, ctx = _ffi.lib(), detector.get_ctx(0)
for name, args in (("fc6", (64, 1, 1000, 12544, 1024, 1, 1, 0, 0, 1)), ("3x3 256 P2", (64, 152, 200, 256, 256, 3, 1, 1, 0, 1))):
    for iters in (2, 10, 40, 10, 2):
        ms, tf = C.c_double(), C.c_double()
        _ffi.check(L.cald_op_conv_bench(ctx, *args, iters, 1, C.byref(ms), C.byref(tf)))
        print("%-12s iters=%3d  %.3f ms  %.1f TF" % (name, iters, ms.value, tf.value), flush=True)
