"""How far is the arithmetic-contract path (C oracle == HIP, bit for bit) from an independent torch-CPU fp32 path
(different summation order inside conv/linear)?  Reports per-image |delta consistency| and top-k overlap."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc, torch_port
from cald_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
mn, mx = (300, 500) if scale < 1 else (600, 1000)
sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
P = orc.prepare_frcnn(sd, 21, 50)
tm = torch_port.TorchFRCNN(sd, 21, 50, mn, mx)
pool = synth.make_pool(n, "voc", 0, scale=scale)
augs = ["flip", "cut_out", "smaller_resize"]
t = time.time(); a, _ = orc.get_uncertainty(P, pool, augs, 21, 1.3, mn, mx, 0); ta = time.time() - t
t = time.time(); b, _ = torch_port.get_uncertainty(tm, pool, augs, 21, 1.3, 0); tb = time.time() - t
a, b = np.array(a), np.array(b)
d = np.abs(a - b)
k = max(1, n // 4)
print("images %d  oracle %.1fs  torch %.1fs" % (n, ta, tb))
print("|delta consistency|: max %.3g  median %.3g  within 1e-4: %d/%d" % (d.max(), np.median(d), int((d <= 1e-4).sum()), n))
print("top-%d selection identical: %s  overlap %d/%d" % (k, np.array_equal(np.argsort(a)[:k], np.argsort(b)[:k]),
                                                        len(set(np.argsort(a)[:k]) & set(np.argsort(b)[:k])), k))
