"""Scores the configs[1] synthetic pool with the INDEPENDENT fp32 path -- the reference-shaped torch-CPU port
(oracle/torch_port.py: oneDNN conv / linear summation order, torch's own exp / softmax / interpolate, python + scipy
scoring loop) -- and stores (consistency, cls_corr) per pool position.  No GPU involved: it runs for hours on host
cores (resumable), and tools/parity_full_pool.py then compares the MI355X path against the stored vectors in seconds.

    python tools/torch_cpu_reference.py OUT.npz [n_images=5217] [threads=all]

TEST INFRASTRUCTURE (imports oracle/); never imported by the product."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

AUGS = os.environ.get("CALD_PARITY_AUGS", "flip,cut_out,smaller_resize").split(",")     # configs[0]: CALD_PARITY_AUGS=flip


def main():
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5217
    if len(sys.argv) > 3:
        torch.set_num_threads(int(sys.argv[3]))
    from oracle import torch_port
    from cald_amd import synth
    sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
    model = torch_port.TorchFRCNN(sd, 21, 50, 600, 1000)
    sizes = synth.pool_sizes(n, "voc", 0)
    cons = np.zeros(n, np.float64); cls = np.zeros((n, 20), np.float64); done = np.zeros(n, bool)
    if os.path.exists(out):
        z = np.load(out)
        m = min(n, len(z["done"]))
        cons[:m], cls[:m], done[:m] = z["consistency"][:m], z["cls_corr"][:m], z["done"][:m]
    t0 = time.time(); k = 0
    for p in range(n):
        if done[p]:
            continue
        img = synth.synth_image(p, *sizes[p])
        c, cc = torch_port.get_uncertainty(model, [img], AUGS, 21, bp=1.3, base_seed=0, positions=[p])
        cons[p], cls[p], done[p] = c[0], cc[0], True
        k += 1
        if k % 25 == 0 or p == n - 1:
            tmp = out + ".tmp.npz"
            np.savez_compressed(tmp, consistency=cons, cls_corr=cls, done=done, augs=np.array(AUGS), base_seed=0,
                                torch_version=torch.__version__, threads=torch.get_num_threads())
            os.replace(tmp, out)
            print("%d / %d done, %.2f s/image" % (int(done.sum()), n, (time.time() - t0) / k), flush=True)


if __name__ == "__main__":
    main()
