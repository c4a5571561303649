"""Soak of the training step: many iterations over batches of changing sizes and box counts (the shapes a real epoch produces), checks
that losses stay finite, that HBM use stops growing, and that two runs with the same seeds produce the SAME losses (every reduction of
the training step is order-independent: fixed-order split sums, single-workgroup losses, fixed-point RoIAlign-backward accumulation).

    python tools/soak_train.py [--steps 300] [--model frcnn|retinanet]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cald_amd import synth, train


def run(model, steps, seed):
    sd = synth.pseudo_trained_retinanet(21, 50, seed=0) if model == "retinanet" else synth.pseudo_trained_frcnn(21, 50, seed=0)
    net = (train.RetinaNetTrainer(sd, 21, min_size=600, max_size=1000) if model == "retinanet"
           else train.FasterRCNNTrainer(sd, 21, min_size=600, max_size=1000, generator=torch.Generator().manual_seed(seed)))
    mdl = train.TrainableDetector(net)
    opt = train.SGD([p for p in mdl.parameters() if p.requires_grad], lr=2e-5, momentum=0.9, weight_decay=1e-4, net=net)
    rs = np.random.RandomState(seed)
    pool = synth.make_pool(24, "voc", seed)
    losses, mem = [], []
    t0 = time.time()
    for it in range(steps):
        bs = int(rs.choice([1, 2, 4, 4, 4, 6]))
        idx = rs.choice(len(pool), bs, replace=False)
        ims, tgs = [], []
        for i in idx:
            im = pool[i]; H, W = im.shape[:2]
            k = int(rs.randint(1, 12))
            x0 = rs.rand(k) * W * 0.7; y0 = rs.rand(k) * H * 0.7
            boxes = np.stack([x0, y0, np.minimum(x0 + W * (0.05 + 0.4 * rs.rand(k)), W - 1), np.minimum(y0 + H * (0.05 + 0.4 * rs.rand(k)), H - 1)], axis=1).astype(np.float32)
            ims.append(torch.from_numpy(im).cuda()); tgs.append({"boxes": torch.from_numpy(boxes), "labels": torch.from_numpy(rs.randint(1, 21, k).astype(np.int64))})
        ld = mdl(ims, tgs); loss = sum(ld.values())
        opt.zero_grad(); loss.backward(); opt.step()
        if it % 10 == 0 or it == steps - 1:
            torch.cuda.synchronize()
            losses.append(float(loss.detach())); mem.append(torch.cuda.memory_reserved() / 2 ** 30)
    torch.cuda.synchronize()
    return losses, mem, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300); ap.add_argument("--model", default="frcnn")
    a = ap.parse_args()
    l1, m1, t1 = run(a.model, a.steps, 1)
    l2, m2, t2 = run(a.model, a.steps, 1)
    ok = bool(np.all(np.isfinite(l1)) and np.all(np.isfinite(l2)))
    d = float(np.max(np.abs(np.array(l1) - np.array(l2)) / np.maximum(1e-9, np.abs(l1))))
    print(json.dumps({"model": a.model, "steps": a.steps, "finite": ok, "loss_first": l1[0], "loss_last": l1[-1], "seconds": [t1, t2],
                      "reserved_GiB_first_mid_last": [m1[1] if len(m1) > 1 else m1[0], m1[len(m1) // 2], m1[-1]],
                      "max_relative_loss_difference_between_two_identical_runs": d}))
    assert ok and d == 0.0, "two identical runs differ"


if __name__ == "__main__":
    main()
