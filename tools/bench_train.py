"""Throughput of the training step on one MI355X: images / s of forward + backward + SGD at the reference's
training configuration (cald_train.py: batch size 4 per GPU, VOC images at min_size 600 / max_size 1000, 2 000 proposals,
512 sampled RoIs per image).  Synthetic VOC-sized images and boxes, pseudo-trained weights.

    python tools/bench_train.py [--batch 4] [--steps 10] [--warmup 3] [--profile out.csv]
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


def measure(batch=4, steps=10, warmup=3, depth=50, verbose=False, model="frcnn", mixed=False):
    from types import SimpleNamespace
    a = SimpleNamespace(batch=batch, steps=steps, warmup=warmup, depth=depth)
    if model == "retinanet":
        sd = synth.pseudo_trained_retinanet(21, a.depth, seed=0)
        net = train.RetinaNetTrainer(sd, 21, depth=a.depth, min_size=600, max_size=1000)
    else:
        sd = synth.pseudo_trained_frcnn(21, a.depth, seed=0)
        net = train.FasterRCNNTrainer(sd, 21, depth=a.depth, min_size=600, max_size=1000, generator=torch.Generator().manual_seed(0))
    arch = model
    model = train.TrainableDetector(net)
    opt = train.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-5, momentum=0.9, weight_decay=1e-4, net=net)
    if mixed:                       # whatever order the pool comes in: landscape and portrait images share batches (pads to 800 x 800)
        imgs = synth.make_pool(a.batch * 2, "voc", 0)
    else:                           # cald_train.py's default sampler (:326-330, --aspect-ratio-group-factor 3): one aspect-ratio group per batch
        from torch.utils.data.sampler import SequentialSampler
        from cald_amd.group_by_aspect_ratio import GroupedBatchSampler, _quantize
        sizes = synth.pool_sizes(16 * a.batch, "voc", 0)
        groups = _quantize([float(w) / float(h) for h, w in sizes], (2 ** np.linspace(-1, 1, 7)).tolist())
        picked = [b for _, b in zip(range(2), GroupedBatchSampler(SequentialSampler(sizes), groups, a.batch))]
        imgs = [synth.synth_image(i, sizes[i][0], sizes[i][1]) for b in picked for i in b]
    rs = np.random.RandomState(0)
    batches = []
    for b in range(2):
        ims, tgs = [], []
        for im in imgs[b * a.batch:(b + 1) * a.batch]:
            H, W = im.shape[:2]
            k = 3
            x0 = rs.rand(k) * W * 0.6; y0 = rs.rand(k) * H * 0.6
            boxes = np.stack([x0, y0, x0 + W * 0.3, y0 + H * 0.3], axis=1).astype(np.float32)
            ims.append(torch.from_numpy(im).cuda()); tgs.append({"boxes": torch.from_numpy(boxes), "labels": torch.from_numpy(rs.randint(1, 21, k).astype(np.int64))})
        batches.append((ims, tgs))

    def step(i, parts=None):
        ims, tgs = batches[i % 2]
        t0 = time.time()
        loss_dict = model(ims, tgs); losses = sum(loss_dict.values())
        if parts is not None:
            torch.cuda.synchronize(); t1 = time.time()
        opt.zero_grad(); losses.backward()
        if parts is not None:
            torch.cuda.synchronize(); t2 = time.time()
        opt.step()
        if parts is not None:
            torch.cuda.synchronize(); parts.append((t1 - t0, t2 - t1, time.time() - t2))
        return losses
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize(); t = time.time()
    per = []
    for i in range(a.steps):
        ts = time.time()
        last = step(i)
        per.append(round((time.time() - ts) * 1e3, 1))
    torch.cuda.synchronize(); dt = (time.time() - t) / a.steps
    if verbose:
        print("host time per step (ms, not synchronized):", per, file=sys.stderr)
    parts = []
    for i in range(4):
        step(i, parts)
    p = np.mean(parts, axis=0)
    ims, tgs = batches[0]
    net.forward(ims, tgs); torch.cuda.synchronize()
    t0 = time.time(); net.backward(); t_enq = time.time() - t0; torch.cuda.synchronize(); t_all = time.time() - t0
    if verbose:
        print("backward: host enqueue %.1f ms, until the GPU is done %.1f ms" % (t_enq * 1e3, t_all * 1e3), file=sys.stderr)
    net.flops = 0.0
    step(0)
    flops, net.flops = net.flops, None
    net.timing = []
    step(0)
    sections = {b[0]: round((b[1] - a_[1]) * 1e3, 2) for a_, b in zip(net.timing[:-1], net.timing[1:])}
    net.timing = None
    if verbose:
        print("forward sections (ms, synchronized):", sections, file=sys.stderr)
    return {"metric": "training step throughput (forward + backward + SGD)", "value": a.batch / dt, "unit": "images/s", "ms_per_step": dt * 1e3,
            "batch": a.batch, "model": arch, "depth": a.depth, "forward_ms": p[0] * 1e3, "backward_ms": p[1] * 1e3, "sgd_ms": p[2] * 1e3,
            "backward_gpu_ms": t_all * 1e3,
            "gemm": {"algorithmic_gflop_per_step": flops / 1e9, "achieved_tflops_over_whole_step": flops / dt / 1e12, "peak_tflops": 157.3,
                     "frac_of_fp32_mfma_peak": flops / dt / 1e12 / 157.3, "note": "forward + data-gradient + weight-gradient GEMM FLOPs (true channels, no padding, strided data gradients at their algorithmic cost) / wall-clock of the whole step incl. host work"}, "loss": float(last.detach()), "dtype": "f32",
            "batch_composition": "mixed orientations" if mixed else "one aspect-ratio group per batch (GroupedBatchSampler, k = 3)",
            "padded_batch_hw": [list(net.last_padded_hw)] if getattr(net, "last_padded_hw", None) else None,
            "config": "cald_train.py defaults: batch 4, VOC-sized synthetic images, min_size 600 / max_size 1000, SGD momentum 0.9"
                      + (", 2000 proposals, 512 RoIs / image" if arch == "frcnn" else ", 9 anchors / location on P3-P7")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--depth", type=int, default=50); ap.add_argument("--model", default="frcnn", choices=["frcnn", "retinanet"])
    ap.add_argument("--mixed", action="store_true", help="landscape and portrait images in one batch instead of the reference's aspect-ratio-grouped batches")
    a = ap.parse_args()
    print(json.dumps(measure(a.batch, a.steps, a.warmup, a.depth, verbose=True, model=a.model, mixed=a.mixed)))


if __name__ == "__main__":
    main()
