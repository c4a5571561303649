#!/usr/bin/env python
"""GPU-side timeline of the training step WITHOUT a profiler and without added synchronization: HIP events recorded on the stream each
forward section is enqueued on (the marks of FasterRCNNTrainer.forward), around backward() and the optimizer step; after the run the
median time of every mark relative to the previous step's "sgd done" event.  Shows where the main stream waits at the step boundary.
    python tools/train_event_timeline.py [steps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cald_amd import train, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
net = train.FasterRCNNTrainer(synth.pseudo_trained_frcnn(21, 50, seed=0), 21, depth=50, min_size=600, max_size=1000,
                              generator=torch.Generator().manual_seed(0))
model = train.TrainableDetector(net)
opt = train.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-5, momentum=0.9, weight_decay=1e-4, net=net)
from torch.utils.data.sampler import SequentialSampler
from cald_amd.group_by_aspect_ratio import GroupedBatchSampler, _quantize
sizes = synth.pool_sizes(64, "voc", 0)
groups = _quantize([float(w) / float(h) for h, w in sizes], (2 ** np.linspace(-1, 1, 7)).tolist())
picked = [b for _, b in zip(range(2), GroupedBatchSampler(SequentialSampler(sizes), groups, 4))]
rs = np.random.RandomState(0)
batches = []
for b in picked:
    ims, tgs = [], []
    for i in b:
        im = synth.synth_image(i, sizes[i][0], sizes[i][1]); H, W = im.shape[:2]
        x0 = rs.rand(3) * W * 0.6; y0 = rs.rand(3) * H * 0.6
        boxes = np.stack([x0, y0, x0 + W * 0.3, y0 + H * 0.3], axis=1).astype(np.float32)
        ims.append(torch.from_numpy(im).cuda()); tgs.append({"boxes": torch.from_numpy(boxes), "labels": torch.from_numpy(rs.randint(1, 21, 3).astype(np.int64))})
    batches.append((ims, tgs))
marks = []
def mark(name):
    ev = torch.cuda.Event(enable_timing=True); ev.record(torch.cuda.current_stream()); marks.append((name, ev))
net._mark = mark
import time
host = []
for i in range(steps):
    ims, tgs = batches[i % 2]
    h0 = time.time()
    loss = sum(model(ims, tgs).values())
    mark("fwd enqueued"); h1 = time.time()
    opt.zero_grad(); loss.backward()
    mark("bwd done"); h2 = time.time()
    opt.step()
    mark("sgd done"); host.append((h1 - h0, h2 - h1, time.time() - h2))
torch.cuda.synchronize()
# split into steps at "sgd done"
per, cur = [], []
for name, ev in marks:
    cur.append((name, ev))
    if name == "sgd done":
        per.append(cur); cur = []
rows = {}
for k in range(steps // 2, steps):
    base = per[k - 1][-1][1]
    for name, ev in per[k]:
        rows.setdefault(name, []).append(base.elapsed_time(ev))
print("median ms after the previous step's 'sgd done' (GPU clock, stream of the section):")
for name, v in rows.items():
    print("  %-14s %7.2f" % (name, float(np.median(v))))
print("host ms per step: forward %.2f  backward %.2f  sgd %.2f" % tuple(np.median(np.array(host[steps // 2:]), axis=0) * 1e3))
