"""Host-side timeline of the training loop (no synchronization added): when the host returns from forward / backward / optimizer step,
against the GPU's own clock for the step -- shows whether the host runs ahead of the GPU across the step boundary."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cald_amd import train, synth

model_name = sys.argv[1] if len(sys.argv) > 1 else "frcnn"
if model_name == "retinanet":
    net = train.RetinaNetTrainer(synth.pseudo_trained_retinanet(21, 50, seed=0), 21, depth=50, min_size=600, max_size=1000)
else:
    net = train.FasterRCNNTrainer(synth.pseudo_trained_frcnn(21, 50, seed=0), 21, depth=50, min_size=600, max_size=1000,
                                  generator=torch.Generator().manual_seed(0))
model = train.TrainableDetector(net)
opt = train.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-5, momentum=0.9, weight_decay=1e-4, net=net)
imgs = synth.make_pool(4, "voc", 0)
rs = np.random.RandomState(0)
ims, tgs = [], []
for im in imgs:
    H, W = im.shape[:2]
    x0 = rs.rand(3) * W * 0.6; y0 = rs.rand(3) * H * 0.6
    boxes = np.stack([x0, y0, x0 + W * 0.3, y0 + H * 0.3], axis=1).astype(np.float32)
    ims.append(torch.from_numpy(im).cuda()); tgs.append({"boxes": torch.from_numpy(boxes), "labels": torch.from_numpy(rs.randint(1, 21, 3).astype(np.int64))})
rows = []
for i in range(12):
    if i == 4:
        if "norepack" in sys.argv:                      # A/B: how much of the step the weight packing costs (weights go stale: timing only)
            net._repack = lambda: None
        torch.cuda.synchronize(); t_start = time.time()
    t0 = time.time()
    marks = []
    net._mark = lambda name, _m=marks: _m.append((name, time.time()))      # host clock only: the stock _mark synchronizes
    losses = sum(model(ims, tgs).values()); t1 = time.time()
    opt.zero_grad(); losses.backward(); t2 = time.time()
    opt.step(); t3 = time.time()
    if i >= 4:
        rows.append((t0 - t_start, t1 - t0, t2 - t1, t3 - t2, marks))
torch.cuda.synchronize(); t_end = time.time()
print("steps 4..11: %.2f ms / step" % ((t_end - t_start) / 8 * 1e3))
for r in rows:
    print("step starts at %7.2f ms: forward returns after %6.2f, backward enqueue %5.2f, optimizer %5.2f" % tuple(x * 1e3 for x in r[:4]))
if rows[-1][4]:
    m = rows[-1][4]
    print("forward marks of the last step (host clock, ms since its start):", [(n, round((t - m[0][1]) * 1e3, 2)) for n, t in m])
