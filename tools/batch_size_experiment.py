import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from cald_amd import detector, synth, sweep
sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
m = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=600, max_size=1000).to("cuda"); m.load_state_dict(sd)
n = 768
pool = [torch.from_numpy(im).cuda() for im in synth.make_pool(n, "voc", 0)]
augs = ["flip", "cut_out", "smaller_resize"]
ref = None
for B in (32, 64, 128):
    sweep.sweep_device_images(m, pool[:B], list(range(B)), augs, batch_images=B)
    torch.cuda.synchronize(); t = time.time()
    c, k = sweep.sweep_device_images(m, pool, list(range(n)), augs, batch_images=B)
    torch.cuda.synchronize(); dt = time.time() - t
    if ref is None: ref = (c, k)
    print("batch_images %3d: %.1f img/s  same results: %s" % (B, n / dt, np.array_equal(c, ref[0]) and np.array_equal(k, ref[1])), flush=True)
