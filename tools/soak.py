"""Soak runs of the non-headline configurations (RetinaNet VOC at scale, ResNet-101 COCO-shaped with six augmentations, all 13
augmentation names in f16x3 mode): finiteness and throughput.  Usage: python tools/soak.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cald_amd import detector, synth, sweep
def run(name, model, pool, augs):
    t = time.time()
    cons, cls = sweep.sweep_device_images(model, pool, list(range(len(pool))), augs, bp=1.3, base_seed=1)
    torch.cuda.synchronize(); dt = time.time() - t
    print(name, len(pool), "images %.1f img/s" % (len(pool) / dt), "finite", bool(np.isfinite(cons).all() and np.isfinite(cls).all()),
          "cons range", float(cons.min()), float(cons.max()), "zero-score", int((cons == 0).sum()))
pool = [torch.from_numpy(im).cuda() for im in synth.make_pool(2048, "voc", 7)]
m = detector.retinanet_resnet50_fpn_cal(num_classes=21, min_size=600, max_size=1000).to("cuda"); m.load_state_dict(synth.pseudo_trained_retinanet(21, 50, seed=0)); m.eval()
run("retinanet voc FCD", m, pool, ["flip", "cut_out", "smaller_resize"]); del m
pool = [torch.from_numpy(im).cuda() for im in synth.make_pool(384, "coco", 3)]
m = detector.fasterrcnn_resnet101_fpn_feature(num_classes=91, min_size=800, max_size=1333).to("cuda"); m.load_state_dict(synth.pseudo_trained_frcnn(91, 101, seed=0)); m.eval()
run("frcnn101 coco FCDRGS", m, pool, ["flip", "cut_out", "smaller_resize", "rotation", "ga", "sp"]); del m
m = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=600, max_size=1000, precision="f16x3").to("cuda"); m.load_state_dict(synth.pseudo_trained_frcnn(21, 50, seed=0)); m.eval()
pool = [torch.from_numpy(im).cuda() for im in synth.make_pool(256, "voc", 11)]
run("frcnn50 voc ALL-13-augs f16x3", m, pool, ["flip", "ga", "multi_ga", "color_adjust", "color_swap", "sp", "multi_sp", "cut_out", "multi_cut_out", "multi_resize", "larger_resize", "smaller_resize", "rotation"])
