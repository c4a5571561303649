"""8-GPU readiness on one GPU: the driver's N = 8 strong-scaling run at --steps 20 hands every rank 160 images.  Times the
sweep of a 160-image shard (strided shard 0 of 8 of a 1 280-image pool, balanced batches of 80) against the 1 280-image sweep (balanced
batches of 92) on the same GPU: ratio of the two rates.  Usage: python tools/shard160.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cald_amd import detector, synth, sweep


def balanced(n, cap=96):
    return max(1, -(-n // max(1, -(-n // cap))))


def main():
    sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
    model = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=600, max_size=1000).to("cuda")
    model.load_state_dict(sd); model.eval()
    pool = [torch.from_numpy(im).cuda() for im in synth.make_pool(1280, "voc", 0)]
    augs = ["flip", "cut_out", "smaller_resize"]
    res = {}
    sweep.sweep_device_images(model, pool[:192], list(range(192)), augs, batch_images=96)     # warm-up
    for name, idx in (("pool_1280", list(range(1280))), ("shard_160_of_8", list(range(0, 1280, 8))), ("pool_1280_again", list(range(1280))),
                      ("shard_160_again", list(range(0, 1280, 8)))):
        B = balanced(len(idx))
        sweep.sweep_device_images(model, [pool[i] for i in idx[:B]], idx[:B], augs, batch_images=B)   # same scratch geometry as the timed call
        torch.cuda.synchronize(); t = time.time()
        c, _ = sweep.sweep_device_images(model, [pool[i] for i in idx], idx, augs, batch_images=B)
        torch.cuda.synchronize(); t = time.time() - t
        res[name] = {"images": len(idx), "batch_images": B, "seconds": t, "images_per_s": len(idx) / t, "mean_consistency": float(np.mean(c))}
        print(name, res[name], flush=True)
    res["ratio_160_vs_1280"] = min(res["shard_160_of_8"]["images_per_s"], res["shard_160_again"]["images_per_s"]) / max(res["pool_1280"]["images_per_s"], res["pool_1280_again"]["images_per_s"])
    res["ratio_best_vs_best"] = max(res["shard_160_of_8"]["images_per_s"], res["shard_160_again"]["images_per_s"]) / max(res["pool_1280"]["images_per_s"], res["pool_1280_again"]["images_per_s"])
    print(json.dumps(res))
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
