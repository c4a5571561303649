"""How often does an INDEPENDENT fp32 implementation (the torch-CPU port: oneDNN summation order, torch's exp / softmax)
disagree with the exact mode (== the oracle, bit for bit) on the same images?  Context for DESIGN section 4b: any path that is
not bit-identical flips some borderline NMS / threshold decisions.  GPU exact mode vs torch-CPU on N full-size images
(CPU side: `workers` processes x 16 threads).  Usage: python tools/exact_vs_torch_cpu.py [n] [workers] [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

AUGS = ["flip", "cut_out", "smaller_resize"]


def worker(args):
    lo, hi = args
    import torch
    torch.set_num_threads(16)
    from oracle import torch_port
    from cald_amd import synth
    sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
    model = torch_port.TorchFRCNN(sd, 21, 50, 600, 1000)
    sizes = synth.pool_sizes(hi, "voc", 0)
    imgs = [synth.synth_image(p, *sizes[p]) for p in range(lo, hi)]
    cons, _ = torch_port.get_uncertainty(model, imgs, AUGS, 21, bp=1.3, base_seed=0, positions=list(range(lo, hi)))
    return lo, cons


if __name__ == "__main__":
    import multiprocessing as mp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    step = (n + workers - 1) // workers
    t = time.time()
    with mp.get_context("spawn").Pool(workers) as pool:
        parts = pool.map(worker, [(lo, min(lo + step, n)) for lo in range(0, n, step)])
    t_cpu = time.time() - t
    cpu = np.concatenate([np.asarray(c, np.float64) for _, c in sorted(parts)])
    import torch
    from cald_amd import detector, synth, sweep
    sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
    m = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=600, max_size=1000).to("cuda")
    m.load_state_dict(sd); m.eval()
    pool_imgs = [torch.from_numpy(im).cuda() for im in synth.make_pool(n, "voc", 0)]
    gpu, _ = sweep.sweep_device_images(m, pool_imgs, list(range(n)), AUGS, bp=1.3, base_seed=0)
    d = np.abs(gpu - cpu)
    k = max(1, n // 10)
    out = {"images": n, "cpu_seconds": t_cpu, "cpu_images_per_s_%dx16_threads" % workers: n / t_cpu,
           "consistency_abs_diff": {"median": float(np.median(d)), "p99": float(np.quantile(d, 0.99)), "max": float(d.max()),
                                    "images_over_1e-4": int((d > 1e-4).sum())},
           "top_%d_same_set" % k: int(len(set(np.argsort(gpu, kind="stable")[:k]) & set(np.argsort(cpu, kind="stable")[:k])))}
    print(json.dumps(out))
    if out_path:
        json.dump(out, open(out_path, "w"), indent=1)
