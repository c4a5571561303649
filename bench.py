#!/usr/bin/env python
"""bench.py -- CALD consistency sweep throughput on MI355X (unlabeled images scored / second).

A "step" is one pass of the hot path (cald_sweep: reference view + 3 augmented views per image,
detector forward x4, consistency scoring) over one batch of synthetic VOC-shaped images that are
already resident in HBM.  Workload = BASELINE.json configs[1]: Faster R-CNN ResNet-50 FPN, 21
classes, min/max size 600/1000, augmentations flip / cut_out / smaller_resize, seeded pseudo-trained
weights (cald_amd/synth.py), float32 (exact fp32 MFMA).

    python bench.py --gpus N --steps K --warmup W
For N > 1 launch with torch.distributed.run (one rank per GPU, RCCL); the pool is sharded by position
(no data-path collective) and one all-gather of the per-image scores closes the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2516.6  # v_mfma_f32_32x32x16_f16: 32 cycles / SIMD -> 16 x the f32-input rate (dense)


def cpu_baseline(sd, pool, augs, budget_s=12.0, max_images=24):
    """The reference-shaped PyTorch-CPU port (oracle/torch_port.py) on a bounded sample of the same workload."""
    import torch
    from oracle import torch_port
    model = torch_port.TorchFRCNN(sd, 21, 50, 600, 1000)
    model.forward(pool[0])          # warm-up view (oneDNN primitive creation), not timed
    # batch-1 convolutions do not scale to every core of a big host: use the thread count that is fastest here
    default_threads = torch.get_num_threads()
    best = (None, 1e30)
    for nt in sorted({default_threads, max(1, default_threads // 2), max(1, default_threads // 4), max(1, default_threads // 8)}, reverse=True):
        torch.set_num_threads(nt)
        model.forward(pool[0])
        t = time.time(); model.forward(pool[0]); t = time.time() - t
        if t < best[1]:
            best = (nt, t)
    torch.set_num_threads(best[0])
    n, t0 = 0, time.time()
    while n < max_images and (n == 0 or time.time() - t0 < budget_s):
        torch_port.get_uncertainty(model, [pool[n]], augs, 21, bp=1.3, base_seed=0, positions=[n])
        n += 1
    dt = time.time() - t0
    used = torch.get_num_threads()
    torch.set_num_threads(default_threads)
    return {"value": n / dt, "unit": "images/s", "cores": used, "kind": "port",
            "sample": "%d synthetic VOC-shaped image(s) x 4 views, batch-1 sequential torch-CPU fp32 forwards + python/scipy "
                      "scoring loop (oracle/torch_port.py), %.1f s; thread count chosen as the fastest of {T, T/2, T/4, T/8}" % (n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-images", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", default="frcnn", choices=["frcnn", "frcnn101", "retinanet"],
                    help="frcnn = the headline workload (BASELINE configs[1]); others are informational runs of configs[2]/[4]")
    ap.add_argument("--shape", default="voc", choices=["voc", "coco"])
    ap.add_argument("--augs", default="FCD", help="letters of cald_train.py --augs (F C D R G S)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "f16x3"],
                    help="fp32 = exact (headline, bit-identical to the oracle); f16x3 = informational split-fp16 MFMA path")
    args = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback for the product path"
    # dry-run aid for 1-GPU boxes: CALD_BENCH_SHARE_GPU=1 puts every rank on cuda:0 with the gloo backend
    share = os.environ.get("CALD_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from cald_amd import _ffi, detector, synth, sweep
    B, K, Wm = args.batch_images, args.steps, args.warmup
    letters = {"F": "flip", "C": "cut_out", "D": "smaller_resize", "R": "rotation", "G": "ga", "S": "sp"}
    augs = [letters[ch] for ch in args.augs]
    ncls = 21 if args.shape == "voc" else 91
    mn, mx = (600, 1000) if args.shape == "voc" else (800, 1333)
    headline = (args.model == "frcnn" and args.shape == "voc" and args.augs == "FCD" and args.precision == "fp32")
    if args.model == "retinanet":
        sd = synth.pseudo_trained_retinanet(ncls, 50, seed=0)
        model = detector.retinanet_resnet50_fpn_cal(num_classes=ncls, min_size=mn, max_size=mx, precision=args.precision)
    else:
        depth = 101 if args.model == "frcnn101" else 50
        sd = synth.pseudo_trained_frcnn(ncls, depth, seed=0)
        model = (detector.fasterrcnn_resnet101_fpn_feature if depth == 101 else detector.fasterrcnn_resnet50_fpn_feature)(
            num_classes=ncls, min_size=mn, max_size=mx, precision=args.precision)
    model = model.to("cuda:%d" % local_rank)
    model.load_state_dict(sd)
    model.eval()

    # distinct images per rank, resident in HBM before the timed region; rank r owns pool positions p % world == r
    n_local = B * min(K + Wm, 2)
    sizes = synth.pool_sizes(n_local * world, args.shape, 0)
    positions = [rank + world * i for i in range(n_local)]
    host_pool = [synth.synth_image(p, *sizes[p]) for p in positions]
    dev_pool = [torch.from_numpy(im).cuda() for im in host_pool]
    torch.cuda.synchronize()

    def step(s):
        lo = (s * B) % n_local
        idx = [(lo + j) % n_local for j in range(B)]
        return sweep.sweep_device_images(model, [dev_pool[i] for i in idx], [positions[i] for i in idx], augs,
                                         bp=1.3, base_seed=0, batch_images=B)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if Wm == 0:
        step(0)       # one-time initialisation (code objects, workspace arena) is model build, not a step
    for s in range(Wm):
        step(s)
    L, ctx = _ffi.lib(), detector.get_ctx(local_rank)
    _ffi.check(L.cald_profile_enable(ctx, 1))     # HIP events around every conv/linear launch on the launch stream
    barrier()
    t0 = time.time()
    last = None
    for s in range(K):
        last = step(Wm + s)
    if world > 1:   # the one RCCL all-gather of (position, consistency, cls_corr) rows
        idx = [(((Wm + K - 1) * B) % n_local + j) % n_local for j in range(B)]
        sweep.allgather_scores([positions[i] for i in idx], last[0], last[1], world * n_local)
    barrier()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    import ctypes as C
    gm, gf, tot = C.c_double(), C.c_double(), C.c_double()
    nl = C.c_int64()
    _ffi.check(L.cald_profile_read(ctx, C.byref(gm), C.byref(gf), C.byref(nl), C.byref(tot)))
    if os.environ.get("CALD_PROFILE_DUMP"):
        _ffi.check(L.cald_profile_dump(ctx, os.environ["CALD_PROFILE_DUMP"].encode()))
    _ffi.check(L.cald_profile_enable(ctx, 0))

    if rank == 0:
        # HBM bytes per GEMM launch from the rocprofv3 PMC passes of this same command (tools/profile_gpu.sh ->
        # tools/summarize_profile.py -> profiles/*_pmc.json; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KB units)
        traffic = None
        try:
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc.json"))
            if cands:
                traffic = json.load(open(os.path.join(ROOT, "profiles", cands[-1])))["conv_mfma"]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        images = world * K * B
        achieved = gf.value / (gm.value * 1e-3) / 1e12 if gm.value > 0 else 0.0
        peak = F32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else F16_MFMA_PEAK_TFLOPS
        out = {
            "metric": "unlabeled images scored/sec (CALD consistency sweep)", "value": images / dt, "unit": "images/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f16x3 (fp16 hi+lo split operands, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[1]: Faster R-CNN ResNet-50 FPN, VOC2012-shaped synthetic pool, "
                                    "3 augs (flip/cut_out/smaller_resize), 21 classes, min/max 600/1000, seeded pseudo-trained weights")
                       if headline else "informational: model=%s shape=%s augs=%s classes=%d min/max %d/%d precision=%s" % (args.model, args.shape, args.augs, ncls, mn, mx, args.precision),
                       "images_per_step_per_gpu": B, "views_per_image": 1 + len(sweep.expand_augs(augs)), "parallelism": "pool sharded by position, dp%d" % world},
            "roofline": {"bound": "mfma",
                         "kernel": ("conv_p4_kernel + conv_mfma_f32_kernel (implicit-GEMM conv + linear, v_mfma_f32_32x32x2_f32)" if args.precision == "fp32"
                                    else "conv_h3_kernel (3 x v_mfma_f32_32x32x16_f16 per product; algorithmic flops counted once) + exact kernels for uncovered shapes"),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic if args.precision == "fp32" else None,
                         "launches": int(nl.value), "avg_launch_ms": gm.value / max(1, nl.value),
                         "gemm_ms_per_step": gm.value / K, "algorithmic_gflop_per_launch": gf.value / max(1, nl.value) / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline and headline:
            out["cpu_baseline"] = cpu_baseline(sd, host_pool, augs)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
            # informational second line, NOT the headline: the opt-in split-fp16 MFMA mode (BASELINE configs[4]'s
            # "fp16 MFMA path") on the same workload.  Parity bar of that mode: 1e-4 / identical ranking, not bit-exact.
            fast = (detector.fasterrcnn_resnet50_fpn_feature(num_classes=ncls, min_size=mn, max_size=mx, precision="f16x3")
                    .to("cuda:%d" % local_rank))
            fast.load_state_dict(sd)
            fast.eval()
            idx = list(range(B))
            run = lambda: sweep.sweep_device_images(fast, [dev_pool[i] for i in idx], [positions[i] for i in idx], augs,
                                                    bp=1.3, base_seed=0, batch_images=B)
            fc, _ = run()
            torch.cuda.synchronize(); tf = time.time()
            for _ in range(2):
                run()
            torch.cuda.synchronize(); tf = time.time() - tf
            ec, _ = sweep.sweep_device_images(model, [dev_pool[i] for i in idx], [positions[i] for i in idx], augs,
                                              bp=1.3, base_seed=0, batch_images=B)
            out["f16x3_mode"] = {"value": 2 * B / tf, "unit": "images/s", "dtype": "fp16 hi+lo split operands, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulate",
                                 "headline": False, "max_abs_consistency_diff_vs_exact": float(np.abs(fc - ec).max()),
                                 "same_ranking_as_exact": bool(np.array_equal(np.argsort(fc, kind="stable"), np.argsort(ec, kind="stable")))}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
