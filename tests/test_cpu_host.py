"""CPU-only tests: C-ABI surface, host-side logic of the product, selection stage, the sharded
all-gather (gloo, world_size 2) and the oracle's float cross-checks against plain torch fp32."""
import ctypes as C
import os
import re
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    hdr = open(os.path.join(ROOT, "include", "cald_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cald_[a-z_0-9]+)\s*\(", hdr)))


def test_library_loads_and_exports_every_declared_symbol():
    from cald_amd import _ffi
    L = _ffi.lib()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libcaldhip.so does not export %s" % n
    assert sorted(_ffi.SIGNATURES) == names, "cald_amd/_ffi.py must bind exactly the symbols of include/cald_hip.h"
    assert L.cald_version() >= 100


def test_product_never_touches_the_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "cald_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert "cald_oracle" not in txt and "#include \"../../oracle" not in txt, f


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cald_amd import detector
    with pytest.raises(RuntimeError):
        detector.get_ctx()


def test_host_cutout_and_transform_size_match_oracle_and_reference(oracle, golden):
    from cald_amd import _ffi
    L = _ffi.lib()
    g = golden("helpers")
    for i in range(4):
        img, boxes = g["img%d" % i], np.ascontiguousarray(g["boxes%d" % i], np.float32)
        H, W, _ = img.shape
        for s in (11, 12, 13):
            rects = np.zeros(16, np.int32); n = C.c_int()
            _ffi.check(L.cald_op_cutout_rects(s, H, W, boxes.shape[0], _ffi.ptr(boxes), 2, _ffi.ptr(rects, _ffi.c_i), C.byref(n)))
            got = img.copy()
            for (l, t, r, b) in rects[:4 * n.value].reshape(-1, 4):
                got[t:b, l:r] = 0
            np.testing.assert_array_equal(got, g["cutout%d_%d_img" % (i, s)])      # the reference's own output
    for (H, W) in [(375, 500), (500, 375), (333, 500), (500, 334), (480, 640), (427, 640), (100, 3000), (31, 37)]:
        for (mn, mx) in [(600, 1000), (800, 1333)]:
            v = [C.c_int() for _ in range(4)]
            _ffi.check(L.cald_op_transform_size(H, W, mn, mx, *[C.byref(x) for x in v]))
            assert tuple(x.value for x in v) == oracle.transform_size(H, W, mn, mx)
    assert L.cald_op_transform_size(0, 5, 600, 1000, *[C.byref(C.c_int()) for _ in range(4)]) != 0
    assert b"bad sizes" in L.cald_last_error()


def test_selection_matches_reference_golden(golden):
    from cald_amd import sweep
    import torch
    g = golden("selection")
    for case in range(4):
        cls_corrs = g["cls_corrs%d" % case]
        labels = g["labels%d" % case]
        loader = [(None, [{"labels": torch.from_numpy(row[row >= 0])}]) for row in labels]
        sel = sweep.cls_kldiv(loader, list(cls_corrs), int(g["budget%d" % case]), 0, uniform=bool(g["uniform%d" % case]))
        np.testing.assert_array_equal(np.array(sel), g["sel%d" % case])
    np.testing.assert_array_equal(np.argsort(g["argsort_in"]), g["argsort_out"])


def test_selection_edge_cases_match_reference_golden(golden):
    """Ties in the JS vector, zero-sum candidates beyond the budget, budget > candidates, 90-wide vectors, --uniform:
    cls_kldiv as executed from /root/reference (oracle/make_golden_selection.py) vs the product's computed-once form."""
    from cald_amd import sweep
    import torch
    g = golden("selection_more")
    for case in range(int(g["n_cases"])):
        labels = g["labels%d" % case]
        loader = [(None, [{"labels": torch.from_numpy(row[row >= 0])}]) for row in labels]
        sel = sweep.cls_kldiv(loader, list(g["cls_corrs%d" % case]), int(g["budget%d" % case]), 0, uniform=bool(g["uniform%d" % case]))
        np.testing.assert_array_equal(np.array(sel, np.int64), g["sel%d" % case])


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _gather_worker(rank, world, port, pool_size, q):
    import torch.distributed as dist
    from cald_amd import sweep
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    pos = [p for p in range(pool_size) if p % world == rank]
    cons = np.array([0.25 + p * 0.5 for p in pos], np.float64)
    cls = np.stack([np.arange(20, dtype=np.float64) * 0.01 + p for p in pos]) if pos else np.zeros((0, 20))
    fc, fcl = sweep.allgather_scores(pos, cons, cls, pool_size)
    q.put((rank, fc, fcl))
    dist.destroy_process_group()


@pytest.mark.parametrize("pool_size", [7, 8, 1])
def test_sharded_allgather_is_rank_count_invariant(pool_size):
    """world_size-2 gloo run of the N>1 path: every rank ends with the full, correctly ordered vectors."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, pool_size, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    want_c = np.array([0.25 + p * 0.5 for p in range(pool_size)])
    want_cls = np.stack([np.arange(20, dtype=np.float64) * 0.01 + p for p in range(pool_size)])
    for _, fc, fcl in res:
        np.testing.assert_array_equal(fc, want_c)
        np.testing.assert_array_equal(fcl, want_cls)


def test_oracle_conv_chain_vs_torch_fp32(oracle):
    import torch
    import torch.nn.functional as F
    rs = np.random.RandomState(0)
    for (H, W, Cin, Cout, K, stride, pad) in [(23, 31, 4, 64, 7, 2, 3), (20, 26, 64, 64, 3, 1, 1), (20, 26, 128, 128, 3, 2, 1),
                                              (17, 19, 256, 512, 1, 2, 0), (9, 11, 256, 15, 1, 1, 0)]:
        x = rs.randn(H, W, Cin).astype(np.float32)
        w = (rs.randn(Cout, Cin, K, K) * np.sqrt(2.0 / (Cin * K * K))).astype(np.float32)
        b = rs.randn(Cout).astype(np.float32)
        wk = np.ascontiguousarray(w.transpose(2, 3, 1, 0).reshape(-1, Cout))
        got = oracle.conv2d(x, wk, K, K, stride, pad, bias=b, relu=True)
        want = F.relu(F.conv2d(torch.from_numpy(x).permute(2, 0, 1)[None], torch.from_numpy(w), torch.from_numpy(b), stride, pad))
        np.testing.assert_allclose(got, want[0].permute(1, 2, 0).numpy(), rtol=1e-4, atol=1e-4)   # tolerance of BASELINE.json
    x = rs.randn(11, 13, 64).astype(np.float32)
    np.testing.assert_array_equal(oracle.maxpool3x3s2(x), F.max_pool2d(torch.from_numpy(x).permute(2, 0, 1)[None], 3, 2, 1)[0].permute(1, 2, 0).numpy())


def test_oracle_detector_vs_torch_port_small(oracle):
    """The C oracle's full forward against the plain-torch fp32 port (float tolerance 1e-4; same detections)."""
    from cald_amd import synth
    from oracle import torch_port
    sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
    P = oracle.prepare_frcnn(sd, 21, 50)
    tm = torch_port.TorchFRCNN(sd, 21, 50, 160, 256)
    img = synth.make_pool(2, "voc", 0, scale=0.3)[1]
    k1, k2 = {}, {}
    a = tm.forward(img, keep=k1)
    b = oracle.frcnn_forward(P, img, 160, 256, keep=k2)
    np.testing.assert_allclose(k1["input"][..., :3], k2["input"][..., :3], atol=1e-5)
    for i in range(5):
        scale = float(np.abs(k2["fpn"][i]).max())
        assert float(np.abs(k1["fpn"][i] - k2["fpn"][i]).max()) <= 1e-4 * max(1.0, scale)
    assert a["boxes"].shape == b["boxes"].shape
    np.testing.assert_allclose(a["scores"], b["scores"], atol=1e-4)
    np.testing.assert_array_equal(a["labels"], b["labels"])
    np.testing.assert_allclose(a["boxes"], b["boxes"], atol=2e-2)


def test_helper_api_mirror_matches_reference_golden(golden):
    """cald_amd.cald_helper keeps the reference's helper names/signatures (CPU-computable parts)."""
    import torch
    from cald_amd import cald_helper as ch
    g = golden("helpers")
    for i in range(4):
        img, boxes = g["img%d" % i], torch.from_numpy(g["boxes%d" % i])
        fi, fb = ch.HorizontalFlip(torch.from_numpy(img), boxes)
        np.testing.assert_array_equal(fb.numpy(), g["flip_boxes%d" % i])
        np.testing.assert_array_equal((fi * 255).round().to(torch.uint8).permute(1, 2, 0).numpy(), g["flip_img%d" % i])
        np.testing.assert_array_equal(ch.intersect(boxes, torch.from_numpy(g["boxes_b%d" % i])).numpy(), g["intersect%d" % i])
        for s in (11, 12, 13):
            ci = ch.cutout(torch.from_numpy(img), boxes, None, 2, seed=s)
            np.testing.assert_array_equal((ci * 255).round().to(torch.uint8).permute(1, 2, 0).numpy(), g["cutout%d_%d_img" % (i, s)])


class _FakeModel:
    num_classes = 21

    def eval(self):
        return self

    def handle(self):
        return None


def _fake_sweep(task_model, images, positions, augs, bp=1.3, base_seed=0, batch_images=64):
    """Stands in for the GPU sweep: a deterministic function of (pool position, image bytes)."""
    cons = np.array([p * 0.125 + float(im.float().mean()) / 255.0 for p, im in zip(positions, images)], np.float64)
    cls = np.stack([np.full(20, p, np.float64) for p in positions]) if positions else np.zeros((0, 20))
    return cons, cls


def _sharded_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from cald_amd import sweep
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    sweep.sweep_device_images = _fake_sweep
    rs = np.random.RandomState(0)
    loader = [((torch.from_numpy((rs.rand(8, 9, 3) * 255).astype(np.uint8)),), (None,)) for _ in range(11)]
    cons, cls = sweep.get_uncertainty(_FakeModel(), loader, ["flip"], 21, rank=rank, world_size=world)
    q.put((rank, np.array(cons), np.stack(cls)))
    dist.destroy_process_group()


def test_get_uncertainty_chunked_upload_is_invisible(monkeypatch):
    """Loader-fed sweeps upload and score `chunk_images` at a time; the returned lists do not depend on the chunking."""
    import torch
    from cald_amd import sweep
    monkeypatch.setattr(sweep, "sweep_device_images", _fake_sweep)
    rs = np.random.RandomState(1)
    loader = [((torch.from_numpy((rs.rand(6, 7, 3) * 255).astype(np.uint8)),), (None,)) for _ in range(11)]
    a = sweep.get_uncertainty(_FakeModel(), loader, ["flip"], 21)
    b = sweep.get_uncertainty(_FakeModel(), loader, ["flip"], 21, chunk_images=3)
    assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1], b[1])) and len(a[0]) == 11
    assert sweep.get_uncertainty(_FakeModel(), [], ["flip"], 21) == ([], [])


def test_get_uncertainty_sharded_equals_single_rank():
    """The N>1 path of get_uncertainty (strided shard + one all-gather) returns, on every rank, exactly
    what the 1-rank run returns, in loader order."""
    import torch
    import torch.multiprocessing as mp
    from cald_amd import sweep
    rs = np.random.RandomState(0)
    loader = [((torch.from_numpy((rs.rand(8, 9, 3) * 255).astype(np.uint8)),), (None,)) for _ in range(11)]
    orig = sweep.sweep_device_images
    try:
        sweep.sweep_device_images = _fake_sweep
        c1, k1 = sweep.get_uncertainty(_FakeModel(), loader, ["flip"], 21)
    finally:
        sweep.sweep_device_images = orig
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for _, cons, cls in res:
        np.testing.assert_array_equal(cons, np.array(c1))
        np.testing.assert_array_equal(cls, np.stack(k1))


class _CountingDataset:
    """Stands in for dataset_aug (cald_train.py:288): records which indices were ever decoded in this process."""

    def __init__(self, n):
        rs = np.random.RandomState(0)
        self.images = [(rs.rand(8, 9, 3) * 255).astype(np.uint8) for _ in range(n)]
        self.touched = []

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        import torch
        self.touched.append(int(i))
        return torch.from_numpy(self.images[i]), None


def _rank_local_worker(rank, world, port, subset, q):
    import torch
    import torch.distributed as dist
    from torch.utils.data import DataLoader
    from cald_amd import sweep
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    sweep.sweep_device_images = _fake_sweep
    ds = _CountingDataset(16)
    # cald_train.py:434 with the rank's own sampler: DataLoader(dataset_aug, batch_size=1, sampler=..., collate_fn=tuple-zip)
    loader = DataLoader(ds, batch_size=1, sampler=sweep.ShardedSequentialSampler(subset, rank, world), num_workers=0,
                        collate_fn=lambda b: tuple(zip(*b)))
    cons, cls = sweep.get_uncertainty(_FakeModel(), loader, ["flip"], 21, rank=rank, world_size=world, loader_is_sharded=True)
    q.put((rank, np.array(cons), np.stack(cls), sorted(ds.touched)))
    dist.destroy_process_group()


def test_rank_local_input_sharding_touches_only_own_images():
    """SURVEY 8e / cald_train.py:434: with a per-rank sampler every rank pulls (decodes) ONLY its strided shard of
    `subset`, and the gathered vectors still come back in `subset` order, equal to the single-process run."""
    import torch
    import torch.multiprocessing as mp
    from torch.utils.data import DataLoader
    from cald_amd import sweep
    subset = [11, 3, 7, 0, 15, 2, 9, 5, 12, 6, 1]          # shuffled unlabeled indices (cald_train.py:427)
    ds = _CountingDataset(16)
    loader = DataLoader(ds, batch_size=1, sampler=sweep.ShardedSequentialSampler(subset, 0, 1), num_workers=0,
                        collate_fn=lambda b: tuple(zip(*b)))
    orig = sweep.sweep_device_images
    try:
        sweep.sweep_device_images = _fake_sweep
        c1, k1 = sweep.get_uncertainty(_FakeModel(), loader, ["flip"], 21)
    finally:
        sweep.sweep_device_images = orig
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_rank_local_worker, args=(r, world, port, subset, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, cons, cls, touched in res:
        np.testing.assert_array_equal(cons, np.array(c1))
        np.testing.assert_array_equal(cls, np.stack(k1))
        assert touched == sorted(subset[rank::world]), (rank, touched)        # never another rank's images
    with pytest.raises(ValueError):                                           # a shard that is not the strided one is refused
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
        try:
            sweep.allgather_scores([1, 0], np.zeros(2), np.zeros((2, 20)), 2)
        finally:
            dist.destroy_process_group()


def test_voc_results_wire_format_matches_reference(golden, tmp_path):
    """engine.write_voc_results_file == detection/voc_eval.py:188-222 on the same detections."""
    import torch
    from cald_amd import engine
    g = golden("voc_results")
    names, classes = [str(n) for n in g["names"]], [str(c) for c in g["classes"]]
    all_boxes = [[] for _ in classes]
    for ii in range(len(names)):
        for c in range(len(classes)):
            key = "d%d_%d" % (ii, c)
            all_boxes[c].append([torch.from_numpy(g[key])] if key in g.files else [])
    out = engine.write_voc_results_file(all_boxes, names, "res", classes, root=str(tmp_path))
    for key in g.files:
        if key.startswith("file_"):
            assert open(os.path.join(out, key[5:])).read() == str(g[key]), key


def test_coco_results_wire_format_matches_reference(golden):
    """engine.coco_results == CocoEvaluator.prepare_for_coco_detection (detection/coco_eval.py:76-98) and
    engine.coco_predictions builds the {image_id: output} dict coco_evaluate feeds it (detection/engine.py:199-205)."""
    import json
    import torch
    from cald_amd import engine
    g = golden("coco_results")
    preds = {int(i): {k: torch.from_numpy(g["p%d_%s" % (i, k)]) for k in ("boxes", "scores", "labels")} for i in g["ids"]}
    assert json.dumps(engine.coco_results(preds)) == str(g["json"])

    class Model:                       # stands in for the HIP detector: returns the stored dicts in call order
        def eval(self): return self
        def __call__(self, images): return [dict(preds[int(im[0, 0, 0])], features=None) for im in images]
    loader = [((torch.full((3, 4, 4), float(i)),), ({"image_id": torch.tensor(int(i))},)) for i in g["ids"]]
    got = engine.coco_predictions(Model(), loader, batch_views=2)
    assert list(got.keys()) == [int(i) for i in g["ids"]] and all("features" not in o for o in got.values())
    assert json.dumps(engine.coco_results(got)) == str(g["json"])


def test_wrapping_a_reference_torch_model_reads_its_configuration():
    """get_uncertainty() accepts the torch model cald_train.py already holds: constructor arguments come from
    the module attributes torchvision exposes, weights from state_dict()."""
    import torch
    from types import SimpleNamespace
    from cald_amd import detector, synth

    class FakeTorchFRCNN:
        transform = SimpleNamespace(min_size=(600,), max_size=1000)
        roi_heads = SimpleNamespace(score_thresh=0.05, nms_thresh=0.5, detections_per_img=100)
        rpn = SimpleNamespace(_pre_nms_top_n={"training": 2000, "testing": 1000}, _post_nms_top_n={"training": 2000, "testing": 1000}, nms_thresh=0.7)

        def __init__(self, sd):
            self._sd = {k: torch.from_numpy(v) for k, v in sd.items()}

        def state_dict(self):
            return self._sd
    m = detector.from_torch_module(FakeTorchFRCNN(synth.pseudo_trained_frcnn(21, 50, 0)))
    assert (m.cfg.arch, m.cfg.depth, m.cfg.num_classes, m.cfg.min_size, m.cfg.max_size) == (0, 50, 21, 600, 1000)
    assert (m.cfg.detections_per_img, m.cfg.rpn_pre_nms_top_n, m.cfg.rpn_post_nms_top_n) == (100, 1000, 1000)

    class FakeTorchRetina(FakeTorchFRCNN):
        score_thresh, nms_thresh, detections_per_img = 0.05, 0.5, 300
    r = detector.from_torch_module(FakeTorchRetina(synth.pseudo_trained_retinanet(21, 50, 0)))
    assert (r.cfg.arch, r.cfg.num_classes, r.cfg.detections_per_img) == (1, 21, 300)


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    import ctypes as C
    from cald_amd import _ffi
    L = _ffi.lib()
    assert L.cald_ctx_create(0, None, None) != 0 and b"null" in L.cald_last_error()
    assert L.cald_model_create(None, None, None) != 0
    assert L.cald_ctx_sync(None) != 0
    n = C.c_int()
    rects = (C.c_int * 16)()
    assert L.cald_op_cutout_rects(1, 10, 10, 0, None, 9, rects, C.byref(n)) != 0      # cut_num > 4
    assert L.cald_ctx_destroy(None) == 0 and L.cald_model_destroy(None) == 0


def _materialise_voc_tree(g, root):
    """The synthetic VOCdevkit tree + results files of tests/golden/voc_eval.npz (annotation table and file text are data)."""
    classes = [str(c) for c in g["classes"]]
    names = [str(n) for n in g["names"]]
    base = os.path.join(root, "VOCdevkit", "VOC2012")
    os.makedirs(os.path.join(base, "ImageSets", "Main")); os.makedirs(os.path.join(base, "Annotations")); os.makedirs(os.path.join(root, "res"))
    with open(os.path.join(base, "ImageSets", "Main", "test.txt"), "w") as f:
        f.write("".join(n + "\n" for n in names))
    for i, n in enumerate(names):
        xml = "<annotation>"
        for (ii, c, diff, x0, y0, x1, y1) in g["objects"]:
            if ii == i:
                xml += ("<object><name>%s</name><difficult>%d</difficult><bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax>"
                        "<ymax>%d</ymax></bndbox></object>" % (classes[c], diff, x0, y0, x1, y1))
        with open(os.path.join(base, "Annotations", n + ".xml"), "w") as f:
            f.write(xml + "</annotation>")
    for cls in classes[1:]:
        with open(os.path.join(root, "res", "det_test_%s.txt" % cls), "w") as f:
            f.write(str(g["det_%s" % cls]))
    return classes, os.path.join(base, "ImageSets/Main/test.txt"), os.path.join(base, "Annotations/{:s}.xml")


def test_voc_ap_matches_reference_golden(golden, tmp_path, capsys):
    """SURVEY 8f rank 1, AP half: cald_amd.voc_eval (annotations parsed once, overlaps computed once per class) returns
    exactly the rec / prec / ap arrays detection/voc_eval.py returned for every class x IoU threshold x metric, and
    do_python_eval prints the reference's table line (oracle/make_golden_voc_eval.py)."""
    from types import SimpleNamespace
    from cald_amd import voc_eval as ve
    g = golden("voc_eval")
    root = str(tmp_path)
    classes, imagesetfile, annopath = _materialise_voc_tree(g, root)
    ann = ve.VocAnnotations(imagesetfile, annopath)
    with np.errstate(all="ignore"):
        for cls in classes[1:]:
            fn = os.path.join(root, "res", "det_test_%s.txt" % cls)
            for use07 in (False, True):
                multi = ve.voc_eval_thresholds(cls, fn, ann, [float(t) for t in g["ious"]], use_07_metric=use07)
                for t, (mrec, mprec, map_) in zip(g["ious"], multi):
                    key = "%s_%d_%d" % (cls, int(round(float(t) * 100)), int(use07))
                    rec, prec, ap = ve.voc_eval(cls, fn, imagesetfile, annopath, ovthresh=float(t), use_07_metric=use07)
                    for got in ((rec, prec, ap), (mrec, mprec, map_)):
                        np.testing.assert_array_equal(got[0], g["rec_" + key])
                        np.testing.assert_array_equal(got[1], g["prec_" + key])
                        np.testing.assert_array_equal(np.float64(got[2]), g["ap_" + key])
        for key, ncls in (("python_eval_stdout", 5), ("python_eval_stdout_3cls", 4)):
            loader = SimpleNamespace(dataset=SimpleNamespace(root=root, image_set="test",
                                                             _transforms=SimpleNamespace(transforms=[SimpleNamespace(CLASSES=tuple(classes[:ncls]))])))
            capsys.readouterr()
            res = ve.do_python_eval(loader, "2012", "res", root=root)
            assert capsys.readouterr().out == str(g[key])
            assert res["line"] == str(g[key]).splitlines()[1]
    assert 0.0 < res["AP50"] < 1.0


def test_voc_dataset_conversion_matches_reference_golden(golden, tmp_path):
    """SURVEY 8f rank 2 (data format feeding cls_kldiv and voc_evaluate): cald_amd.voc_utils parses VOC annotation XML into the
    torchvision dict layout and converts it exactly as detection/voc_utils.py:16-44 did (oracle/make_golden_voc_utils.py):
    same values AND dtypes, for the object-list layout and the older bare-dict single-object layout; then the same through
    VOCDetection over a VOCdevkit tree on disk (ids, targets without image decode, images through PIL)."""
    import xml.etree.ElementTree as ET
    import torch
    from PIL import Image
    from cald_amd import voc_utils as vu
    g = golden("voc_utils")
    assert tuple(str(c) for c in g["classes"]) == vu.ConvertVOCtoCOCO.CLASSES
    n = int(g["n"])
    base = tmp_path / "VOCdevkit" / "VOC2012"
    for d in ("ImageSets/Main", "Annotations", "JPEGImages"):
        (base / d).mkdir(parents=True)
    stems = []
    rs = np.random.RandomState(0)
    for i in range(n):
        xml = str(g["xml_%d" % i])
        anno = vu.parse_voc_xml(ET.fromstring(xml))["annotation"]
        assert isinstance(anno["object"], list)
        if bool(g["bare_%d" % i]):
            anno["object"] = anno["object"][0]
        _, t = vu.ConvertVOCtoCOCO()("img", dict(image_id=i, annotations=anno))
        for k, dt in (("boxes", torch.float32), ("labels", torch.int64), ("ishard", torch.int64), ("name", torch.int8)):
            assert t[k].dtype == dt
            np.testing.assert_array_equal(t[k].numpy(), g["%s_%d" % (k, i)])
        stem = "".join(chr(int(c)) for c in g["name_%d" % i])
        stems.append(stem)
        (base / "Annotations" / (stem + ".xml")).write_text(xml)
        Image.fromarray(rs.randint(0, 256, (20 + i, 30, 3), dtype=np.uint8)).save(str(base / "JPEGImages" / (stem + ".jpg")), quality=90)
    (base / "ImageSets" / "Main" / "trainval.txt").write_text("".join(s + "\n" for s in stems))
    ds = vu.get_voc2012(str(tmp_path), "trainval", vu.ToTensor())
    assert len(ds) == n and ds.root == str(tmp_path) and ds.image_set == "trainval"
    assert ds._transforms.transforms[0].CLASSES == vu.VOC_CLASSES
    for i in (0, 7, n - 1):
        img, t = ds[i]
        assert img.dtype == torch.float32 and tuple(img.shape) == (3, 20 + i, 30) and float(img.max()) <= 1.0
        t2 = ds.target(i)
        for k in ("boxes", "labels", "ishard", "name"):
            np.testing.assert_array_equal(t[k].numpy(), g["%s_%d" % (k, i)])
            assert torch.equal(t[k], t2[k])
    labeled = ds.label_loader([1, 2, 3])
    assert [int(x) for _, (t,) in labeled for x in t["labels"]] == [int(x) for i in (1, 2, 3) for x in g["labels_%d" % i]]


def test_sampler_draws_are_uniform_subsets():
    """cald_amd.train.choose_k (the O(k) stand-in for BalancedPositiveNegativeSampler's randperm(n)[:k]):
    k distinct indices in range, reproducible from the generator, every element equally likely."""
    import torch
    from cald_amd.train import choose_k
    g = torch.Generator().manual_seed(3)
    for n, k in ((100000, 128), (1000, 128), (300, 256), (40, 64), (5, 0), (0, 8)):
        idx = choose_k(n, k, g).numpy()
        assert len(idx) == min(n, k) and len(set(idx.tolist())) == len(idx)
        assert len(idx) == 0 or (idx.min() >= 0 and idx.max() < n)
    a = choose_k(100000, 128, torch.Generator().manual_seed(9)); b = choose_k(100000, 128, torch.Generator().manual_seed(9))
    assert torch.equal(a, b)
    counts = np.zeros(50)
    g = torch.Generator().manual_seed(0)
    for _ in range(4000):
        counts[choose_k(50, 5, g).numpy()] += 1
    assert abs(counts / 4000 - 0.1).max() < 0.02          # each element is drawn with probability k / n = 0.1


def test_training_transform_flip_matches_reference_golden(golden):
    """cald_amd.voc_utils.RandomHorizontalFlip (the training transform of get_transform(train=True)) == the imported
    detection/transforms.py class on ten draws of Python's `random` (seed 11): same flips, same mirrored boxes."""
    import random
    import torch
    from cald_amd import voc_utils as vu
    g = golden("voc_utils")
    flip = vu.RandomHorizontalFlip(0.5)
    random.seed(11)
    flipped = 0
    for k in range(10):
        im, t = flip(torch.from_numpy(g["flip_image_in"].copy()), {"boxes": torch.from_numpy(g["flip_boxes_in"].copy())})
        np.testing.assert_array_equal(im.numpy(), g["flip_images_out"][k])
        np.testing.assert_array_equal(t["boxes"].numpy(), g["flip_boxes_out"][k])
        flipped += not np.array_equal(g["flip_images_out"][k], g["flip_image_in"])
    assert 0 < flipped < 10
    tr = vu.get_transform(train=True)
    assert isinstance(tr.transforms[0], vu.ToTensor) and isinstance(tr.transforms[1], vu.RandomHorizontalFlip) and len(vu.get_transform(False).transforms) == 1


def test_resident_training_loader_flips_like_the_reference_transform(golden):
    """voc_utils.ResidentTrainLoader (training batches from an HBM-resident pool) applies RandomHorizontalFlip to the uint8 HWC pool
    image and the boxes: same decisions (one `random` draw per image in batch order) and the same mirrored boxes / pixels as the
    imported transform produced on its CHW float tensor (ten draws, seed 11); batches follow the batch sampler; targets are copies."""
    import random
    import torch
    from cald_amd import voc_utils as vu
    g = golden("voc_utils")
    chw = g["flip_image_in"]                                                     # 3 x 4 x 7 float
    hwc = torch.from_numpy(np.ascontiguousarray(np.round(chw * 100).astype(np.uint8).transpose(1, 2, 0)))      # same pattern as uint8 HWC
    boxes0 = torch.from_numpy(g["flip_boxes_in"].copy())

    class DS(object):
        def target(self, i):
            return {"boxes": boxes0, "labels": torch.tensor([1, 2]), "image_id": i}

    batches = [[0, 1, 2, 3], [4, 5, 6], [7, 8, 9]]
    loader = vu.ResidentTrainLoader(DS(), batches, indices=range(10), pool=[hwc] * 10, flip_prob=0.5)
    assert len(loader) == 3
    random.seed(11)
    k = 0
    for (images, targets), want in zip(loader, batches):
        assert len(images) == len(targets) == len(want)
        for img, t, i in zip(images, targets, want):
            assert t["image_id"] == i
            np.testing.assert_array_equal(t["boxes"].numpy(), g["flip_boxes_out"][k])
            np.testing.assert_array_equal(img.permute(2, 0, 1).numpy(), np.round(g["flip_images_out"][k] * 100).astype(np.uint8))
            k += 1
    assert k == 10
    np.testing.assert_array_equal(boxes0.numpy(), g["flip_boxes_in"])            # the dataset's tensors are not modified in place
    random.seed(11)
    assert all(torch.equal(im, hwc) for ims, _ in vu.ResidentTrainLoader(DS(), batches, range(10), pool=[hwc] * 10, flip_prob=0.0) for im in ims)


def test_training_operator_host_side_contracts():
    """The cald_train_* entry points that need no GPU: packed-weight geometry (what cald_amd/train.py allocates for) and argument
    validation -- a null context / bad mode is an error code with a message, never a crash, and nothing silently falls back."""
    from cald_amd import _ffi
    L = _ffi.lib()
    n = C.c_int64()
    # forward pack of a 3x3 256 -> 256 conv: K = 2304 rows (multiple of 16), N padded to 256: [wk | w4 | bias, scale, shift]
    assert L.cald_train_packed_floats(256, 256, 3, 3, 256, 0, C.byref(n)) == 0 and n.value == 2 * 2304 * 256 + 3 * 256
    # merged RPN head (15 outputs): N pads to 32; its data gradient contracts over 16 channels of dY and produces 256 columns
    assert L.cald_train_packed_floats(15, 256, 1, 1, 256, 0, C.byref(n)) == 0 and n.value == 2 * 256 * 32 + 3 * 32
    assert L.cald_train_packed_floats(15, 256, 1, 1, 16, 1, C.byref(n)) == 0 and n.value == 2 * 16 * 256 + 3 * 256
    # fc6 on RoIAlign rows: K = 49 * 256
    assert L.cald_train_packed_floats(1024, 256, 49, 1, 256, 2, C.byref(n)) == 0 and n.value == 2 * 12544 * 1024 + 3 * 1024
    assert L.cald_train_packed_floats(1024, 256, 49, 1, 256, 7, C.byref(n)) != 0 and b"bad arguments" in L.cald_last_error()
    assert L.cald_train_pack_conv(None, None, None, None, None, 8, 8, 1, 1, 8, 0, None) != 0
    assert L.cald_train_conv(None, 1, 8, 8, None, 8, None, 8, 8, 1, 1, 1, 0, 0, 0, None, None, 0, 0, None, None, 8) != 0
    assert L.cald_train_sgd(None, 10, None, None, None, 0.1, 0.9, 0.0, 1) != 0
    assert L.cald_train_focal_loss(None, 1, None, 9, 21, 192, None, None, None, None, None, 0.25, 1.0, None, None) != 0


def test_coco_dataset_conversion_matches_reference_golden(golden, tmp_path):
    """SURVEY 8f rank 2 on the COCO side (configs[3] / [4]): cald_amd.coco_utils reads an instances_*.json tree directly and converts
    targets exactly as detection/coco_utils.py:49-100 did (oracle/make_golden_coco_utils.py): crowd objects dropped, xywh -> clamped
    xyxy, degenerate boxes removed, area / iscrowd of the non-crowd objects; the training-set filter keeps the same images."""
    import json
    import torch
    from PIL import Image
    from cald_amd import coco_utils as cu
    g = golden("coco_utils")
    n = int(g["n"])
    root = tmp_path
    (root / "train2017").mkdir(); (root / "val2017").mkdir(); (root / "annotations").mkdir()
    images, annotations = [], []
    rs = np.random.RandomState(1)
    for i in range(n):
        w, h = [int(v) for v in g["size_%d" % i]]
        name = "%012d.jpg" % (100 + i)
        images.append({"id": 100 + i, "file_name": name, "width": w, "height": h})
        annotations += json.loads(str(g["anno_%d" % i]))
        for split in ("train2017", "val2017"):
            Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(str(root / split / name), quality=85)
    rs.shuffle(images)                                                  # ids are sorted by the dataset, not taken in file order
    for split in ("train", "val"):
        (root / "annotations" / ("instances_%s2017.json" % split)).write_text(json.dumps({"images": images, "annotations": annotations, "categories": [{"id": c} for c in range(1, 91)]}))
    val = cu.get_coco(str(root), "val", None)
    assert len(val) == n and val.ids == [100 + i for i in range(n)]
    for i in range(n):
        img, t = val[i]
        assert img.size == tuple(int(v) for v in g["size_%d" % i])
        t2 = val.target(i)
        for k, dt in (("boxes", torch.float32), ("labels", torch.int64), ("area", None), ("iscrowd", None), ("image_id", torch.int64)):
            np.testing.assert_array_equal(t[k].numpy(), g["%s_%d" % (k, i)])
            assert dt is None or t[k].dtype == dt
            assert torch.equal(t[k], t2[k])
    train = cu.get_coco(str(root), "train", None)
    assert [train.dataset.ids[j] - 100 for j in train.indices] == [i for i in range(n) if bool(g["kept_by_train_filter"][i])]
    assert 0 < len(train) < n
    labeled = train.label_loader([0, 1])
    assert [int(x) for _, (t,) in labeled for x in t["labels"]] == [int(x) for j in train.indices[:2] for x in g["labels_%d" % (train.dataset.ids[j] - 100)]]


def test_coco_bbox_ap_hand_computed_cases(capsys):
    """cald_amd.coco_eval (the COCO side of the evaluation consumer; pycocotools is absent, parity unpinned): cases whose AP / AR
    follow from the published definition by hand -- perfect detections, a (TP, FP, TP-at-low-IoU) ranking, a detection absorbed by a
    crowd region, area ranges, maxDets."""
    from cald_amd.coco_eval import CocoGT, CocoEvaluator, bbox_iou
    cats = [{"id": 3, "name": "c3"}, {"id": 7, "name": "c7"}]
    def run(images, anns, preds):
        ev = CocoEvaluator(CocoGT(images, anns, cats), ["bbox"])
        ev.update(preds); ev.accumulate()
        stats = ev.summarize()
        capsys.readouterr()
        return stats, ev
    img = [{"id": 1, "width": 400, "height": 300}, {"id": 2, "width": 400, "height": 300}]
    gt = [{"id": 1, "image_id": 1, "category_id": 3, "bbox": [10, 10, 100, 100], "area": 10000, "iscrowd": 0},
          {"id": 2, "image_id": 1, "category_id": 3, "bbox": [200, 50, 100, 100], "area": 10000, "iscrowd": 0},
          {"id": 3, "image_id": 2, "category_id": 7, "bbox": [20, 20, 20, 20], "area": 400, "iscrowd": 0}]
    xyxy = lambda b: [b[0], b[1], b[0] + b[2], b[1] + b[3]]
    # 1. perfect detections: every AP is 1 where ground truth exists, -1 where an area range is empty; AR@1 = mean(1/2, 1)
    perfect = {1: {"boxes": np.array([xyxy(gt[0]["bbox"]), xyxy(gt[1]["bbox"])], np.float32), "scores": np.array([0.9, 0.8]), "labels": np.array([3, 3])},
               2: {"boxes": np.array([xyxy(gt[2]["bbox"])], np.float32), "scores": np.array([0.7]), "labels": np.array([7])}}
    s, ev = run(img, gt, perfect)
    np.testing.assert_allclose(s[[0, 1, 2]], 1.0)
    assert abs(s[3] - 1.0) < 1e-12 and s[4] == -1 and abs(s[5] - 1.0) < 1e-12     # (tp / (tp + fp + eps) as in pycocotools) small: the 400-px box; medium: none; large: the two 10 000-px boxes
    np.testing.assert_allclose(s[6], 0.75); np.testing.assert_allclose(s[[7, 8]], 1.0)
    assert ev.coco_eval["bbox"].eval["precision"].shape == (10, 101, 2, 4, 3)
    # 2. one class: scores .9 (IoU 1 with gt 1), .8 (false positive), .7 (IoU 0.62 with gt 2)
    d3 = [200, 50, 100, 62]                                              # inside gt 2: IoU = 6200 / 10000
    assert abs(bbox_iou([d3], [gt[1]["bbox"]], [0])[0, 0] - 0.62) < 1e-12
    preds = {1: {"boxes": np.array([xyxy(gt[0]["bbox"]), [300, 200, 350, 280], xyxy(d3)], np.float32), "scores": np.array([0.9, 0.8, 0.7]), "labels": np.array([3, 3, 3])}}
    s, _ = run(img[:1], gt[:2], preds)
    ap_lo = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101                         # IoU <= 0.60: (TP, FP, TP): precision 1 up to recall 0.5, 2/3 beyond
    ap_hi = 51 * 1.0 / 101                                              # IoU >= 0.65: the third detection is a false positive, recall stops at 0.5
    np.testing.assert_allclose(s[1], ap_lo, rtol=1e-12)
    np.testing.assert_allclose(s[2], ap_hi, rtol=1e-12)
    np.testing.assert_allclose(s[0], (3 * ap_lo + 7 * ap_hi) / 10, rtol=1e-12)
    np.testing.assert_allclose(s[8], (3 * 1.0 + 7 * 0.5) / 10, rtol=1e-12)       # AR@100
    # 3. a detection lying inside a crowd region is ignored (intersection / own area >= 0.5), not counted as a false positive
    gtc = [dict(gt[0]), {"id": 9, "image_id": 1, "category_id": 3, "bbox": [150, 0, 250, 300], "area": 75000, "iscrowd": 1}]
    preds = {1: {"boxes": np.array([[200, 100, 260, 160], xyxy(gt[0]["bbox"])], np.float32), "scores": np.array([0.95, 0.5]), "labels": np.array([3, 3])}}
    s, _ = run(img[:1], gtc, preds)
    np.testing.assert_allclose(s[[0, 1, 2]], 1.0)
    preds[1]["boxes"][0] = [100, 100, 160, 160]                          # moved out of the crowd region: now a higher-scored false positive
    s, _ = run(img[:1], gtc, preds)
    np.testing.assert_allclose(s[0], 0.5, rtol=1e-12)                   # precision 1/2 at every recall level


def test_aspect_ratio_batch_sampler_equals_reference_golden(golden, tmp_path):
    """cald_amd/group_by_aspect_ratio.py (cald_train.py:326-332: the default batch sampler of the training loop) against what the
    imported reference produced (oracle/make_golden_group_sampler.py): group ids for k = 0 / 1 / 3, including ratios exactly on bin
    edges, and every batch of GroupedBatchSampler over sequential and permuted orders, batch sizes 2 / 4 / 7 -- incomplete groups
    topped up in the reference's order.  Plus the dataset fast paths (COCO image table, VOC JPEG headers, Subset)."""
    import torch
    from torch.utils.data.sampler import Sampler
    from cald_amd import group_by_aspect_ratio as G
    z = golden("group_sampler")

    class Sizes(object):
        def __init__(self, hw):
            self.hw = hw

        def __len__(self):
            return len(self.hw)

        def get_height_and_width(self, i):
            return int(self.hw[i][0]), int(self.hw[i][1])

    class Order(Sampler):
        def __init__(self, order):
            self.order = list(order)

        def __iter__(self):
            return iter(self.order)

        def __len__(self):
            return len(self.order)

    n_batches = 0
    for name in ("voc", "coco", "edges", "one_group"):
        hw = z["hw_" + name]
        for k in (0, 1, 3):
            groups = G.create_aspect_ratio_groups(Sizes(hw), k=k)
            assert groups == z["groups_%s_k%d" % (name, k)].tolist(), (name, k)
            for bs in (2, 4, 7):
                for oi in (0, 1):
                    key = "%s_k%d_b%d_o%d" % (name, k, bs, oi)
                    if "order_" + key not in z.files:
                        continue
                    s = G.GroupedBatchSampler(Order(z["order_" + key].tolist()), groups, bs)
                    got = [list(b) for b in s]
                    assert len(got) == len(s) == len(hw) // bs
                    assert got == z["batches_" + key].tolist(), key
                    assert all(len({groups[i] for i in b}) == 1 for b in got), key     # one group per batch
                    n_batches += len(got)
    assert n_batches > 1000
    with pytest.raises(ValueError):
        G.GroupedBatchSampler([0, 1, 2], [0, 0, 0], 2)
    # dataset fast paths: a VOC tree on disk (JPEG headers only), a COCO image table, a Subset of either
    from PIL import Image
    from cald_amd import coco_utils, voc_utils
    base = tmp_path / "VOCdevkit" / "VOC2007"
    for d in ("JPEGImages", "Annotations", "ImageSets/Main"):
        (base / d).mkdir(parents=True)
    sizes = [(375, 500), (500, 375), (333, 500), (500, 500)]
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(np.zeros((h, w, 3), np.uint8)).save(str(base / "JPEGImages" / ("%06d.jpg" % i)))
        (base / "Annotations" / ("%06d.xml" % i)).write_text("<annotation><filename>%06d.jpg</filename></annotation>" % i)
    (base / "ImageSets" / "Main" / "trainval.txt").write_text("".join("%06d\n" % i for i in range(len(sizes))))
    ds = voc_utils.VOCDetection(str(tmp_path), "2007", "trainval", None)
    want = [float(w) / float(h) for h, w in sizes]
    assert G.compute_aspect_ratios(ds) == want
    assert G.compute_aspect_ratios(coco_utils.Subset(ds, [2, 0])) == [want[2], want[0]]
    assert G.create_aspect_ratio_groups(ds, k=3) == G._quantize(want, (2 ** np.linspace(-1, 1, 7)).tolist())
    import json
    ann = tmp_path / "instances.json"
    ann.write_text(json.dumps({"images": [{"id": 7, "width": 640, "height": 480, "file_name": "a.jpg"}, {"id": 3, "width": 427, "height": 640, "file_name": "b.jpg"}],
                               "annotations": [], "categories": []}))
    cds = coco_utils.CocoDetection(str(tmp_path), str(ann), None)
    assert G.compute_aspect_ratios(cds) == [427.0 / 640.0, 640.0 / 480.0]          # ids sorted: 3, 7


def _golden_loss_case(g, k, dtype):
    import torch
    N = int(g["l%d_N" % k])
    gts = [torch.from_numpy(g["l%d_gt%d" % (k, i)]).to(dtype) for i in range(N)]
    labels = [torch.from_numpy(g["l%d_labels%d" % (k, i)]) for i in range(N)]
    return (N, torch.from_numpy(g["l%d_cls_logits" % k]).to(dtype), torch.from_numpy(g["l%d_bbox_regression" % k]).to(dtype),
            torch.from_numpy(g["l%d_anchors" % k]).to(dtype), gts, labels)


def test_training_checker_retinanet_losses_match_the_reference_code(golden):
    """SURVEY 8f rank 4 pin: the float64 training checker's RetinaNet losses (oracle/torch_train.retina_losses) against the
    REFERENCE's own compute_loss bodies (detection/retinanet_cal.py:100-133, :185-223, :389-400, executed by
    oracle/make_golden_train_losses.py): matched indices identical, both losses to float64 round-off; and to float32
    round-off of the reference's own float32 run."""
    import torch
    from oracle import torch_train as tt
    g = golden("train_losses")
    for k in range(int(g["l_n"])):
        N, cls, reg, anchors, gts, labels = _golden_loss_case(g, k, torch.float64)
        # matcher decisions are taken on float32 IoUs, as the reference's float32 run takes them
        lc, lr, matched = tt.retina_losses(cls, reg, anchors.float(), [b.float() for b in gts], labels)
        assert np.array_equal(torch.stack(matched).numpy(), g["l%d_matched" % k]), k
        assert abs(float(lc) - float(g["l%d_cls_f64" % k])) <= 1e-12 * abs(float(g["l%d_cls_f64" % k])), (k, float(lc))
        # the checker encodes targets from the float32 boxes in float64; the reference's f64 run did the same on the same values
        assert abs(float(lr) - float(g["l%d_reg_f64" % k])) <= 1e-12 * abs(float(g["l%d_reg_f64" % k])), (k, float(lr))
        assert abs(float(lc) - float(g["l%d_cls_f32" % k])) <= 2e-6 * abs(float(lc))
        assert abs(float(lr) - float(g["l%d_reg_f32" % k])) <= 2e-6 * abs(float(lr))


def test_training_loop_follows_the_reference_loop(golden):
    """cald_amd.engine.train_one_epoch / warmup_lr_scheduler against cald_train.py:40-74 + detection/utils.py:239-247 run on
    the same stand-in model (oracle/make_golden_train_losses.py): the learning rate every update is taken with and the
    parameter trajectory are identical (float64, bit for bit)."""
    import torch
    from cald_amd import engine
    g = golden("train_losses")

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor([0.5, -1.25, 2.0], dtype=torch.float64))
            self.b = torch.nn.Parameter(torch.tensor([0.1], dtype=torch.float64))

        def forward(self, images, targets):
            x = torch.stack([im.double().mean() for im in images])
            t = torch.stack([tg["boxes"].double().sum() for tg in targets])
            pred = x[:, None] * self.w[None, :] + self.b
            return {"loss_a": ((pred.sum(1) - t) ** 2).mean() * 0.1, "loss_b": (self.w ** 2).sum() * 0.01 + self.b.abs().sum()}

    for k in range(int(g["t_n"])):
        n_iter, epochs = int(g["t%d_iters" % k]), int(g["t%d_epochs" % k])
        data = [([torch.from_numpy(im) for im in g["t%d_im%d" % (k, i)]], [{"boxes": torch.from_numpy(b)} for b in g["t%d_bx%d" % (k, i)]])
                for i in range(n_iter)]
        model = Toy()
        lrs, params = [], []

        class SpySGD(torch.optim.SGD):
            def step(self, closure=None):
                lrs.append(self.param_groups[0]["lr"])
                r = super().step(closure)
                params.append(np.concatenate([model.w.detach().numpy().copy(), model.b.detach().numpy().copy()]))
                return r
        opt = SpySGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3)
        for ep in range(epochs):
            engine.train_one_epoch(model, opt, data, torch.device("cpu"), 0, ep, 0)
        assert np.array_equal(np.array(lrs), g["t%d_lrs" % k]), (k, lrs)
        assert np.array_equal(np.stack(params), g["t%d_params" % k]), k


def test_training_checker_frcnn_losses_match_the_reference_repo_copies(golden):
    """The float64 training checker's loss arithmetic (oracle/torch_train.py: smooth_l1_sum, cross entropy, BCE with logits) against the
    reference tree's own copies of torchvision's Faster R-CNN losses, executed from it (detection/frcnn_ll.py:28-63, :245-281 ->
    tests/golden/frcnn_losses.npz): 1e-12 in float64.  The copies use beta = 1 / L1 where torchvision (the hot path) uses 1/9."""
    import torch
    import torch.nn.functional as F
    from oracle import torch_train as tt
    g = golden("frcnn_losses")
    for k in range(int(g["b_n"])):
        logits, deltas = torch.from_numpy(g["b%d_logits" % k]).double(), torch.from_numpy(g["b%d_deltas" % k]).double()
        labels, tgt = torch.from_numpy(g["b%d_labels" % k]), torch.from_numpy(g["b%d_targets" % k]).double()
        R, Cc = logits.shape
        pos = torch.nonzero(labels > 0).squeeze(1)
        cls = F.cross_entropy(logits, labels)
        box = tt.smooth_l1_sum(deltas.view(R, Cc, 4)[pos, labels[pos]], tgt[pos], 1.0) / R
        assert abs(float(cls) - float(g["b%d_cls_f64" % k])) <= 1e-12 and abs(float(box) - float(g["b%d_box_f64" % k])) <= 1e-12
    for k in range(int(g["r_n"])):
        obj, deltas, tgt = [torch.from_numpy(g["r%d_%s" % (k, n)]).double() for n in ("obj", "deltas", "targets")]
        pos, neg = torch.from_numpy(g["r%d_pos" % k]), torch.from_numpy(g["r%d_neg" % k])
        samp = torch.cat([pos, neg])
        lab = torch.cat([torch.ones(len(pos)), torch.zeros(len(neg))]).double()
        o = F.binary_cross_entropy_with_logits(obj.flatten()[samp], lab)
        b = tt.smooth_l1_sum(deltas[pos], tgt[pos], 0.0) / len(samp)
        assert abs(float(o) - float(g["r%d_obj_f64" % k])) <= 1e-12 and abs(float(b) - float(g["r%d_box_f64" % k])) <= 1e-12


def test_coco_bbox_ap_known_answers_from_the_designed_scene(golden, capsys):
    """cald_amd.coco_eval against tests/golden/coco_ap_known_answers.npz: a scene in which every detection's fate at every IoU threshold
    follows from its construction (a ground-truth box shrunk to a chosen IoU, a box that overlaps nothing, a box inside a crowd region),
    turned into the twelve COCO statistics by oracle/make_known_answers_coco.py from the published definitions only -- that script runs
    no matcher and none of this package's code.  Two categories, three images, all three area ranges, a crowd, maxDets 1 / 10 / 100."""
    import json
    from cald_amd.coco_eval import CocoGT, COCOeval
    g = golden("coco_ap_known_answers")
    images, cats = json.loads(str(g["images"])), json.loads(str(g["categories"]))
    anns, dets = json.loads(str(g["annotations"])), json.loads(str(g["detections"]))
    ev = COCOeval(CocoGT(images, anns, cats))
    ev.add_results(dets)
    ev.evaluate([im["id"] for im in images])
    ev.accumulate()
    got = ev.summarize(quiet=True)
    np.testing.assert_allclose(got, g["stats"], rtol=0, atol=1e-12)
    assert (g["stats"] > 0).all() and len(set(np.round(g["stats"], 6))) >= 10          # the scene separates the twelve numbers


def test_transform_size_matches_torch_interpolate_output_shape(oracle):
    """GeneralizedRCNNTransform.resize (torchvision 0.8.2, called at frcnn_la.py:234) sizes its output with
    F.interpolate(scale_factor=s, recompute_scale_factor=True): floor(in * s) in torch's own arithmetic.  That is an INDEPENDENT
    statement of the resized shape (neither the oracle nor the library): both must agree with it on a sweep of image sizes that
    includes the float-product traps (333 * (600 / 333), 1 x N strips, sizes near the max-size switch)."""
    import ctypes as C
    import torch
    import torch.nn.functional as F
    from cald_amd import _ffi
    L = _ffi.lib()
    rng = np.random.RandomState(5)
    sizes = [(375, 500), (500, 375), (333, 500), (500, 334), (480, 640), (427, 640), (640, 427), (333, 333), (600, 1000), (1000, 600),
             (601, 1001), (599, 999), (800, 1333), (1333, 800), (37, 2000), (2000, 37), (17, 19), (1, 50), (50, 1), (999, 1665)]
    sizes += [(int(h), int(w)) for h, w in zip(rng.randint(20, 1400, 400), rng.randint(20, 1400, 400))]
    n = 0
    for (H, W) in sizes:
        for (mn, mx) in [(600, 1000), (800, 1333), (512, 512), (300, 500)]:
            s = float(mn) / float(min(H, W))
            if max(H, W) * s > mx:
                s = float(mx) / float(max(H, W))
            if int(np.floor(H * s)) < 1 or int(np.floor(W * s)) < 1:
                continue
            want = tuple(F.interpolate(torch.zeros(1, 1, H, W), scale_factor=s, mode="bilinear", recompute_scale_factor=True,
                                       align_corners=False).shape[-2:])
            want = want + tuple(-(-d // 32) * 32 for d in want)
            assert oracle.transform_size(H, W, mn, mx) == want, (H, W, mn, mx)
            v = [C.c_int() for _ in range(4)]
            _ffi.check(L.cald_op_transform_size(H, W, mn, mx, *[C.byref(x) for x in v]))
            assert tuple(x.value for x in v) == want, (H, W, mn, mx)
            n += 1
    assert n > 1500


def test_roi_sample_host_labels_balanced_sampling_and_index_lists():
    """cald_train_roi_sample_host (the training forward's one host stop) against a plain numpy restatement of
    roi_heads.select_training_samples: labels from the Matcher values (>= 0 -> the ground truth's label, -1 -> background, -2 ->
    ignored), unused proposal slots skipped, min(batch * fraction, #pos) positives + the rest negatives = the smallest keys of each
    class, rows in (image, table row) order, the loss kernels' index lists, images without ground truth all background."""
    from cald_amd import train_ops
    rs = np.random.RandomState(5)
    for case in range(6):
        N = 1 + case % 4
        slots = [int(rs.randint(20, 400)) for _ in range(N)]
        n_gt = [int(rs.randint(0, 4)) if case else 2 for _ in range(N)]
        counts = [int(rs.randint(1, s + 1)) for s in slots] if case % 2 else None
        batch, frac, pred_ld, ncls = [(64, 0.25), (512, 0.25), (16, 0.5)][case % 3] + (108, 21)
        rows = [s + g for s, g in zip(slots, n_gt)]
        T = sum(rows)
        matched = np.concatenate([np.where(rs.rand(r) < 0.2, rs.randint(0, max(g, 1), r), np.where(rs.rand(r) < 0.1, -2, -1)).astype(np.int32)
                                  if g else np.full(r, -1, np.int32) for r, g in zip(rows, n_gt)] + [np.zeros(N, np.int32)])
        gt_labels = np.concatenate([rs.randint(1, ncls, g).astype(np.int64) for g in n_gt] + [np.zeros(1, np.int64)])
        keys = rs.rand(T)
        blk, cap, R, n_pos, per = train_ops.roi_sample_host(slots, n_gt, counts, matched, gt_labels, keys, batch, frac, pred_ld, ncls)
        keep, gsel, lab, pidx, prow = blk[:R], blk[cap:cap + R], blk[2 * cap:2 * cap + R], blk[3 * cap:3 * cap + n_pos], blk[4 * cap:4 * cap + n_pos]
        img = blk[5 * cap:].view(np.float32)[:R]
        assert cap == N * batch and sum(per) == R
        o, r0, g0, want_keep, want_lab, want_g, want_img = 0, 0, 0, [], [], [], []
        for i in range(N):
            used = counts[i] if counts is not None else slots[i]
            cand = np.concatenate([np.arange(used), slots[i] + np.arange(n_gt[i])])
            m = matched[r0 + cand]
            l = np.where(m >= 0, gt_labels[g0 + np.maximum(m, 0)], np.where(m == -1, 0, -1)) if n_gt[i] else np.zeros(len(cand), np.int64)
            pos, neg = cand[l >= 1], cand[l == 0]
            k_pos = min(int(batch * frac), len(pos)); k_neg = min(batch - k_pos, len(neg))
            sp = pos[np.argsort(keys[r0 + pos], kind="stable")[:k_pos]]; sn = neg[np.argsort(keys[r0 + neg], kind="stable")[:k_neg]]
            kp = np.sort(np.concatenate([sp, sn]))
            assert per[i] == len(kp)
            lmap = dict(zip(cand.tolist(), l.tolist())); mmap = dict(zip(cand.tolist(), m.tolist()))
            want_keep += [r0 + c for c in kp]; want_lab += [lmap[c] for c in kp]; want_img += [float(i)] * len(kp)
            want_g += [(g0 + max(mmap[c], 0)) if n_gt[i] else sum(n_gt) for c in kp]
            r0 += rows[i]; g0 += n_gt[i]
        assert keep.tolist() == want_keep and lab.tolist() == want_lab and gsel.tolist() == want_g and img.tolist() == want_img
        assert prow.tolist() == [r for r in range(R) if lab[r] > 0]
        assert pidx.tolist() == [r * pred_ld + ncls + 4 * int(lab[r]) for r in range(R) if lab[r] > 0]
    with pytest.raises(RuntimeError):
        train_ops.roi_sample_host([10], [1], [11], np.zeros(12, np.int32), np.ones(1, np.int64), np.zeros(11), 8, 0.25, 108, 21)
