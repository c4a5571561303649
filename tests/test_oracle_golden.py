"""The CPU oracle against golden vectors captured from the imported reference
(oracle/make_golden.py; reference cald_train.py:91-271, cald/cald_helper.py:23-243)."""
import numpy as np
import pytest

SCORING = ["scoring_frcnn_F", "scoring_frcnn_FCD", "scoring_retina_FCD", "scoring_frcnn_coco_FD", "scoring_frcnn_FSCDR",
           "scoring_frcnn_ALL"]      # ALL = every augmentation branch of get_uncertainty that runs in the reference (28 views)


def _dets(g, i, v):
    return {k: g["det%d_%d_%s" % (i, v, k)] for k in ("boxes", "labels", "scores", "prob_max", "scores_cls")}


def materialize(orc, img, flip, rects):
    out = img[:, ::-1].copy() if flip else img.copy()
    if rects is not None:
        for (l, t, r, b) in rects:
            out[t:b, l:r] = 0
    return out


@pytest.mark.parametrize("name", SCORING)
def test_scoring_matches_reference(oracle, golden, name):
    g = golden(name)
    augs = [str(a) for a in g["augs"]]
    C, bp, base_seed = int(g["C"]), float(g["bp"]), int(g["base_seed"])
    for i in range(int(g["n_images"])):
        img = g["img%d" % i]
        ref = oracle.subsample_ref(_dets(g, i, 0))
        nviews = int(g["per_image"][i])
        if nviews == 1:
            c, cc = oracle.score_image(ref, [], [], C, bp)
        else:
            views = oracle.build_views(img, augs, ref, oracle.image_seed(base_seed, i))
            assert len(views) == nviews - 1
            outs = [_dets(g, i, v) for v in range(1, nviews)]
            c, cc = oracle.score_image(ref, outs, [v[3] for v in views], C, bp)
            for vi, v in enumerate(views):   # pixel-exact augmented images as the reference's detector saw them
                key = "seen%d_%d" % (i, vi + 1)
                if len(v) > 4:               # GaussianNoise view: float image, torch's libm vs the oracle's polynomials
                    if "seenf%d_%d" % (i, vi + 1) in g.files:
                        got = (v[0].astype(np.float32) / np.float32(255.0)) + v[4].transpose(1, 2, 0)
                        np.testing.assert_allclose(got, g["seenf%d_%d" % (i, vi + 1)], rtol=0, atol=2e-6)
                elif key in g.files:
                    np.testing.assert_array_equal(materialize(oracle, v[0], v[1], v[2]), g[key])
        assert abs(c - g["consistency"][i]) <= 1e-5, (name, i, c, g["consistency"][i])
        np.testing.assert_allclose(cc, g["cls_all"][i], rtol=0, atol=1e-7)


def test_js_matches_scipy(oracle, golden):
    g = golden("js")
    for p, q, (C, js) in zip(g["p"], g["q"], g["cj"]):
        C = int(C)
        got = oracle.js_divergence(p[:C], q[:C])
        assert abs(got - js) <= 2e-6, (C, got, js)


def test_helpers_match_reference(oracle, golden):
    g = golden("helpers")
    for i in range(4):
        img, boxes = g["img%d" % i], g["boxes%d" % i]
        H, W, _ = img.shape
        np.testing.assert_array_equal(oracle.flip_boxes(boxes, W), g["flip_boxes%d" % i])
        np.testing.assert_array_equal(img[:, ::-1], g["flip_img%d" % i])
        for r10, r in ((8, 0.8), (12, 1.2), (7, 0.7)):
            np.testing.assert_array_equal(oracle.resize_aug(img, r), g["resize%d_%d_img" % (i, r10)])
            np.testing.assert_array_equal((boxes * np.float32(r)).astype(np.float32), g["resize%d_%d_boxes" % (i, r10)])
        for s in (11, 12, 13):
            rects = oracle.cutout_rects(s, H, W, boxes, 2)
            np.testing.assert_array_equal(materialize(oracle, img, False, rects), g["cutout%d_%d_img" % (i, s)])
        ri, rb = oracle.rotate_aug(img, boxes, 5)
        np.testing.assert_array_equal(ri, g["rotate%d_img" % i])                      # PIL rotate(expand) + default resize
        np.testing.assert_allclose(rb, g["rotate%d_boxes" % i], rtol=0, atol=1e-4)    # float tolerance of BASELINE.json
        for s in (21, 22):
            np.testing.assert_array_equal(oracle.salt_pepper(img, 0.1, s), g["sp%d_%d_img" % (i, s)])
        if i >= 2:   # GaussianNoise: torch.randn's stream structure, float tolerance 1e-4 (measured ~1e-7)
            noise = oracle.gaussian_noise(31 + i, H, W, 16)
            got = img.astype(np.float32) / np.float32(255) + noise.transpose(1, 2, 0)
            np.testing.assert_allclose(got, g["ga%d_img" % i], rtol=0, atol=1e-6)
    for s in (0, 1, 12345, (1 << 40) + 17):
        np.testing.assert_array_equal(oracle.py_random(s, 8), g["pyrandom_%d" % s])


def test_pil_resize_against_installed_pillow(oracle):
    from PIL import Image
    rs = np.random.RandomState(0)
    for (H, W) in [(375, 500), (333, 500), (500, 375), (61, 47)]:
        img = (rs.rand(H, W, 3) * 255).astype(np.uint8)
        for r in (0.8, 1.2, 0.5):
            ow, oh = int(W * r), int(H * r)
            want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
            np.testing.assert_array_equal(oracle.pil_resize_bilinear(img, oh, ow), want)


def test_subsample_indices(oracle):
    for n in (1, 40, 41, 49, 50, 51, 100, 300, 6300):
        want = np.arange(n) if n <= 40 else np.round(np.linspace(0, n - 1, 50)).astype(int)
        np.testing.assert_array_equal(oracle.subsample_indices(n), want)


def test_det_math_accuracy(oracle):
    x = np.linspace(-80, 80, 20001).astype(np.float32)
    e = oracle.exp_array(x)
    np.testing.assert_allclose(e, np.exp(x.astype(np.float64)), rtol=3e-7)
    y = np.concatenate([np.logspace(-30, 30, 20001), np.linspace(0.5, 2, 5001)]).astype(np.float32)
    l = oracle.log_array(y)
    np.testing.assert_allclose(l, np.log(y.astype(np.float64)), rtol=3e-7, atol=2e-7)


def test_baseline_sweeps_match_reference(oracle, golden):
    """lt_c_train.py:105-121 and ls_c_train.py:108-155 scoring against the imported reference."""
    g = golden("baselines")
    for i in range(4):
        out = {k: g["lt%d_%s" % (i, k)] for k in ("boxes", "props", "prob_max")}
        assert abs(oracle.lt_uncertainty(out) - g["lt_unc"][i]) <= 1e-6
    for i in range(int(g["ls_n"])):
        per = int(g["ls_per"][i])
        ref = {k: g["ls%d_0_%s" % (i, k)] for k in ("boxes", "prob_max")}
        outs = [{"boxes": g["ls%d_%d_boxes" % (i, v)]} for v in range(1, per)]
        assert abs(oracle.ls_score_image(ref, outs) - g["ls_unc"][i]) <= 1e-6, i
    img = g["ls_img0"]
    H, W, _ = img.shape
    noise = oracle.gaussian_noise_seq(oracle.image_seed(5, 0), H, W, [8.0 * k for k in range(1, 7)])
    for v in range(1, 7):    # the six sequential GaussianNoise views exactly as the reference's detector saw them
        got = img.astype(np.float32) / np.float32(255) + noise[v - 1].transpose(1, 2, 0)
        np.testing.assert_allclose(got, g["ls_seen0_%d" % v], rtol=0, atol=2e-6)


def test_postprocess_bookkeeping_matches_reference_bodies(oracle, golden):
    """A19 / A22: the oracle's postprocess against the outputs of the reference's OWN method bodies
    (frcnn_la.py:32-87, retinanet_cal.py:402-490 run under the stub harness, oracle/make_golden_postprocess.py).
    Index bookkeeping must agree exactly (count, labels, order); floats within 1e-4 (torch's exp / softmax / sigmoid vs
    the arithmetic contract's polynomials)."""
    g = golden("postprocess")
    for k in range(int(g["f_n"])):
        H, W = [int(v) for v in g["f%d_hw" % k]]
        got = oracle.frcnn_postprocess(g["f%d_logits" % k], g["f%d_deltas" % k], g["f%d_props" % k], H, W, H, W)
        want = {n: g["f%d_out_%s" % (k, n)] for n in ("boxes", "scores", "labels", "props", "prob_max", "scores_cls")}
        assert got["labels"].shape == want["labels"].shape, (k, got["labels"].shape, want["labels"].shape)
        np.testing.assert_array_equal(got["labels"], want["labels"])
        np.testing.assert_array_equal(got["props"], want["props"])
        for n in ("scores", "prob_max", "scores_cls"):
            np.testing.assert_allclose(got[n], want[n].reshape(got[n].shape), rtol=0, atol=1e-5)
        np.testing.assert_allclose(got["boxes"], want["boxes"].reshape(-1, 4), rtol=0, atol=1e-3)
    base = g["r_base"]
    for k in range(int(g["r_n"])):
        K = int(g["r%d_K" % k]); Hp, Wp, Hr, Wr = [int(v) for v in g["r%d_sizes" % k]]
        cls = [g["r%d_cls%d" % (k, l)] for l in range(5)]; reg = [g["r%d_reg%d" % (k, l)] for l in range(5)]
        got = oracle.retina_postprocess(cls, reg, base, Hp, Wp, Hr, Wr, Hr, Wr, K)
        want = {n: g["r%d_out_%s" % (k, n)] for n in ("boxes", "scores", "labels", "scores_cls", "prob_max")}
        assert got["labels"].shape == want["labels"].shape, (k, got["labels"].shape, want["labels"].shape)
        np.testing.assert_array_equal(got["labels"], want["labels"])
        for n in ("scores", "prob_max", "scores_cls"):
            np.testing.assert_allclose(got[n], want[n].reshape(got[n].shape), rtol=0, atol=1e-5)
        np.testing.assert_allclose(got["boxes"], want["boxes"], rtol=0, atol=1e-3)


# ---------------------------------------------------------------------------------------------------------------------------
# Detector code that lives in the reference repo itself, executed under the stub harness (oracle/make_golden_rpn.py): the RPN's
# concat / filter / forward (detection/frcnn_ll.py:207-238, :284-321, :323-374), RetinaNet's heads and default anchor sizes
# (detection/retinanet_cal.py:57-62, :135-151, :225-241, :346-351), resize_boxes (detection/frcnn_la.py:292-315).
# ---------------------------------------------------------------------------------------------------------------------------
def _rpn_cases(g):
    for name in [str(n) for n in g["names"]]:
        Hp, Wp, Hr, Wr, pre, post = [int(v) for v in g[name + "_cfg"]]
        yield name, [g["%s_head%d" % (name, l)] for l in range(5)], (Hp, Wp, Hr, Wr, pre, post)


def test_rpn_layout_and_filter_match_the_reference_code(oracle, golden):
    g = golden("rpn_filter")
    base = g["base"].reshape(5, 3, 4)
    seen = 0
    for name, heads, (Hp, Wp, Hr, Wr, pre, post) in _rpn_cases(g):
        # concat_box_prediction_layers: row (level, y, x, a) of the flattened tensors is channel a / 3 + 4 a + j of the NHWC head the
        # oracle (and rpn.hip) read -- the (N, A * C, H, W) -> (N * HWA, C) permutation, by the reference's own function
        flat_l = np.concatenate([h[:, :, :3].reshape(-1) for h in heads]); flat_d = np.concatenate([h[:, :, 3:].reshape(-1, 4) for h in heads])
        np.testing.assert_array_equal(g[name + "_flat_logits"].reshape(-1), flat_l)
        np.testing.assert_array_equal(g[name + "_flat_deltas"], flat_d)
        boxes, scores = oracle.rpn_proposals(heads, base, Hp, Wp, Hr, Wr, A=3, pre_n=pre, post_n=post, nms_thr=0.7, min_size=1e-3)
        want_b, want_s = g[name + "_boxes"], g[name + "_scores"]
        if name == "pad":
            # frcnn_ll.py:314-316 (the learning-loss baseline's copy) returns post_nms_top_n all-zero boxes when fewer survive; stock
            # torchvision -- what the hot path's frcnn_la.py instantiates -- returns the survivors.  The oracle follows the hot path.
            assert want_b.shape == (post, 4) and not want_b.any() and 0 < len(boxes) < post
            continue
        assert len(boxes) == len(want_b) == min(post, len(want_b)), name
        np.testing.assert_array_equal(scores, want_s, err_msg=name)            # raw logits gathered in selection order: same anchors, same order
        np.testing.assert_allclose(boxes, want_b, rtol=0, atol=1e-4, err_msg=name)   # exp(): fixed polynomial here, libm there
        seen += 1
    assert seen >= 6


def test_retinanet_heads_match_the_reference_modules(oracle, golden):
    g = golden("retina_heads")
    kmaj = lambda w: np.ascontiguousarray(w.transpose(2, 3, 1, 0).reshape(-1, w.shape[0]))
    for k in range(int(g["n"])):
        cin, K, L = [int(v) for v in g["h%d_cfg" % k]]
        W = lambda n: g["h%d_w_%s" % (k, n)]
        cls_rows, reg_rows = [[], []], [[], []]
        for l in range(L):
            feats = g["h%d_feat%d" % (k, l)]                                   # [N][C][H][W]
            for n in range(feats.shape[0]):
                x = np.ascontiguousarray(feats[n].transpose(1, 2, 0))
                t = x
                for i in range(4):
                    t = oracle.conv2d(t, kmaj(W("classification_head.conv.%d.weight" % (2 * i))), 3, 3, 1, 1, bias=W("classification_head.conv.%d.bias" % (2 * i)), relu=True)
                c = oracle.conv2d(t, kmaj(W("classification_head.cls_logits.weight")), 3, 3, 1, 1, bias=W("classification_head.cls_logits.bias"))
                t = x
                for i in range(4):
                    t = oracle.conv2d(t, kmaj(W("regression_head.conv.%d.weight" % (2 * i))), 3, 3, 1, 1, bias=W("regression_head.conv.%d.bias" % (2 * i)), relu=True)
                r = oracle.conv2d(t, kmaj(W("regression_head.bbox_reg.weight")), 3, 3, 1, 1, bias=W("regression_head.bbox_reg.bias"))
                cls_rows[n].append(c.reshape(-1, K)); reg_rows[n].append(r.reshape(-1, 4))      # NHWC channel a * K + k  ->  row (y, x, a), column k
        for n in range(2):
            got_c, got_r = np.concatenate(cls_rows[n]), np.concatenate(reg_rows[n])
            want_c, want_r = g["h%d_cls_logits" % k][n], g["h%d_bbox_regression" % k][n]
            assert got_c.shape == want_c.shape and got_r.shape == want_r.shape
            np.testing.assert_allclose(got_c, want_c, rtol=0, atol=1e-5 * float(np.abs(want_c).max()))
            np.testing.assert_allclose(got_r, want_r, rtol=0, atol=1e-5 * float(np.abs(want_r).max()))
    # RetinaNet.__init__ (retinanet_cal.py:346-355): anchor sizes, nine anchors per location, K * 9 classification outputs
    assert [tuple(int(v) for v in row) for row in g["anchor_sizes"]] == [tuple(s) for s in oracle.retina_anchor_sizes()]
    assert g["aspect_ratios"].tolist() == [[0.5, 1.0, 2.0]] * 5 and g["anchors_per_location"].tolist() == [9] * 5
    assert int(g["init_cls_out_channels"]) == 9 * 4 and g["init_thresholds"].tolist() == [0.05, 0.5, 300.0]


def test_resize_boxes_matches_the_reference_function(oracle, golden):
    g = golden("resize_boxes")
    for k in range(int(g["n"])):
        Hr, Wr, Ho, Wo = [int(v) for v in g["r%d_sizes" % k]]
        assert (Hr, Wr) != (Ho, Wo)
        np.testing.assert_array_equal(oracle.resize_boxes(g["r%d_boxes" % k], Hr, Wr, Ho, Wo), g["r%d_out_boxes" % k])
        np.testing.assert_array_equal(oracle.resize_boxes(g["r%d_props" % k], Hr, Wr, Ho, Wo), g["r%d_out_props" % k])


# ---------------------------------------------------------------------------------------------------------------------------
# torchvision 0.8.2 primitives whose source is not in /root/reference: known answers derived in float64 by
# oracle/make_known_answers.py (which calls none of the oracle's code) -- the oracle is checked here, the HIP kernels in
# tests/test_gpu_parity.py::test_tv_known_answers_*.
# ---------------------------------------------------------------------------------------------------------------------------
def _ramp_feats(level_hw, C):
    c = np.arange(C)
    a, b, g = (c % 7 - 3) / 8.0, (c % 5 - 2) / 4.0, c / 16.0
    feats = []
    for l, (H, W) in enumerate(level_hw):
        yy, xx = np.mgrid[0:H, 0:W]
        feats.append((a[None, None, :] * xx[:, :, None] + b[None, None, :] * yy[:, :, None] + g[None, None, :] + 10.0 * l).astype(np.float32))
    return feats


@pytest.mark.parametrize("C", [8, 256])
def test_tv_known_answers_roi_align_on_affine_ramps(oracle, golden, C):
    """MultiScaleRoIAlign (frcnn_la.py:205-209) on affine feature ramps: bilinear sampling is exact there, so every bin is the ramp at
    the bin's (clamped) sample points -- boxes inside, hanging off each edge, narrower than a pixel, empty, in the far corner."""
    g = golden("tv_known_answers")
    level_hw = [tuple(int(v) for v in hw) for hw in g["roi_level_hw"]]
    got = oracle.roi_align(_ramp_feats(level_hw, C), g["roi_rois"])
    np.testing.assert_allclose(got, g["roi_expected_c%d" % C], rtol=0, atol=2e-3)


def test_tv_known_answers_level_mapper_edges(oracle, golden):
    """LevelMapper at 112 * 2^j +- float32 steps and at both clamps (the + 1e-6 keeps a side one step below a boundary on the upper level)."""
    import ctypes as C
    g = golden("tv_known_answers")
    f = oracle.lib().orc_roi_level
    f.restype = C.c_int
    got = [f(np.ascontiguousarray(b, np.float32).ctypes.data_as(C.POINTER(C.c_float))) for b in g["level_rois"]]
    assert got == [int(v) for v in g["level_expected"]]


@pytest.mark.parametrize("case", ["nms_a", "nms_b", "decode"])
def test_tv_known_answers_nms_and_box_decode(oracle, golden, case):
    """nms / batched_nms / BoxCoder.decode through postprocess_detections (frcnn_la.py:32-87): IoU exactly at the threshold is kept and
    a hair above is dropped, equal scores keep the lower index, classes do not suppress each other, a suppressed box suppresses
    nothing; coordinates near 2e4 with label 90 (the float32 class offset decides); dw clamped at log(1000 / 16)."""
    g = golden("tv_known_answers")
    Hr, Wr = [int(v) for v in g[case + "_hw"]]
    got = oracle.frcnn_postprocess(g[case + "_logits"], g[case + "_deltas"], g[case + "_props"], Hr, Wr, Hr, Wr)
    assert len(got["boxes"]) == len(g[case + "_boxes"])
    np.testing.assert_array_equal(got["labels"], g[case + "_labels"])
    np.testing.assert_allclose(got["boxes"], g[case + "_boxes"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(got["scores"], g[case + "_scores"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got["props"], g[case + "_props"][g[case + "_src"]], rtol=0, atol=0)


def test_tv_known_answers_base_anchor_table(oracle, golden):
    """AnchorGenerator base anchors for sizes 32..512 x ratios (0.5, 1, 2): the published table ([-23, -11, 23, 11], ...)."""
    g = golden("tv_known_answers")
    got = np.stack([oracle.base_anchors([s], [0.5, 1.0, 2.0]) for s in (32, 64, 128, 256, 512)])
    np.testing.assert_array_equal(got, g["base_anchors"].astype(np.float32))


def test_mfma_f16_model_matches_the_hardware_sample_and_its_fast_forms(oracle):
    """oracle/mfma_f16_model.h (the CPU statement of v_mfma_f32_32x32x16_f16, the primitive of CALD_PRECISION_F16X3) against results the
    instruction itself produced on an MI355X (tests/golden/mfma_f16_hw.npz <- tools/mfma_model/make_golden.py: 1 000 cases of each
    directed family), and the two double-precision evaluations oracle/f16x3_oracle.c convolves with against the integer statement on
    two million random operands.  The `-m gpu` suite repeats the first half live on more than 10^7 cases."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "mfma_f16_hw.npz"))
    assert g["D"].size > 20000 and len(set(g["family"].tolist())) >= 20
    for fast in (None, 0, 1):
        got = oracle.mfma_f16_dot16(g["A"], g["B"], g["C"], fast)
        if got is None: continue
        bad = np.nonzero(got != g["D"])[0]
        assert bad.size == 0, (fast, bad.size, g["family"][bad[:5]])
    rs = np.random.RandomState(3)
    n = 500000
    for spread in (0, 3, 8, 14):
        ea = np.clip(rs.randint(-spread, spread + 1, (n, 16)) + rs.randint(-3, 4, (n, 1)), -15, 15)
        eb = np.clip(rs.randint(-spread, spread + 1, (n, 16)) + rs.randint(-3, 4, (n, 1)), -15, 15)
        A = ((rs.randint(0, 2, (n, 16)) << 15) | ((ea + 15) << 10) | rs.randint(0, 1024, (n, 16))).astype(np.uint16)
        B = ((rs.randint(0, 2, (n, 16)) << 15) | ((eb + 15) << 10) | rs.randint(0, 1024, (n, 16))).astype(np.uint16)
        A[rs.rand(n, 16) < 0.15] = 0
        Cc = ((rs.randint(0, 2, n).astype(np.uint32) << 31) | ((rs.randint(-40, 41, n) + 127).astype(np.uint32) << 23) | rs.randint(0, 1 << 23, n).astype(np.uint32))
        Cc[rs.rand(n) < 0.05] = 0
        want = oracle.mfma_f16_dot16(A, B, Cc)
        for fast in (0, 1):
            got = oracle.mfma_f16_dot16(A, B, Cc, fast)
            if got is None: continue
            assert np.array_equal(got, want), (spread, fast, int((got != want).sum()))


def test_conv2d_f16x3_oracle_is_the_exact_conv_to_split_precision(oracle):
    """oracle.conv2d_f16x3 (the CPU restatement of CALD_PRECISION_F16X3) stays within 2^-20 sum |a||w| of the exact chain, is the same in its
    portable and AVX-512 forms, and does not depend on the thread count."""
    rs = np.random.RandomState(1)
    for (H, W, Cin, Cout, K, stride, pad) in [(13, 17, 4, 64, 7, 2, 3), (9, 11, 64, 64, 3, 1, 1), (7, 9, 32, 128, 1, 2, 0)]:
        x = rs.randn(H, W, Cin).astype(np.float32); x[rs.rand(H, W, Cin) < 0.3] = 0
        wk = (rs.randn(K * K * Cin, Cout) * np.sqrt(2.0 / (Cin * K * K))).astype(np.float32)
        b = rs.randn(Cout).astype(np.float32)
        assert oracle.uses_f16x3(Cin, Cout, K, K)
        y = oracle.conv2d_f16x3(x, wk, K, K, stride, pad, bias=b, relu=True)
        ex = oracle.conv2d(x, wk, K, K, stride, pad, bias=b, relu=True)
        mag = oracle.conv2d(np.abs(x), np.abs(wk), K, K, stride, pad)
        assert np.all(np.abs(y - ex) <= mag * 2.0 ** -20 + 1e-6)
        oracle.lib().orc_f16x3_force_portable(1); oracle.set_threads(1)
        try:
            y0 = oracle.conv2d_f16x3(x, wk, K, K, stride, pad, bias=b, relu=True)
        finally:
            oracle.lib().orc_f16x3_force_portable(0); oracle.set_threads(8)
        assert y0.tobytes() == y.tobytes()
    assert not oracle.uses_f16x3(256, 15, 1, 1) and oracle.uses_f16x3(1024, 105, 1, 1) and oracle.uses_f16x3(256, 36, 3, 3)


def test_f16x3_oracle_reproduces_the_gpu_golden(oracle):
    """oracle/f16x3_oracle.c (the CPU restatement of CALD_PRECISION_F16X3) against bits an MI355X produced in that mode
    (tests/golden/f16x3_gpu_small.npz <- tools/make_golden_f16x3.py, run on the GPU box): stage tensors of one forward by SHA-1, its
    detections, and a 2-image sweep -- so a CPU-only box also sees the f16x3 oracle pinned to the hardware, not only to itself."""
    import hashlib, os
    from conftest import GOLDEN
    from cald_amd import synth
    path = os.path.join(GOLDEN, "f16x3_gpu_small.npz")
    g = np.load(path)
    sd = synth.pseudo_trained_frcnn(21, 50, seed=0)
    P = oracle.prepare_frcnn(sd, 21, 50); P["precision"] = "f16x3"
    pool = synth.make_pool(2, "voc", 0, scale=0.4)
    keep = {}
    want = oracle.frcnn_forward(P, pool[0], 240, 400, keep=keep)
    for name, t in (("conv1", keep["conv1"]), ("P2", keep["fpn"][0]), ("P3", keep["fpn"][1]), ("P4", keep["fpn"][2]), ("P5", keep["fpn"][3]),
                    ("rpn0", keep["rpn_head"][0]), ("rpn4", keep["rpn_head"][4])):
        assert hashlib.sha1(np.ascontiguousarray(t).tobytes()).digest() == g["sha1_" + name].tobytes(), name
    np.testing.assert_array_equal(want["boxes"], g["boxes"]); np.testing.assert_array_equal(want["scores"], g["scores"])
    np.testing.assert_array_equal(want["labels"], g["labels"])
    wc, wk = oracle.get_uncertainty(P, pool, ["flip"], 21, bp=1.3, min_size=240, max_size=400, base_seed=3)
    np.testing.assert_array_equal(np.array(wc), g["consistency"]); np.testing.assert_array_equal(np.stack(wk), g["cls_corr"])
