"""TEST INFRASTRUCTURE.  Known-answer vectors for the torchvision 0.8.2 primitives whose source is NOT in /root/reference (RoIAlign,
nms / batched_nms, AnchorGenerator's base anchors, the MultiScaleRoIAlign level mapper, BoxCoder.decode; call sites
detection/frcnn_la.py:37, :50, :76, :199-222, detection/frcnn_ll.py:284-321) -> tests/golden/tv_known_answers.npz.

Every expected value is derived HERE, in float64 (or in explicitly simulated float32 where torchvision's float32 rounding IS the
behaviour), from the published definition of the operation -- this script imports nothing from oracle/ or cald_amd/ and calls none
of their code.  The C oracle and the HIP kernels are both checked against these values (tests/test_oracle_golden.py,
tests/test_gpu_parity.py), so the two no longer only agree with each other.

    python oracle/make_known_answers.py          (writes tests/golden/tv_known_answers.npz)
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


# ----------------------------------------------------------------------------------------------------------------------------
# 1. RoIAlign on affine ramps.  Bilinear interpolation reproduces an affine function exactly, so with f(y, x) = a x + b y + g the
#    value of a sample at (y, x) is f at the CLAMPED point (roi_align's border rule: outside [-1, size] -> 0; <= 0 -> 0;
#    >= size - 1 -> size - 1), and a bin is the mean of its 2 x 2 samples.
# ----------------------------------------------------------------------------------------------------------------------------
def level_of(box):
    """MultiScaleRoIAlign's LevelMapper (k_min 2, k_max 5, canonical scale 224, canonical level 4, eps 1e-6), float64; also returns
    the distance of the un-floored value to the nearest integer (a case is usable only if float32 rounding cannot move it across)."""
    s = np.sqrt((box[2] - box[0]) * (box[3] - box[1]))
    if s == 0.0:
        return 0, np.inf                                              # log2(0) = -inf: clamped to k_min
    val = 4.0 + np.log2(s / 224.0) + 1e-6
    k = int(np.clip(np.floor(val), 2, 5))
    return k - 2, abs(val - np.round(val))


def ramp_coeffs(C):
    c = np.arange(C)
    return (c % 7 - 3) / 8.0, (c % 5 - 2) / 4.0, c / 16.0          # a (per x), b (per y), g


def ramp_level(H, W, C, l):
    a, b, g = ramp_coeffs(C)
    yy, xx = np.mgrid[0:H, 0:W]
    return (a[None, None, :] * xx[:, :, None] + b[None, None, :] * yy[:, :, None] + g[None, None, :] + 10.0 * l).astype(f32)


def roi_align_expected(level_hw, C, rois, value_fn):
    """value_fn(l, yc, xc) -> [C] float64 feature value at a clamped continuous position of level l."""
    out = np.zeros((len(rois), 49, C))
    min_border = np.inf
    for r, box in enumerate(np.asarray(rois, np.float64)):
        l, _ = level_of(box)
        H, W = level_hw[l]
        scale = 1.0 / (4 << l)
        x1, y1, x2, y2 = box * scale
        rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
        bw, bh = rw / 7.0, rh / 7.0
        for ph in range(7):
            for pw in range(7):
                acc = np.zeros(C)
                for iy in range(2):
                    y = y1 + ph * bh + (iy + 0.5) * bh / 2.0
                    for ix in range(2):
                        x = x1 + pw * bw + (ix + 0.5) * bw / 2.0
                        min_border = min(min_border, abs(y + 1), abs(y - H), abs(x + 1), abs(x - W))
                        if y < -1.0 or y > H or x < -1.0 or x > W:
                            continue
                        acc += value_fn(l, min(max(y, 0.0), H - 1.0), min(max(x, 0.0), W - 1.0))
                out[r, ph * 7 + pw] = acc / 4.0
    return out, min_border


def roi_cases():
    level_hw = [(64, 80), (32, 40), (16, 20), (8, 10)]                # P2..P5 of a 256 x 320 padded input
    rois = np.array([
        [40.3, 30.7, 90.9, 77.2],          # inside, level 0
        [10.0, 20.0, 250.0, 200.0],        # level 1 / 2 territory
        [0.0, 0.0, 310.0, 250.0],          # the whole image
        [-40.0, -30.0, 60.0, 50.0],        # hangs off the left / top edge: samples left of -1 read zero, those in [-1, 0] clamp
        [280.0, 200.0, 360.0, 290.0],      # hangs off the right / bottom edge (beyond the map: zero)
        [100.2, 100.2, 100.9, 100.5],      # narrower than one feature pixel: roi_w = roi_h = max(., 1)
        [318.1, 254.3, 319.6, 255.7],      # tiny box in the far corner: samples between size - 1 and size clamp to the last pixel
        [150.0, 10.0, 150.0, 10.0],        # empty box (area 0 -> level 2 by the clamp, one-pixel window)
        [5.5, 7.25, 300.0, 40.0],          # wide and flat
        [-3.9, 100.0, 20.0, 130.0],        # x starts inside (-1, 0) at level 0: -0.975 -> valid, clamped
    ], np.float64)
    return level_hw, rois


# ----------------------------------------------------------------------------------------------------------------------------
# 2. nms / batched_nms through RoIHeads.postprocess_detections: softmax, decode (weights 10, 10, 5, 5), clip, score > 0.05,
#    batched_nms(boxes, scores, labels, 0.5), first 100.  torchvision's batched_nms adds label * (max_coordinate + 1) to the
#    boxes IN FLOAT32 and runs one nms: keep box i, drop later j with IoU(i, j) > thr (strict), areas (x2 - x1)(y2 - y1).
#    The float32 steps below are numpy float32 scalars -- the rounding is part of the behaviour being pinned.
# ----------------------------------------------------------------------------------------------------------------------------
def softmax64(l):
    e = np.exp(l - l.max())
    return e / e.sum()


def decode64(box, d, w=(10.0, 10.0, 5.0, 5.0)):
    width, height = box[2] - box[0], box[3] - box[1]
    cx, cy = box[0] + 0.5 * width, box[1] + 0.5 * height
    dx, dy = d[0] / w[0], d[1] / w[1]
    dw, dh = min(d[2] / w[2], np.log(1000.0 / 16.0)), min(d[3] / w[3], np.log(1000.0 / 16.0))
    pcx, pcy, pw, ph = dx * width + cx, dy * height + cy, np.exp(dw) * width, np.exp(dh) * height
    return np.array([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph])


def iou32(a, b):
    """torchvision nms kernel arithmetic on float32 boxes."""
    xx1, yy1, xx2, yy2 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
    w, h = max(f32(0), f32(xx2 - xx1)), max(f32(0), f32(yy2 - yy1))
    inter = f32(w * h)
    aa, ab = f32(f32(a[2] - a[0]) * f32(a[3] - a[1])), f32(f32(b[2] - b[0]) * f32(b[3] - b[1]))
    return f32(inter / f32(f32(aa + ab) - inter))


def postprocess_expected(logits, deltas, props, Hr, Wr, thr=0.05, nms_thr=0.5, det_max=100):
    R, C = logits.shape
    cand = []
    for r in range(R):
        p = softmax64(logits[r].astype(np.float64))
        for c in range(1, C):
            if p[c] > thr:
                b = decode64(props[r].astype(np.float64), deltas[r, 4 * c:4 * c + 4].astype(np.float64))
                b = np.array([min(max(b[0], 0), Wr), min(max(b[1], 0), Hr), min(max(b[2], 0), Wr), min(max(b[3], 0), Hr)])
                cand.append((p[c], r * (C - 1) + (c - 1), r, c, b.astype(f32)))
    cand.sort(key=lambda t: (-t[0], t[1]))
    maxc = max(float(t[4].max()) for t in cand)
    off = [f32(f32(t[3]) * f32(f32(maxc) + f32(1))) for t in cand]
    ob = [np.array([f32(t[4][k] + o) for k in range(4)], f32) for t, o in zip(cand, off)]
    keep, min_gap = [], np.inf
    for j in range(len(cand)):
        dead = False
        for i in keep:
            v = float(iou32(ob[i], ob[j]))
            if v != nms_thr:
                min_gap = min(min_gap, abs(v - nms_thr))
            if v > nms_thr:
                dead = True
                break
        if not dead:
            keep.append(j)
    keep = keep[:det_max]
    return (np.array([cand[k][4] for k in keep], f32), np.array([cand[k][0] for k in keep]), np.array([cand[k][3] for k in keep], np.int64),
            np.array([cand[k][2] for k in keep], np.int64), min_gap)


def logit_for(p):
    return np.log(p / (1.0 - p))


def nms_case_threshold_and_ties():
    """C = 3.  Group A: IoU exactly 0.5 is kept (strict >), a hair above is dropped.  Group B: equal scores -> the lower candidate
    index survives.  Group C: the same box in two classes -> both kept (batched by label).  Group D: a chain a > b > c where a
    suppresses b, so b does not suppress c (greedy order)."""
    props = np.array([[10, 10, 30, 30], [10, 10, 30, 20], [50, 10, 70, 30], [50, 10, 70, 20.25],
                      [100, 10, 120, 30], [101, 10, 121, 30],
                      [200, 10, 220, 30], [200, 10, 220, 30],
                      [10, 100, 30, 120], [10, 105, 30, 125], [10, 109, 30, 129]], f32)
    R, C = len(props), 3
    logits = np.full((R, C), -20.0, f32); logits[:, 0] = 0.0
    score = [(0, 1, 0.90), (1, 1, 0.80), (2, 1, 0.88), (3, 1, 0.78), (4, 1, 0.70), (5, 1, 0.70), (6, 1, 0.60), (7, 2, 0.65),
             (8, 1, 0.55), (9, 1, 0.50), (10, 1, 0.45)]
    for r, c, p in score:
        logits[r, c] = f32(logit_for(p))
    return logits, np.zeros((R, 4 * C), f32), props, 400, 400


def nms_case_float32_offset():
    """C = 91, coordinates near 2e4, label 90: the offset 90 * (max_coord + 1) = 1.8e6 leaves float32 a spacing of 0.125, so the
    offset boxes are rounded before IoU is taken.  Pair P: exact IoU 0.505 (would be dropped) but the rounded height is 5.0 ->
    IoU exactly 0.5 -> KEPT.  Pair Q (label 1, offset 2e4: spacing 0.002): IoU 0.505 stays above the threshold -> dropped."""
    props = np.array([[20000, 20000, 20010, 20010], [20000, 20000, 20010, 20005.05],
                      [19000, 19000, 19010, 19010], [19000, 19000, 19010, 19005.05]], f32)
    R, C = len(props), 91
    logits = np.full((R, C), -20.0, f32); logits[:, 0] = 0.0
    for r, c, p in [(0, 90, 0.9), (1, 90, 0.8), (2, 1, 0.7), (3, 1, 0.6)]:
        logits[r, c] = f32(logit_for(p))
    return logits, np.zeros((R, 4 * C), f32), props, 30000, 30000


def decode_case():
    """BoxCoder.decode at the log(1000 / 16) clamp: dw = 25 / 5 = 5 is clamped (width x 62.5), dh = 20 / 5 = 4 is not (height x e^4)."""
    props = np.array([[2000, 2000, 2010, 2020]], f32)
    C = 2
    logits = np.array([[0.0, 5.0]], f32)
    deltas = np.zeros((1, 4 * C), f32); deltas[0, 4:8] = [1.0, -2.0, 25.0, 20.0]
    return logits, deltas, props, 5000, 5000


# ----------------------------------------------------------------------------------------------------------------------------
# 3. AnchorGenerator.generate_anchors(scales = (s,), aspect_ratios = (0.5, 1, 2)): h_r = sqrt(ar), w_r = 1 / h_r,
#    base = round([-w, -h, w, h] / 2) -- the published torchvision anchor table.
# ----------------------------------------------------------------------------------------------------------------------------
def base_anchor_table():
    out = []
    for s in (32, 64, 128, 256, 512):
        for ar in (0.5, 1.0, 2.0):
            h_r = np.sqrt(ar); w_r = 1.0 / h_r
            ws, hs = w_r * s, h_r * s
            out.append(np.round(np.array([-ws, -hs, ws, hs]) / 2.0))
    return np.array(out).reshape(5, 3, 4)


# ----------------------------------------------------------------------------------------------------------------------------
# 4. LevelMapper edges: square boxes (0, 0, s, s) with s = 112 * 2^j * (1 + m ulp) -- the + 1e-6 keeps a side ONE float32 step below
#    the boundary on the upper level; 64 steps below is on the lower level; and the clamps at both ends.
# ----------------------------------------------------------------------------------------------------------------------------
def level_cases():
    sides = []
    for j in range(3):
        base = f32(112.0 * 2 ** j)
        for m in (-64, -1, 0, 1):
            s = base
            for _ in range(abs(m)):
                s = np.nextafter(s, f32(np.inf) if m > 0 else f32(0))
            sides.append(float(s))
    sides += [1.0, 20.0, 111.0, 113.0, 300.0, 447.0, 449.0, 700.0, 895.0, 897.0, 1000.0]
    rois = np.array([[0.0, 0.0, s, s] for s in sides], np.float64)
    lv, margin = zip(*[level_of(b) for b in rois])
    assert min(margin) > 5e-7, "a level case sits within float32 noise of its boundary"
    return rois, np.array(lv, np.int64)


def main():
    out = {}
    level_hw, rois = roi_cases()
    for C in (8, 256):
        a, b, g = ramp_coeffs(C)
        exp, border = roi_align_expected(level_hw, C, rois, lambda l, y, x: a * x + b * y + g + 10.0 * l)
        assert border > 1e-3, "a RoIAlign sample sits on the border rule's discontinuity"
        out["roi_expected_c%d" % C] = exp
    out["roi_level_hw"] = np.array(level_hw, np.int64); out["roi_rois"] = rois.astype(f32)
    for name, case in (("nms_a", nms_case_threshold_and_ties), ("nms_b", nms_case_float32_offset), ("decode", decode_case)):
        logits, deltas, props, Hr, Wr = case()
        boxes, scores, labels, src, gap = postprocess_expected(logits, deltas, props, Hr, Wr)
        assert gap > 1e-5, (name, gap)          # apart from the deliberate exact ties, no decision within float32 noise
        out[name + "_logits"], out[name + "_deltas"], out[name + "_props"], out[name + "_hw"] = logits, deltas, props, np.array([Hr, Wr], np.int64)
        out[name + "_boxes"], out[name + "_scores"], out[name + "_labels"], out[name + "_src"] = boxes, scores, labels, src
    # what the cases must demonstrate, asserted on the derived answers themselves
    la, sa = out["nms_a_labels"], out["nms_a_src"]
    assert list(sa) == [0, 2, 1, 4, 7, 6, 8, 10], list(sa)      # r3 (IoU > .5), r5 (tie, higher index), r9 (suppressed by r8) are gone; r10 survives
    assert list(out["nms_b_src"]) == [0, 1, 2], list(out["nms_b_src"])     # pair P both kept (float32 offset), pair Q's second dropped
    d = out["decode_boxes"][0]
    assert abs((d[2] - d[0]) - 625.0) < 1e-2 and abs((d[3] - d[1]) - 20.0 * np.exp(4.0)) < 1e-2
    out["base_anchors"] = base_anchor_table()
    assert out["base_anchors"][0].tolist() == [[-23, -11, 23, 11], [-16, -16, 16, 16], [-11, -23, 11, 23]]
    lr, lv = level_cases()
    out["level_rois"], out["level_expected"] = lr.astype(f32), lv
    assert np.array_equal(lr.astype(f32).astype(np.float64), lr)
    path = os.path.join(ROOT, "tests", "golden", "tv_known_answers.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
