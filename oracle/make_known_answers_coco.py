"""TEST INFRASTRUCTURE.  Known answers for COCO bounding-box AP / AR (pycocotools 2.0 COCOeval, which the reference calls through
detection/coco_eval.py:19-64 and detection/engine.py:178-256 and which is NOT installed here) -> tests/golden/coco_ap_known_answers.npz.

The scene is DESIGNED so that what every detection is at every IoU threshold follows from its construction, not from running a matcher:
each detection is either a ground-truth box shrunk in width to a chosen IoU q (a true positive exactly at the thresholds t <= q), a box
that overlaps nothing (a false positive), or a box inside a crowd region (ignored).  This script turns that table into the twelve COCO
statistics with the published definitions only -- cumulative precision / recall over the score-sorted detections of a category, the
monotone precision envelope sampled at recall 0:0.01:1, area ranges [0, 32^2, 96^2, 1e10] (ground truth outside the range and whatever
matches it are ignored; unmatched detections outside the range are ignored), maxDets (1, 10, 100) per image and category -- in float64.
It imports nothing from cald_amd/ or oracle/ and calls none of their code.

    python oracle/make_known_answers_coco.py
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = [0.5 + 0.05 * i for i in range(10)]
REC = [i / 100.0 for i in range(101)]
AREAS = {"all": (0.0, 1e10), "small": (0.0, 32.0 ** 2), "medium": (32.0 ** 2, 96.0 ** 2), "large": (96.0 ** 2, 1e10)}

IMAGES = [{"id": 11, "width": 640, "height": 480}, {"id": 12, "width": 640, "height": 480}, {"id": 13, "width": 640, "height": 480}]
CATS = [{"id": 1, "name": "a"}, {"id": 2, "name": "b"}]
# ground truth: name -> (image, category, xywh, crowd)
GT = {"g1": (11, 1, [10, 10, 100, 100], 0), "g2": (11, 1, [200, 10, 60, 60], 0), "g3": (12, 1, [10, 10, 20, 20], 0),
      "g4": (12, 1, [300, 200, 120, 100], 0), "g5": (13, 1, [50, 50, 50, 50], 0), "gc": (13, 1, [300, 300, 200, 100], 1),
      "h1": (11, 2, [400, 300, 80, 80], 0), "h2": (13, 2, [10, 300, 30, 30], 0)}
# detections: (score, image, category, kind, target, q): kind "tp" = the target shrunk in width to IoU q; "fp" = overlaps nothing (xywh
# given as target); "crowd" = inside the crowd region (xywh given as target)
DT = [(0.95, 11, 1, "tp", "g1", 0.97), (0.90, 12, 1, "tp", "g4", 0.72), (0.85, 11, 1, "fp", [500, 400, 50, 50], None),
      (0.80, 11, 1, "tp", "g2", 0.83), (0.75, 13, 1, "crowd", [320, 310, 100, 80], None), (0.70, 12, 1, "tp", "g3", 0.61),
      (0.65, 13, 1, "tp", "g5", 0.52), (0.60, 13, 1, "fp", [150, 50, 15, 50], None),
      (0.92, 11, 2, "tp", "h1", 0.88), (0.55, 13, 2, "tp", "h2", 0.66), (0.50, 12, 2, "fp", [100, 100, 200, 150], None)]


def det_box(d):
    if d[3] == "tp":
        x, y, w, h = GT[d[4]][2]
        return [x, y, w * d[5], h]                 # same corner, width * q: IoU = q
    return list(d[4])


def in_range(area, rng):
    return rng[0] <= area <= rng[1]


def stats():
    prec = {}                                       # (t index, category, area, maxdet) -> [101] or None (no ground truth in range)
    rec = {}
    for cat in (1, 2):
        for an, rng in AREAS.items():
            npos = sum(1 for g in GT.values() if g[1] == cat and not g[3] and in_range(g[2][2] * g[2][3], rng))
            for md in (1, 10, 100):
                dets = []
                for img in (11, 12, 13):            # the maxDets highest-scored detections of the category in each image
                    mine = sorted([d for d in DT if d[1] == img and d[2] == cat], key=lambda d: -d[0])[:md]
                    dets += mine
                dets.sort(key=lambda d: -d[0])
                for ti, t in enumerate(T):
                    if npos == 0:
                        prec[(ti, cat, an, md)] = None; rec[(ti, cat, an, md)] = None
                        continue
                    tp = fp = 0
                    pr, rc = [], []
                    for d in dets:
                        b = det_box(d); area = b[2] * b[3]
                        if d[3] == "crowd":
                            continue                                        # matched to the crowd region: ignored at every threshold
                        if d[3] == "tp" and d[5] >= t - 1e-9:
                            g = GT[d[4]]
                            if not in_range(g[2][2] * g[2][3], rng):
                                continue                                    # matched to ground truth outside the area range: ignored
                            tp += 1
                        else:
                            if not in_range(area, rng):
                                continue                                    # unmatched and outside the area range: ignored
                            fp += 1
                        pr.append(tp / (tp + fp + np.spacing(1))); rc.append(tp / npos)
                    for i in range(len(pr) - 1, 0, -1):                     # monotone envelope
                        pr[i - 1] = max(pr[i - 1], pr[i])
                    q = []
                    for r in REC:                                           # first index whose recall reaches r
                        k = next((i for i, v in enumerate(rc) if v >= r), None)
                        q.append(pr[k] if k is not None else 0.0)
                    prec[(ti, cat, an, md)] = q; rec[(ti, cat, an, md)] = rc[-1] if rc else 0.0

    def mean_p(tis, an, md):
        v = [x for ti in tis for cat in (1, 2) if prec[(ti, cat, an, md)] is not None for x in prec[(ti, cat, an, md)]]
        return float(np.mean(v)) if v else -1.0

    def mean_r(an, md):
        v = [rec[(ti, cat, an, md)] for ti in range(10) for cat in (1, 2) if rec[(ti, cat, an, md)] is not None]
        return float(np.mean(v)) if v else -1.0
    allt = range(10)
    return np.array([mean_p(allt, "all", 100), mean_p([0], "all", 100), mean_p([5], "all", 100), mean_p(allt, "small", 100),
                     mean_p(allt, "medium", 100), mean_p(allt, "large", 100), mean_r("all", 1), mean_r("all", 10), mean_r("all", 100),
                     mean_r("small", 100), mean_r("medium", 100), mean_r("large", 100)])


def main():
    anns = [{"id": i + 1, "image_id": g[0], "category_id": g[1], "bbox": [float(v) for v in g[2]], "area": float(g[2][2] * g[2][3]), "iscrowd": g[3]}
            for i, g in enumerate(GT.values())]
    dets = [{"image_id": d[1], "category_id": d[2], "score": d[0], "bbox": [float(v) for v in det_box(d)]} for d in DT]
    # the construction's own premises, checked in float64: a shrunk box has IoU q with its target and overlaps no other ground truth of
    # its category; "fp" boxes overlap nothing; q is never within 1e-6 of a threshold
    def iou(a, b):
        w = min(a[0] + a[2], b[0] + b[2]) - max(a[0], b[0]); h = min(a[1] + a[3], b[1] + b[3]) - max(a[1], b[1])
        inter = w * h if w > 0 and h > 0 else 0.0
        return inter / (a[2] * a[3] + b[2] * b[3] - inter)
    for d in DT:
        b = det_box(d)
        for name, g in GT.items():
            if g[0] != d[1] or g[1] != d[2]:
                continue
            v = iou(b, g[2])
            if d[3] == "tp" and name == d[4]:
                assert abs(v - d[5]) < 1e-12 and min(abs(d[5] - t) for t in T) > 1e-3
            elif d[3] == "crowd" and name == "gc":
                assert b[0] >= g[2][0] and b[1] >= g[2][1] and b[0] + b[2] <= g[2][0] + g[2][2] and b[1] + b[3] <= g[2][1] + g[2][3]
            else:
                assert v == 0.0, (d, name, v)
    s = stats()
    path = os.path.join(ROOT, "tests", "golden", "coco_ap_known_answers.npz")
    np.savez_compressed(path, images=json.dumps(IMAGES), categories=json.dumps(CATS), annotations=json.dumps(anns), detections=json.dumps(dets), stats=s)
    print("wrote", path); print(np.round(s, 6))


if __name__ == "__main__":
    main()
