"""COCO bounding-box AP / AR for the evaluation consumer (COCO side: detection/engine.py:178-256
``coco_evaluate`` -> detection/coco_eval.py ``CocoEvaluator`` -> pycocotools ``COCOeval``).

pycocotools is a third-party dependency of the reference that is absent from this image (no wheel, no network): **parity unpinned**.
What the reference vendors in-repo is followed where it exists -- ``CocoEvaluator.update / accumulate / summarize`` and
``prepare_for_coco_detection`` (coco_eval.py:19-98), ``loadRes`` (:242-297: result ids 1..n, area = w * h), the patched
``evaluate`` loop order (:304-348) -- and the rest restates the published algorithm of pycocotools 2.0 ``cocoeval.py``
(``computeIoU`` with crowd handling as in ``maskApi.c bbIou``, ``evaluateImg``, ``accumulate``, ``summarize``): greedy matching of
score-sorted detections per (image, category, area range) at IoU 0.50:0.05:0.95, crowd / out-of-range ground truth ignored, 101-point
interpolated precision, maxDets (1, 10, 100).  Pinned only by self-consistency and hand-computed cases (tests/test_cpu_host.py).
Only ``iou_type='bbox'`` (the detectors here produce no masks or keypoints).
"""
import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
REC_THRS = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
MAX_DETS = [1, 10, 100]
AREA_RNG = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
AREA_LBL = ["all", "small", "medium", "large"]


class CocoGT(object):
    """The ground-truth side of pycocotools' COCO object as far as bbox evaluation reads it."""

    def __init__(self, images, annotations, categories):
        self.imgs = {im["id"]: im for im in images}
        self.cats = {c["id"]: c for c in categories}
        self.anns = [dict(a) for a in annotations]
        for a in self.anns:
            a["ignore"] = int(a.get("iscrowd", 0))            # COCOeval._prepare: ignore = iscrowd
            a["iscrowd"] = int(a.get("iscrowd", 0))

    def get_cat_ids(self):
        return sorted(self.cats)

    @classmethod
    def from_dataset(cls, dataset):
        """get_coco_api_from_dataset (coco_utils.py:199-208): a cald_amd.coco_utils dataset carries its annotation file; any other
        dataset is converted from its targets (convert_to_coco_api, :146-196: annotation ids from 1, xywh boxes)."""
        for _ in range(10):
            if hasattr(dataset, "anns") and hasattr(dataset, "imgs"):
                break
            if hasattr(dataset, "dataset"):
                dataset = dataset.dataset
        if hasattr(dataset, "anns") and hasattr(dataset, "imgs"):
            anns = [a for lst in dataset.anns.values() for a in lst]
            return cls(list(dataset.imgs.values()), anns, list(dataset.cats.values()) or [{"id": c} for c in sorted({a["category_id"] for a in anns})])
        images, anns, cats, ann_id = [], [], set(), 1
        for idx in range(len(dataset)):
            img, t = dataset[idx]
            image_id = int(t["image_id"].item() if hasattr(t["image_id"], "item") else t["image_id"])
            images.append({"id": image_id, "height": int(img.shape[-2]), "width": int(img.shape[-1])})
            boxes = np.asarray(t["boxes"], np.float64).reshape(-1, 4).copy()
            boxes[:, 2:] -= boxes[:, :2]
            for i in range(len(boxes)):
                lab = int(t["labels"][i])
                cats.add(lab)
                anns.append({"image_id": image_id, "bbox": boxes[i].tolist(), "category_id": lab, "area": float(t["area"][i]),
                             "iscrowd": int(t["iscrowd"][i]), "id": ann_id})
                ann_id += 1
        return cls(images, anns, [{"id": c} for c in sorted(cats)])


def bbox_iou(dts, gts, iscrowd):
    """maskApi.c bbIou on xywh boxes: [D, G] float64; against a crowd ground truth the union is the detection's own area."""
    d = np.asarray(dts, np.float64).reshape(-1, 4); g = np.asarray(gts, np.float64).reshape(-1, 4)
    if len(d) == 0 or len(g) == 0:
        return np.zeros((len(d), len(g)))
    w = np.minimum(d[:, None, 0] + d[:, None, 2], g[None, :, 0] + g[None, :, 2]) - np.maximum(d[:, None, 0], g[None, :, 0])
    h = np.minimum(d[:, None, 1] + d[:, None, 3], g[None, :, 1] + g[None, :, 3]) - np.maximum(d[:, None, 1], g[None, :, 1])
    inter = np.where((w > 0) & (h > 0), w * h, 0.0)
    da, ga = (d[:, 2] * d[:, 3])[:, None], (g[:, 2] * g[:, 3])[None, :]
    union = np.where(np.asarray(iscrowd, bool)[None, :], da + 0 * ga, da + ga - inter)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(inter > 0, inter / union, 0.0)


class COCOeval(object):
    """bbox evaluation over the images seen so far; ``eval['precision']`` is [T, R, K, A, M], ``eval['recall']`` [T, K, A, M]."""

    def __init__(self, coco_gt):
        self.gt = coco_gt
        self.cat_ids = coco_gt.get_cat_ids()
        self._gts = {}
        for a in coco_gt.anns:
            self._gts.setdefault((a["image_id"], a["category_id"]), []).append(a)
        self._dts = {}
        self.img_ids = []
        self.eval, self.stats = {}, []

    def add_results(self, results):
        """loadRes (coco_eval.py:242-297): ids continue from 1, area = w * h."""
        n = sum(len(v) for v in self._dts.values())
        for k, r in enumerate(results):
            bb = r["bbox"]
            d = dict(r, area=bb[2] * bb[3], id=n + k + 1, iscrowd=0)
            self._dts.setdefault((d["image_id"], d["category_id"]), []).append(d)

    def evaluate_img(self, img_id, cat_id, a_rng, max_det):
        gt, dt = self._gts.get((img_id, cat_id), []), self._dts.get((img_id, cat_id), [])
        if len(gt) == 0 and len(dt) == 0:
            return None
        ign = np.array([1 if (g["ignore"] or g["area"] < a_rng[0] or g["area"] > a_rng[1]) else 0 for g in gt], dtype=np.int64)
        gtind = np.argsort(ign, kind="mergesort")
        gt = [gt[i] for i in gtind]
        gt_ig = ign[gtind]
        dtind = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in dtind[:max_det]]
        iscrowd = [int(g["iscrowd"]) for g in gt]
        ious = bbox_iou([d["bbox"] for d in dt], [g["bbox"] for g in gt], iscrowd)
        T, G, D = len(IOU_THRS), len(gt), len(dt)
        gtm, dtm, dt_ig = np.zeros((T, G)), np.zeros((T, D)), np.zeros((T, D))
        if G and D:
            for ti, t in enumerate(IOU_THRS):
                for di, d in enumerate(dt):
                    iou, m = min(t, 1 - 1e-10), -1
                    for gi in range(G):
                        if gtm[ti, gi] > 0 and not iscrowd[gi]:
                            continue                          # already matched, and not a crowd
                        if m > -1 and gt_ig[m] == 0 and gt_ig[gi] == 1:
                            break                             # a regular match exists; the ignored ground truth comes last
                        if ious[di, gi] < iou:
                            continue
                        iou, m = ious[di, gi], gi
                    if m == -1:
                        continue
                    dt_ig[ti, di] = gt_ig[m]; dtm[ti, di] = gt[m]["id"]; gtm[ti, m] = d["id"]
        out_rng = np.array([d["area"] < a_rng[0] or d["area"] > a_rng[1] for d in dt]).reshape(1, D)
        dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(out_rng, T, 0)))
        return {"dtMatches": dtm, "dtScores": [d["score"] for d in dt], "gtIgnore": gt_ig, "dtIgnore": dt_ig}

    def evaluate(self, img_ids):
        """Per-image evaluation of `img_ids`, in the loop order of coco_eval.py:338-345; results are kept per image for accumulate()."""
        img_ids = list(np.unique(img_ids))
        max_det = MAX_DETS[-1]
        for img_id in img_ids:
            if img_id in self.img_ids:
                continue
            self.img_ids.append(img_id)
        self._eval_imgs = getattr(self, "_eval_imgs", {})
        for cat_id in self.cat_ids:
            for ai, a_rng in enumerate(AREA_RNG):
                for img_id in img_ids:
                    self._eval_imgs[(cat_id, ai, img_id)] = self.evaluate_img(img_id, cat_id, a_rng, max_det)

    def accumulate(self):
        T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(self.cat_ids), len(AREA_RNG), len(MAX_DETS)
        precision, recall, scores = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M)), -np.ones((T, R, K, A, M))
        img_ids = sorted(self.img_ids)
        for k, cat_id in enumerate(self.cat_ids):
            for a in range(A):
                E = [self._eval_imgs.get((cat_id, a, i)) for i in img_ids]
                E = [e for e in E if e is not None]
                if len(E) == 0:
                    continue
                for m, max_det in enumerate(MAX_DETS):
                    dt_scores = np.concatenate([e["dtScores"][0:max_det] for e in E])
                    inds = np.argsort(-dt_scores, kind="mergesort")
                    dt_sorted = dt_scores[inds]
                    dtm = np.concatenate([e["dtMatches"][:, 0:max_det] for e in E], axis=1)[:, inds]
                    dt_ig = np.concatenate([e["dtIgnore"][:, 0:max_det] for e in E], axis=1)[:, inds]
                    gt_ig = np.concatenate([e["gtIgnore"] for e in E])
                    npig = np.count_nonzero(gt_ig == 0)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dt_ig))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                    tp_sum = np.cumsum(tps, axis=1).astype(dtype=float)
                    fp_sum = np.cumsum(fps, axis=1).astype(dtype=float)
                    for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        q, ss = np.zeros((R,)), np.zeros((R,))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = pr.tolist(); q = q.tolist()
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        ri_inds = np.searchsorted(rc, REC_THRS, side="left")
                        try:
                            for ri, pi in enumerate(ri_inds):
                                q[ri] = pr[pi]; ss[ri] = dt_sorted[pi]
                        except IndexError:
                            pass
                        precision[t, :, k, a, m] = np.array(q); scores[t, :, k, a, m] = np.array(ss)
        self.eval = {"precision": precision, "recall": recall, "scores": scores, "counts": [T, R, K, A, M]}

    def _summarize(self, ap=1, iou_thr=None, area="all", max_dets=100, quiet=False):
        aind, mind = AREA_LBL.index(area), MAX_DETS.index(max_dets)
        s = self.eval["precision"] if ap == 1 else self.eval["recall"]
        if iou_thr is not None:
            s = s[np.where(iou_thr == IOU_THRS)[0]]
        s = s[:, :, :, aind, mind] if ap == 1 else s[:, :, aind, mind]
        mean_s = -1 if len(s[s > -1]) == 0 else np.mean(s[s > -1])
        if not quiet:
            title, typ = ("Average Precision", "(AP)") if ap == 1 else ("Average Recall", "(AR)")
            iou = "{:0.2f}:{:0.2f}".format(IOU_THRS[0], IOU_THRS[-1]) if iou_thr is None else "{:0.2f}".format(iou_thr)
            print(" {:<18} {} @[ IoU={:<9} | area={:>6s} | maxDets={:>3d} ] = {:0.3f}".format(title, typ, iou, area, max_dets, mean_s))
        return mean_s

    def summarize(self, quiet=False):
        S = self._summarize
        self.stats = np.array([S(1, quiet=quiet), S(1, iou_thr=.5, quiet=quiet), S(1, iou_thr=.75, quiet=quiet),
                               S(1, area="small", quiet=quiet), S(1, area="medium", quiet=quiet), S(1, area="large", quiet=quiet),
                               S(0, max_dets=1, quiet=quiet), S(0, max_dets=10, quiet=quiet), S(0, max_dets=100, quiet=quiet),
                               S(0, area="small", quiet=quiet), S(0, area="medium", quiet=quiet), S(0, area="large", quiet=quiet)])
        return self.stats


def convert_to_xywh(boxes):
    b = np.asarray(boxes, np.float64).reshape(-1, 4)
    return np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], axis=1)


class CocoEvaluator(object):
    """detection/coco_eval.py:19-64 for iou_types == ['bbox']."""

    def __init__(self, coco_gt, iou_types=("bbox",)):
        assert list(iou_types) == ["bbox"], "only bounding-box evaluation is built"
        self.coco_gt, self.iou_types = coco_gt, list(iou_types)
        self.coco_eval = {"bbox": COCOeval(coco_gt)}
        self.img_ids = []

    @staticmethod
    def prepare_for_coco_detection(predictions):
        """coco_eval.py:76-98."""
        out = []
        for original_id, p in predictions.items():
            if len(p) == 0:
                continue
            boxes = convert_to_xywh(p["boxes"].cpu().numpy() if hasattr(p["boxes"], "cpu") else p["boxes"]).tolist()
            scores = [float(s) for s in p["scores"]]; labels = [int(l) for l in p["labels"]]
            out.extend({"image_id": original_id, "category_id": labels[k], "bbox": box, "score": scores[k]} for k, box in enumerate(boxes))
        return out

    def update(self, predictions):
        img_ids = list(np.unique(list(predictions.keys())))
        self.img_ids.extend(img_ids)
        ev = self.coco_eval["bbox"]
        ev.add_results(self.prepare_for_coco_detection(predictions))
        ev.evaluate(img_ids)

    def synchronize_between_processes(self):
        pass                                                  # single process (cald_train.py evaluates on one rank)

    def accumulate(self):
        self.coco_eval["bbox"].accumulate()

    def summarize(self):
        print("IoU metric: bbox")
        return self.coco_eval["bbox"].summarize()
