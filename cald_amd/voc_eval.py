"""PASCAL VOC average precision from the results files the HIP detector writes (AP half).

Mirrors detection/voc_eval.py of the reference: ``parse_rec`` (:15-32), ``voc_ap`` (:35-66), ``voc_eval`` (:67-186) and
``_do_python_eval`` (:225-266), same names, arguments and numbers.  What differs is the work done, not the result:

* the reference re-parses every annotation XML for every (class, IoU threshold) pair -- 20 x 10 passes over the test set
  per evaluation; here the annotation set is parsed ONCE (``VocAnnotations``) and shared;
* a detection's best-overlapping ground-truth box (``ovmax``, ``jmax``) does not depend on the IoU threshold, so it is
  computed once per class and only the greedy true/false-positive marking is repeated per threshold.

Host code (numpy float64, like the reference); the detections come from ``cald_amd.engine.voc_detections`` +
``write_voc_results_file``.  Tie order of equal confidences follows ``np.argsort(-confidence)`` exactly as in the reference.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

IOU_THRESHOLDS = (0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95)     # voc_eval.py:246


def parse_rec(filename):
    """One VOC annotation file -> [{'name', 'difficult', 'bbox': [xmin, ymin, xmax, ymax]}] (voc_eval.py:15-32)."""
    objects = []
    for obj in ET.parse(filename).findall('object'):
        box = obj.find('bndbox')
        objects.append({'name': obj.find('name').text,
                        'difficult': int(obj.find('difficult').text),
                        'bbox': [int(box.find(k).text) for k in ('xmin', 'ymin', 'xmax', 'ymax')]})
    return objects


def voc_ap(rec, prec, use_07_metric=False):
    """AP from a precision / recall curve (voc_eval.py:35-66): 11-point VOC07 rule or the area under the precision envelope."""
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = np.max(prec[rec >= t]) if np.sum(rec >= t) != 0 else 0
            ap = ap + p / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]            # precision envelope (running max from the right)
    i = np.where(mrec[1:] != mrec[:-1])[0]                    # points where recall changes
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


class VocAnnotations:
    """The annotation set of one image list, parsed once: per class, per image, the boxes and `difficult` flags."""

    def __init__(self, imagesetfile, annopath):
        with open(imagesetfile, 'r') as f:
            self.imagenames = [x.strip() for x in f.readlines()]
        self.recs = {name: parse_rec(annopath.format(name)) for name in self.imagenames}
        self._per_class = {}

    def of_class(self, classname):
        """{imagename: (bbox float64 [n, 4], difficult bool [n])}, npos = number of non-difficult objects."""
        if classname not in self._per_class:
            table, npos = {}, 0
            for name in self.imagenames:
                objs = [o for o in self.recs[name] if o['name'] == classname]
                bbox = np.array([o['bbox'] for o in objs]).astype(float)
                difficult = np.array([o['difficult'] for o in objs]).astype(bool)
                npos += int(np.sum(~difficult))
                table[name] = (bbox, difficult)
            self._per_class[classname] = (table, npos)
        return self._per_class[classname]


def read_detections(detfile):
    """A results file (image_id score x1 y1 x2 y2 per line) in descending-confidence order, as voc_eval.py:121-139 reads it."""
    with open(detfile, 'r') as f:
        rows = [x.strip().split(' ') for x in f.readlines()]
    image_ids = [r[0] for r in rows]
    confidence = np.array([float(r[1]) for r in rows])
    BB = np.array([[float(z) for z in r[2:]] for r in rows])
    if BB.shape[0] > 0:
        order = np.argsort(-confidence)
        BB = BB[order, :]
        image_ids = [image_ids[i] for i in order]
    return image_ids, BB


def best_overlaps(image_ids, BB, table):
    """Per detection: IoU with its image's best ground-truth box of the class and that box's index (voc_eval.py:144-168;
    the +1 pixel convention).  Independent of the IoU threshold.  ovmax = -inf when the image has no box of the class."""
    nd = len(image_ids)
    ovmax = np.full(nd, -np.inf)
    jmax = np.zeros(nd, dtype=np.int64)
    for d in range(nd):
        gt = table[image_ids[d]][0]
        if gt.size == 0:
            continue
        bb = BB[d, :].astype(float)
        iw = np.maximum(np.minimum(gt[:, 2], bb[2]) - np.maximum(gt[:, 0], bb[0]) + 1., 0.)
        ih = np.maximum(np.minimum(gt[:, 3], bb[3]) - np.maximum(gt[:, 1], bb[1]) + 1., 0.)
        inters = iw * ih
        uni = ((bb[2] - bb[0] + 1.) * (bb[3] - bb[1] + 1.) + (gt[:, 2] - gt[:, 0] + 1.) * (gt[:, 3] - gt[:, 1] + 1.) - inters)
        overlaps = inters / uni
        ovmax[d] = np.max(overlaps)
        jmax[d] = np.argmax(overlaps)
    return ovmax, jmax


def mark_detections(image_ids, ovmax, jmax, table, ovthresh):
    """Greedy marking in confidence order (voc_eval.py:170-178): a detection over the threshold claims its ground-truth box if
    it is free (true positive), is a false positive if the box was already claimed, and is ignored if the box is `difficult`."""
    nd = len(image_ids)
    tp, fp = np.zeros(nd), np.zeros(nd)
    claimed = {name: np.zeros(len(v[1]), dtype=bool) for name, v in table.items()}
    for d in range(nd):
        if ovmax[d] > ovthresh:
            name, j = image_ids[d], jmax[d]
            if not table[name][1][j]:
                if not claimed[name][j]:
                    tp[d] = 1.
                    claimed[name][j] = True
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.
    return tp, fp


def _curve(tp, fp, npos, use_07_metric):
    fp = np.cumsum(fp)
    tp = np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)     # no division by zero when the first match is `difficult`
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def voc_eval(classname, detpath, imagesetfile, annopath='', ovthresh=0.5, use_07_metric=False, annotations=None):
    """rec, prec, ap of one class at one IoU threshold (voc_eval.py:67-186, same positional signature).
    ``annotations``: an already parsed VocAnnotations for (imagesetfile, annopath), to skip the XML pass."""
    ann = annotations if annotations is not None else VocAnnotations(imagesetfile, annopath)
    table, npos = ann.of_class(classname)
    image_ids, BB = read_detections(detpath.format(classname))
    ovmax, jmax = best_overlaps(image_ids, BB, table)
    tp, fp = mark_detections(image_ids, ovmax, jmax, table, ovthresh)
    return _curve(tp, fp, npos, use_07_metric)


def voc_eval_thresholds(classname, detfile, annotations, thresholds=IOU_THRESHOLDS, use_07_metric=False):
    """[(rec, prec, ap)] of one class at every threshold: one read of the file, one overlap pass, one marking per threshold."""
    table, npos = annotations.of_class(classname)
    image_ids, BB = read_detections(detfile)
    ovmax, jmax = best_overlaps(image_ids, BB, table)
    return [_curve(*mark_detections(image_ids, ovmax, jmax, table, t), npos, use_07_metric) for t in thresholds]


def voc_paths(root, year, image_set):
    """(imagesetfile, annopath) of detection/voc_eval.py:226-235."""
    for y in ('2012', '2007'):
        if y in year:
            base = os.path.join(root, 'VOCdevkit/VOC' + y)
            found = (os.path.join(base, 'ImageSets/Main/' + image_set + '.txt'), os.path.join(base, 'Annotations/{:s}.xml'))
    return found


def do_python_eval(data_loader, year, path, root='/tmp', classes=None, quiet=False):
    """detection/voc_eval.py:225-266 ``_do_python_eval``: mAP@[.5:.95], AP50, AP75, mean recall@.5 and per-class AP50 of the
    results files under ``{root}/{path}``; prints the reference's table line and returns the numbers.
    ``data_loader.dataset`` provides ``root``, ``image_set`` and the class list (``_transforms.transforms[0].CLASSES``)."""
    ds = data_loader.dataset
    imagesetfile, annopath = voc_paths(ds.root, year, ds.image_set)
    if classes is None:
        classes = ds._transforms.transforms[0].CLASSES
    ann = VocAnnotations(imagesetfile, annopath)
    ap_cls, rec_cls, ap_75, ap_50, ap_iou = [], [], [], [], []
    for cls in classes:
        if cls == '__background__':
            continue
        filename = os.path.join(root, path, 'det_test_{:s}.txt'.format(cls))
        for iou, (rec, prec, ap) in zip(IOU_THRESHOLDS, voc_eval_thresholds(cls, filename, ann)):
            if len(rec) == 0:
                rec = 0.
            ap_iou.append(ap)
            if iou == 0.5:
                ap_50.append(ap); ap_cls.append(ap); rec_cls.append(np.mean(rec))
            if iou == 0.75:
                ap_75.append(ap)
    line = '{}|{}|{}|{}|'.format(round(np.mean(ap_iou) * 100, 1), round(np.mean(ap_50) * 100, 1),
                                 round(np.mean(ap_75) * 100, 1), round(np.mean(rec_cls) * 100, 1))
    line += ''.join('{}|'.format(round(ap * 100, 1)) for ap in ap_cls)
    if not quiet:
        bar = '=' * 101
        print(bar); print(line); print(bar)
    return {'mAP': float(np.mean(ap_iou)), 'AP50': float(np.mean(ap_50)), 'AP75': float(np.mean(ap_75)),
            'recall50': float(np.mean(rec_cls)), 'ap_per_class': [float(a) for a in ap_cls], 'line': line}


_do_python_eval = do_python_eval      # the reference's name (imported by detection/engine.py:13 through `from .voc_eval import ...`)
