"""COCO target-conversion fixtures from the IMPORTED reference (TEST INFRASTRUCTURE ONLY; build container only).

    python oracle/make_golden_coco_utils.py   ->  tests/golden/coco_utils.npz

detection/coco_utils.py ``ConvertCocoPolysToMask.__call__`` (:49-100) and the validity rule of
``_coco_remove_images_without_annotations`` (:103-131, through a dataset stand-in) are executed as they lie in /root/reference on
synthetic annotation lists.  pycocotools is absent: its two mask functions (frPyObjects / decode), which only feed the ``masks`` entry
that is not part of the fixture, are replaced by a stand-in returning empty masks.  Stored: the annotation JSON text per image, image
sizes, and the boxes / labels / area / iscrowd tensors the reference returned, plus which images its training-set filter keeps.
"""
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


class _Img(object):
    def __init__(self, w, h):
        self.size = (w, h)


def main():
    ref_harness.load_reference()
    cu = importlib.import_module("detection.coco_utils")
    cu.coco_mask.frPyObjects = lambda polygons, h, w: (h, w)
    cu.coco_mask.decode = lambda rles: np.zeros((rles[0], rles[1], 1), np.uint8)
    conv = cu.ConvertCocoPolysToMask()
    rs = np.random.RandomState(4)
    blob, n_img = {}, 16
    ann_id = 1
    keep_flags = []
    for i in range(n_img):
        w, h = int(rs.randint(60, 640)), int(rs.randint(60, 480))
        anno = []
        for _ in range(int(rs.randint(0, 7)) if i else 0):          # image 0: no annotation at all
            kind = rs.rand()
            bw, bh = rs.rand() * w * 0.7, rs.rand() * h * 0.7
            if kind < 0.15:
                bw = 0.0                                                 # degenerate: removed by `keep`
            elif kind < 0.3:
                bw, bh = 0.6, 0.8                                        # "close to zero area" for the training-set filter
            x, y = rs.rand() * w - 10, rs.rand() * h - 10               # may stick out of the image: clamped
            anno.append({"id": ann_id, "image_id": 100 + i, "category_id": int(rs.randint(1, 91)), "iscrowd": int(rs.rand() < 0.15),
                         "bbox": [round(float(x), 2), round(float(y), 2), round(float(bw), 2), round(float(bh), 2)],
                         "area": round(float(bw * bh), 2), "segmentation": [[0.0, 0.0, 1.0, 0.0, 1.0, 1.0]]})
            ann_id += 1
        _, t = conv(_Img(w, h), dict(image_id=100 + i, annotations=[dict(a) for a in anno]))
        blob["anno_%d" % i] = np.array(json.dumps(anno)); blob["size_%d" % i] = np.array([w, h])
        for k in ("boxes", "labels", "area", "iscrowd", "image_id"):
            blob["%s_%d" % (k, i)] = t[k].numpy()
        # the training-set filter's verdict for this image, through the reference function on a one-image dataset stand-in
        class _Coco(object):
            def getAnnIds(self, imgIds, iscrowd=None):
                return [0]
            def loadAnns(self, ids, _a=anno):
                return _a
        class _DS(cu.torchvision.datasets.CocoDetection):
            ids = [100 + i]; coco = _Coco()
        kept = cu._coco_remove_images_without_annotations(_DS())
        keep_flags.append(len(kept.indices) == 1)
    blob["kept_by_train_filter"] = np.array(keep_flags); blob["n"] = np.array(n_img)
    np.savez_compressed(os.path.join(OUT, "coco_utils.npz"), **blob)
    print("wrote", os.path.join(OUT, "coco_utils.npz"), "kept", int(np.sum(keep_flags)), "of", n_img)


if __name__ == "__main__":
    main()
