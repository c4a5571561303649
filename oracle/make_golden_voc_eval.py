"""AP fixtures from the IMPORTED reference (TEST INFRASTRUCTURE ONLY; build container only).

    python oracle/make_golden_voc_eval.py   ->  tests/golden/voc_eval.npz

detection/voc_eval.py's ``voc_eval`` (:67-186), ``voc_ap`` (:35-66) and ``_do_python_eval`` (:225-266) are executed as they lie
in /root/reference on a synthetic VOCdevkit tree (annotation XML files + ImageSets list) and on results files written by the
reference's own ``_write_voc_results_file`` (:188-222).  Stored: the annotation table, the results-file text and every
(rec, prec, ap) the reference returned, plus the table line ``_do_python_eval`` printed.  Harness patch: ``np.bool`` (removed
from numpy >= 1.24, used at voc_eval.py:110) is aliased to ``bool``.
"""
import contextlib
import importlib
import io
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
CLASSES = ('__background__', 'aeroplane', 'bicycle', 'bird', 'boat')
IOUS = [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]


def write_tree(root, names, objects):
    """objects: rows (image idx, class idx, difficult, xmin, ymin, xmax, ymax)"""
    base = os.path.join(root, "VOCdevkit", "VOC2012")
    os.makedirs(os.path.join(base, "ImageSets", "Main")); os.makedirs(os.path.join(base, "Annotations"))
    with open(os.path.join(base, "ImageSets", "Main", "test.txt"), "w") as f:
        f.write("".join(n + "\n" for n in names))
    for i, n in enumerate(names):
        rows = [o for o in objects if o[0] == i]
        xml = "<annotation><filename>%s.jpg</filename>" % n
        for (_, c, diff, x0, y0, x1, y1) in rows:
            xml += ("<object><name>%s</name><pose>Unspecified</pose><truncated>0</truncated><difficult>%d</difficult>"
                    "<bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>" % (CLASSES[c], diff, x0, y0, x1, y1))
        xml += "</annotation>"
        with open(os.path.join(base, "Annotations", n + ".xml"), "w") as f:
            f.write(xml)


def main():
    ref_harness.load_reference()
    np.bool = bool                                  # voc_eval.py:110 (numpy < 1.24 API)
    ve = importlib.import_module("detection.voc_eval")
    rs = np.random.RandomState(11)
    names = ["2009_%06d" % i for i in range(40)]
    objects = []
    for i in range(len(names)):
        for _ in range(rs.randint(0, 5)):
            c = int(rs.randint(1, 4))              # class 4 ('boat') never annotated: npos == 0
            x0, y0 = rs.randint(1, 300), rs.randint(1, 200)
            objects.append((i, c, int(rs.rand() < 0.2), x0, y0, x0 + rs.randint(20, 180), y0 + rs.randint(20, 150)))
    # detections: jittered copies of ground truth (some duplicated -> second match is a false positive), random boxes, scores
    # rounded so that the 3-decimal results files hold exact confidence ties; class 3 ('bird') has no detections at all
    all_boxes = [[] for _ in CLASSES]
    for i in range(len(names)):
        for c in range(len(CLASSES)):
            rows = []
            if c in (1, 2, 4):
                for o in objects:
                    if o[0] == i and (o[1] == c or (c == 4 and rs.rand() < 0.3)) and rs.rand() < 0.85:
                        for _ in range(1 + int(rs.rand() < 0.25)):
                            j = rs.randn(4) * rs.choice([2.0, 10.0, 30.0])
                            rows.append([o[3] + j[0] - 1, o[4] + j[1] - 1, o[5] + j[2] - 1, o[6] + j[3] - 1, np.round(rs.rand(), 2)])
                for _ in range(rs.randint(0, 3)):
                    x0, y0 = rs.rand() * 300, rs.rand() * 200
                    rows.append([x0, y0, x0 + 30 + rs.rand() * 100, y0 + 30 + rs.rand() * 100, np.round(rs.rand() * 0.6, 2)])
            all_boxes[c].append([torch.tensor(rows, dtype=torch.float32)] if rows else [])
    blob = {"classes": np.array(CLASSES), "names": np.array(names), "objects": np.array(objects, np.int64), "ious": np.array(IOUS)}
    with tempfile.TemporaryDirectory() as root:
        write_tree(root, names, objects)
        path = "cald_golden_voc_eval_%d" % os.getpid()
        ve._write_voc_results_file([list(b) for b in all_boxes], list(names), path, CLASSES)
        imagesetfile = os.path.join(root, "VOCdevkit/VOC2012/ImageSets/Main/test.txt")
        annopath = os.path.join(root, "VOCdevkit/VOC2012/Annotations/{:s}.xml")
        for c, cls in enumerate(CLASSES):
            if c == 0:
                continue
            fn = "/tmp/%s/det_test_%s.txt" % (path, cls)
            blob["det_%s" % cls] = np.array(open(fn).read())
            for use07 in (False, True):
                for t in IOUS:
                    with np.errstate(all="ignore"):
                        rec, prec, ap = ve.voc_eval(cls, fn, imagesetfile, annopath, ovthresh=t, use_07_metric=use07)
                    key = "%s_%d_%d" % (cls, int(round(t * 100)), int(use07))
                    blob["rec_" + key], blob["prec_" + key], blob["ap_" + key] = np.asarray(rec, np.float64), np.asarray(prec, np.float64), np.float64(ap)
        loader = SimpleNamespace(dataset=SimpleNamespace(root=root, image_set="test",
                                                         _transforms=SimpleNamespace(transforms=[SimpleNamespace(CLASSES=CLASSES)])))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), np.errstate(all="ignore"):
            ve._do_python_eval(loader, "2012", path)
        blob["python_eval_stdout"] = np.array(buf.getvalue())
        loader3 = SimpleNamespace(dataset=SimpleNamespace(root=root, image_set="test",
                                                          _transforms=SimpleNamespace(transforms=[SimpleNamespace(CLASSES=CLASSES[:4])])))
        buf3 = io.StringIO()      # without the never-annotated class (its 0/0 recall turns every mean into nan)
        with contextlib.redirect_stdout(buf3), np.errstate(all="ignore"):
            ve._do_python_eval(loader3, "2012", path)
        blob["python_eval_stdout_3cls"] = np.array(buf3.getvalue())
        import shutil
        shutil.rmtree("/tmp/" + path)
    np.savez_compressed(os.path.join(OUT, "voc_eval.npz"), **blob)
    print("voc_eval ok:", buf.getvalue().strip().splitlines()[1], "|", buf3.getvalue().strip().splitlines()[1])


if __name__ == "__main__":
    main()
