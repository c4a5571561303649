"""Dataset-conversion fixtures from the IMPORTED reference (TEST INFRASTRUCTURE ONLY; build container only).

    python oracle/make_golden_voc_utils.py   ->  tests/golden/voc_utils.npz

detection/voc_utils.py's ``ConvertVOCtoCOCO.__call__`` (:16-44) is executed as it lies in /root/reference on annotation dicts
in the layout torchvision's ``VOCDetection.parse_voc_xml`` produces (the XML tree as nested dicts with string leaves; the
``object`` entry a list, or -- older torchvision -- a bare dict when the image holds one object; both are fed).  Stored: the
XML text each annotation was written from, the single-object-as-dict flag, and the boxes / labels / ishard / name tensors the
reference returned.
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    ref_harness.load_reference()
    vu = importlib.import_module("detection.voc_utils")
    conv = vu.ConvertVOCtoCOCO()
    classes = conv.CLASSES
    rs = np.random.RandomState(5)
    blob = {"classes": np.array(classes)}
    n_img = 24
    for i in range(n_img):
        stem = ("2008_%06d" % rs.randint(0, 999999)) if i % 3 else ("%06d" % rs.randint(0, 999999))     # VOC2012 / VOC2007 stems
        n_obj = 1 if i < 6 else int(rs.randint(1, 9))
        objs, xml = [], "<annotation><folder>VOC2012</folder><filename>%s.jpg</filename><size><width>500</width><height>375</height><depth>3</depth></size><segmented>0</segmented>" % stem
        for _ in range(n_obj):
            c = int(rs.randint(1, len(classes)))
            x0, y0 = int(rs.randint(1, 400)), int(rs.randint(1, 300))
            x1, y1 = x0 + int(rs.randint(1, 100)), y0 + int(rs.randint(1, 75))
            d = int(rs.rand() < 0.25)
            objs.append({"name": classes[c], "pose": "Unspecified", "truncated": "0", "difficult": str(d),
                         "bndbox": {"xmin": str(x0), "ymin": str(y0), "xmax": str(x1), "ymax": str(y1)}})
            xml += ("<object><name>%s</name><pose>Unspecified</pose><truncated>0</truncated><difficult>%d</difficult>"
                    "<bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>" % (classes[c], d, x0, y0, x1, y1))
        xml += "</annotation>"
        bare = n_obj == 1 and i % 2 == 0
        anno = {"folder": "VOC2012", "filename": stem + ".jpg", "size": {"width": "500", "height": "375", "depth": "3"},
                "segmented": "0", "object": objs[0] if bare else objs}
        _, t = conv(None, dict(image_id=i, annotations=anno))
        blob["xml_%d" % i] = np.array(xml)
        blob["bare_%d" % i] = np.array(bare)
        for k in ("boxes", "labels", "ishard", "name"):
            blob["%s_%d" % (k, i)] = t[k].numpy()
    # detection/transforms.py RandomHorizontalFlip (:27-37) and ToTensor, as cald_train.py's get_transform(train=True) chains them:
    # ten draws of Python's `random` seeded with 11 on a small image + boxes
    import random
    import torch
    T = importlib.import_module("detection.transforms")
    flip = T.RandomHorizontalFlip(0.5)
    random.seed(11)
    img = torch.arange(3 * 4 * 7, dtype=torch.float32).reshape(3, 4, 7) / 100.0
    boxes0 = torch.tensor([[0.0, 1.0, 3.0, 2.0], [2.0, 0.0, 6.0, 3.0]])
    outs_i, outs_b = [], []
    for _ in range(10):
        im2, t2 = flip(img.clone(), {"boxes": boxes0.clone()})
        outs_i.append(im2.numpy()); outs_b.append(t2["boxes"].numpy())
    blob["flip_image_in"], blob["flip_boxes_in"] = img.numpy(), boxes0.numpy()
    blob["flip_images_out"], blob["flip_boxes_out"] = np.stack(outs_i), np.stack(outs_b)
    blob["n"] = np.array(n_img)
    np.savez_compressed(os.path.join(OUT, "voc_utils.npz"), **blob)
    print("wrote", os.path.join(OUT, "voc_utils.npz"))


if __name__ == "__main__":
    main()
