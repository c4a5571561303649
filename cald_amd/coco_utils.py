"""The COCO side of the dataset formats (BASELINE configs[3] / [4] run on COCO 2017).

Mirrors detection/coco_utils.py without torchvision / pycocotools: ``ConvertCocoPolysToMask`` (:49-100: crowd objects dropped, xywh ->
xyxy, clamped to the image, degenerate boxes removed; ``labels`` = category ids, ``area`` / ``iscrowd`` of the non-crowd objects),
``CocoDetection`` (:211-222 over torchvision's: image ids sorted, annotations in file order), the training-set filter
``_coco_remove_images_without_annotations`` (:103-143) and ``get_coco`` (:225-249).  The annotation JSON is read directly.
Segmentation masks are NOT produced (they need pycocotools' RLE code and the detectors here never read them); keypoints are passed
through as in the reference.  Like cald_amd.voc_utils, a dataset hands the sweep an HBM-resident pool (``device_pool``) and the
selection stage a label loader.
"""
import json
import os

import torch


class ConvertCocoPolysToMask(object):
    """detection/coco_utils.py:49-100 minus ``masks``."""

    def __call__(self, image, target):
        w, h = image.size if hasattr(image, "size") and not callable(image.size) else (target["width"], target["height"])
        image_id = torch.tensor([target["image_id"]])
        anno = [obj for obj in target["annotations"] if obj["iscrowd"] == 0]
        boxes = torch.as_tensor([obj["bbox"] for obj in anno], dtype=torch.float32).reshape(-1, 4)
        boxes[:, 2:] += boxes[:, :2]
        boxes[:, 0::2].clamp_(min=0, max=w)
        boxes[:, 1::2].clamp_(min=0, max=h)
        classes = torch.tensor([obj["category_id"] for obj in anno], dtype=torch.int64)
        keypoints = None
        if anno and "keypoints" in anno[0]:
            keypoints = torch.as_tensor([obj["keypoints"] for obj in anno], dtype=torch.float32)
            if keypoints.shape[0]:
                keypoints = keypoints.view(keypoints.shape[0], -1, 3)
        keep = (boxes[:, 3] > boxes[:, 1]) & (boxes[:, 2] > boxes[:, 0])
        out = {"boxes": boxes[keep], "labels": classes[keep], "image_id": image_id}
        if keypoints is not None:
            out["keypoints"] = keypoints[keep]
        out["area"] = torch.tensor([obj["area"] for obj in anno])          # of the non-crowd objects, NOT filtered by `keep` (as the reference)
        out["iscrowd"] = torch.tensor([obj["iscrowd"] for obj in anno])
        return image, out


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target):
        for t in self.transforms:
            image, target = t(image, target)
        return image, target


class CocoDetection(object):
    """torchvision.datasets.CocoDetection + detection/coco_utils.py:211-222 over an instances_*.json file."""

    def __init__(self, img_folder, ann_file, transforms):
        self.root, self._transforms = img_folder, transforms
        with open(ann_file) as f:
            data = json.load(f)
        self.imgs = {im["id"]: im for im in data["images"]}
        self.anns = {}
        for a in data.get("annotations", []):                       # COCO.getAnnIds(imgIds=id): the image's annotations in file order
            self.anns.setdefault(a["image_id"], []).append(a)
        self.cats = {c["id"]: c for c in data.get("categories", [])}
        self.ids = sorted(self.imgs)

    def __len__(self):
        return len(self.ids)

    def path(self, idx):
        return os.path.join(self.root, self.imgs[self.ids[idx]]["file_name"])

    def annotations(self, idx):
        return self.anns.get(self.ids[idx], [])

    def __getitem__(self, idx):
        from PIL import Image
        img = Image.open(self.path(idx)).convert("RGB")
        target = dict(image_id=self.ids[idx], annotations=self.annotations(idx))
        if self._transforms is not None:
            img, target = self._transforms(img, target)
        return img, target

    def target(self, idx):
        """The converted target without decoding the image (sizes come from the JSON's width / height)."""
        info = self.imgs[self.ids[idx]]

        class _Size(object):
            size = (info["width"], info["height"])
        return ConvertCocoPolysToMask()(_Size(), dict(image_id=self.ids[idx], annotations=self.annotations(idx)))[1]


class Subset(object):
    """torch.utils.data.Subset with the pool helpers of the datasets here."""

    def __init__(self, dataset, indices):
        self.dataset, self.indices = dataset, list(indices)

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        return self.dataset[self.indices[i]]

    def path(self, i):
        return self.dataset.path(self.indices[i])

    def target(self, i):
        return self.dataset.target(self.indices[i])

    def device_pool(self, indices=None):
        return device_pool(self, indices)

    def label_loader(self, indices):
        return label_loader(self, indices)


def device_pool(dataset, indices=None):
    """JPEG files decoded once on the GPU into an HBM-resident pool (cald_amd.pool.DevicePool); ``pool.loader()`` feeds get_uncertainty."""
    from .pool import DevicePool
    idx = range(len(dataset)) if indices is None else indices
    return DevicePool.from_files([dataset.path(int(i)) for i in idx])


def label_loader(dataset, indices):
    """[(None, (target,))]: the labeled loader of cald_train.py:434-444 as far as cls_kldiv reads it."""
    return [(None, (dataset.target(int(i)),)) for i in indices]


CocoDetection.device_pool = device_pool


def resident_train_loader(dataset, batch_sampler, indices, pool=None, flip_prob=0.5):
    """voc_utils.ResidentTrainLoader over a COCO dataset (or Subset): HBM-resident training batches, flip on the device."""
    from .voc_utils import ResidentTrainLoader
    return ResidentTrainLoader(dataset, batch_sampler, indices, pool, flip_prob)


CocoDetection.resident_train_loader = resident_train_loader
Subset.resident_train_loader = resident_train_loader
CocoDetection.label_loader = label_loader


def _coco_remove_images_without_annotations(dataset, cat_list=None):
    """detection/coco_utils.py:103-143: keep images with at least one annotation whose box is not (nearly) empty; keypoint sets need
    >= 10 visible keypoints."""
    def valid(anno):
        if len(anno) == 0:
            return False
        if all(any(o <= 1 for o in obj["bbox"][2:]) for obj in anno):
            return False
        if "keypoints" not in anno[0]:
            return True
        return sum(sum(1 for v in ann["keypoints"][2::3] if v > 0) for ann in anno) >= 10
    ids = []
    for i in range(len(dataset)):
        anno = dataset.annotations(i)
        if cat_list:
            anno = [obj for obj in anno if obj["category_id"] in cat_list]
        if valid(anno):
            ids.append(i)
    return Subset(dataset, ids)


def get_coco(root, image_set, transforms, mode="instances"):
    """detection/coco_utils.py:225-249."""
    paths = {"train": ("train2017", os.path.join("annotations", "%s_train2017.json" % mode)),
             "val": ("val2017", os.path.join("annotations", "%s_val2017.json" % mode))}
    t = [ConvertCocoPolysToMask()]
    if transforms is not None:
        t.append(transforms)
    img_folder, ann_file = paths[image_set]
    dataset = CocoDetection(os.path.join(root, img_folder), os.path.join(root, ann_file), transforms=Compose(t))
    if image_set == "train":
        dataset = _coco_remove_images_without_annotations(dataset)
    return dataset
