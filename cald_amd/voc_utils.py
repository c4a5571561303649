"""The dataset side of the sweep and of the evaluation consumer (data formats either side of the path).

Mirrors detection/voc_utils.py: ``ConvertVOCtoCOCO`` (:7-44: boxes - 1 as float32, class name -> index, ``ishard``, the file stem as
an int8 ``name`` tensor), ``VOCDetection`` (:47-58: ``(image, dict(image_id, annotations))`` through the transforms) and
``get_voc2012`` / ``get_voc2007`` (:61-82) -- without torchvision: the VOCdevkit tree (ImageSets/Main/<set>.txt,
Annotations/<id>.xml, JPEGImages/<id>.jpg) is read directly, annotations come out in the dict layout of torchvision's
``parse_voc_xml`` (what the reference's ConvertVOCtoCOCO consumes).

What it feeds:
  * ``cald_amd.sweep.get_uncertainty`` -- ``dataset.device_pool(indices)`` decodes the JPEG files ONCE on the GPU into an HBM-resident
    pool (``cald_amd.pool.DevicePool``), or the dataset goes through a torch DataLoader exactly as in cald_train.py:434;
  * ``cald_amd.sweep.cls_kldiv`` -- the labeled loader's ``target['labels']`` (cald_train.py:237-242);
  * ``cald_amd.engine.voc_evaluate`` -- ``target['name']``, ``dataset.root / image_set / _transforms.transforms[0].CLASSES``.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np
import torch

VOC_CLASSES = ("__background__", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow",
               "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor")


def parse_voc_xml(node):
    """An ElementTree node as nested dicts, the layout of torchvision's VOCDetection.parse_voc_xml: leaves are their text, repeated
    children become lists, and ``annotation/object`` is always a list."""
    children = list(node)
    if not children:
        return {node.tag: (node.text.strip() if node.text else node.text)}
    merged = {}
    for child in children:
        for k, v in parse_voc_xml(child).items():
            merged.setdefault(k, []).append(v)
    out = {k: (v[0] if len(v) == 1 else v) for k, v in merged.items()}
    if node.tag == "annotation" and "object" in out and not isinstance(out["object"], list):
        out["object"] = [out["object"]]
    return {node.tag: out}


class ConvertVOCtoCOCO(object):
    """detection/voc_utils.py:7-44.  target in: {'image_id', 'annotations': parse_voc_xml(...)['annotation']};
    target out: boxes float32 [n, 4] (VOC corners - 1), labels int64 [n], ishard int64 [n], name int8 [len(stem)]."""
    CLASSES = VOC_CLASSES

    def __call__(self, image, target):
        anno = target["annotations"]
        stem = anno["filename"].split(".")[0]
        objects = anno["object"] if isinstance(anno["object"], list) else [anno["object"]]
        corners = ("xmin", "ymin", "xmax", "ymax")
        boxes = [[int(o["bndbox"][c]) - 1 for c in corners] for o in objects]
        labels = [self.CLASSES.index(o["name"]) for o in objects]
        hard = [int(o["difficult"]) for o in objects]
        out = {"boxes": torch.as_tensor(boxes, dtype=torch.float32), "labels": torch.as_tensor(labels),
               "ishard": torch.as_tensor(hard), "name": torch.tensor([ord(ch) for ch in stem], dtype=torch.int8)}
        return image, out


class Compose(object):
    """detection/transforms.py Compose: (image, target) through every transform."""

    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target):
        for t in self.transforms:
            image, target = t(image, target)
        return image, target


class ToTensor(object):
    """detection/transforms.py ToTensor (uint8 HWC -> float32 CHW / 255), applied to the image only."""

    def __call__(self, image, target):
        return torch.from_numpy(np.array(image)).permute(2, 0, 1).float().div(255), target


class RandomHorizontalFlip(object):
    """detection/transforms.py RandomHorizontalFlip (detection/transforms.py:27-37; the training transform of get_transform): with probability
    ``prob`` (Python ``random``) the CHW tensor is mirrored and the boxes become (width - xmax, ymin, width - xmin, ymax)."""

    def __init__(self, prob):
        self.prob = prob

    def __call__(self, image, target):
        import random
        if random.random() < self.prob:
            width = image.shape[-1]
            image = image.flip(-1)
            bbox = target["boxes"]
            bbox[:, [0, 2]] = width - bbox[:, [2, 0]]
            target["boxes"] = bbox
        return image, target


def get_transform(train):
    """detection/train.py:54-59 (used at cald_train.py:288-294): ToTensor (+ RandomHorizontalFlip(0.5) for training)."""
    t = [ToTensor()]
    if train:
        t.append(RandomHorizontalFlip(0.5))
    return Compose(t)


class VOCDetection(object):
    """detection/voc_utils.py:47-58 over a VOCdevkit tree: ``root/VOCdevkit/VOC<year>/{ImageSets/Main,Annotations,JPEGImages}``."""

    def __init__(self, img_folder, year, image_set, transforms):
        self.root, self.year, self.image_set = img_folder, str(year), image_set
        self._transforms = transforms
        base = os.path.join(img_folder, "VOCdevkit", "VOC" + self.year)
        with open(os.path.join(base, "ImageSets", "Main", image_set + ".txt")) as f:
            self.ids = [ln.strip().split()[0] for ln in f if ln.strip()]
        self.images = [os.path.join(base, "JPEGImages", i + ".jpg") for i in self.ids]
        self.annotations = [os.path.join(base, "Annotations", i + ".xml") for i in self.ids]

    def __len__(self):
        return len(self.ids)

    def annotation(self, idx):
        return parse_voc_xml(ET.parse(self.annotations[idx]).getroot())["annotation"]

    def __getitem__(self, idx):
        from PIL import Image
        img = Image.open(self.images[idx]).convert("RGB")
        target = dict(image_id=idx, annotations=self.annotation(idx))
        if self._transforms is not None:
            img, target = self._transforms(img, target)
        return img, target

    def target(self, idx):
        """The converted target alone (no image decode): what cls_kldiv reads from the labeled loader and voc_evaluate from
        the test loader.  Image-changing transforms after the conversion (flips) do not apply here."""
        convert = self._transforms.transforms[0] if isinstance(self._transforms, Compose) else ConvertVOCtoCOCO()
        return convert(None, dict(image_id=idx, annotations=self.annotation(idx)))[1]

    def device_pool(self, indices=None):
        """The JPEG files of `indices` (default: all) decoded once on the GPU into an HBM-resident pool (bit-identical to
        Image.open(path).convert('RGB')); ``pool.loader()`` is what get_uncertainty takes."""
        from .pool import DevicePool
        idx = range(len(self)) if indices is None else indices
        return DevicePool.from_files([self.images[int(i)] for i in idx])

    def label_loader(self, indices):
        """[(None, (target,))] over `indices`: the labeled loader of cald_train.py:434-444 as far as cls_kldiv reads it."""
        return [(None, (self.target(int(i)),)) for i in indices]

    def resident_train_loader(self, batch_sampler, indices, pool=None, flip_prob=0.5):
        """ResidentTrainLoader over the dataset indices `indices` (the labeled set) -- see the class."""
        return ResidentTrainLoader(self, batch_sampler, indices, pool, flip_prob)

    def resident_loader(self, indices=None, pool=None):
        """A loader over HBM-resident images WITH their targets (batch size 1, the reference's test loader shape), carrying
        ``.dataset`` so voc_evaluate finds root / image_set / CLASSES."""
        idx = list(range(len(self)) if indices is None else indices)
        return _ResidentLoader(self, idx, pool if pool is not None else self.device_pool(idx))


class ResidentTrainLoader(object):
    """The training loader of cald_train.py:333-336 over an HBM-resident pool: for every batch of ``batch_sampler`` (dataset indices,
    e.g. GroupedBatchSampler over SubsetRandomSampler(labeled_set)) the uint8 HWC device images and their targets -- what
    ``task_model(images, targets)`` takes in train mode -- with get_transform(train=True)'s RandomHorizontalFlip(``flip_prob``) applied
    on the device: one draw of Python's ``random`` per image in batch order (detection/transforms.py:27-37), the image mirrored along
    its width, the boxes mapped to (width - xmax, ymin, width - xmin, ymax).  No file is read or decoded during the epoch."""

    def __init__(self, dataset, batch_sampler, indices, pool=None, flip_prob=0.5):
        self.dataset, self.batch_sampler, self.flip_prob = dataset, batch_sampler, flip_prob
        self.indices = [int(i) for i in indices]
        self.pool = pool if pool is not None else dataset.device_pool(self.indices)
        self._pos = {i: k for k, i in enumerate(self.indices)}

    def __len__(self):
        return len(self.batch_sampler)

    def __iter__(self):
        import random
        for batch in self.batch_sampler:
            images, targets = [], []
            for i in batch:
                img = self.pool[self._pos[int(i)]]
                t = dict(self.dataset.target(int(i)))
                if self.flip_prob and random.random() < self.flip_prob:
                    width = img.shape[1]
                    img = img.flip(1)
                    b = t["boxes"].clone()
                    b[:, [0, 2]] = width - b[:, [2, 0]]
                    t["boxes"] = b
                images.append(img); targets.append(t)
            yield images, targets


def resident_train_loader(dataset, batch_sampler, indices, pool=None, flip_prob=0.5):
    return ResidentTrainLoader(dataset, batch_sampler, indices, pool, flip_prob)


class _ResidentLoader(object):
    def __init__(self, dataset, indices, pool):
        self.dataset, self.indices, self.pool = dataset, indices, pool

    def __len__(self):
        return len(self.indices)

    def __iter__(self):
        for k, i in enumerate(self.indices):
            yield [self.pool[k]], [self.dataset.target(int(i))]


def _get(root, year, image_set, transforms):
    t = [ConvertVOCtoCOCO()]
    if transforms is not None:
        t.append(transforms)
    return VOCDetection(img_folder=root, year=year, image_set=image_set, transforms=Compose(t))


def get_voc2012(root, image_set, transforms):
    """detection/voc_utils.py:61-70."""
    return _get(root, "2012", image_set, transforms)


def get_voc2007(root, image_set, transforms):
    """detection/voc_utils.py:73-82."""
    return _get(root, "2007", image_set, transforms)
