"""Aspect-ratio batch sampler of the training loop (host side of SURVEY 8f rank 4).

Mirrors detection/group_by_aspect_ratio.py as cald_train.py uses it (:326-332, default ``--aspect-ratio-group-factor 3``):
``create_aspect_ratio_groups(dataset, k)`` quantises width / height into 2k + 2 bins with edges 2 ** linspace(-1, 1, 2k + 1)
(:187-195), ``GroupedBatchSampler`` (:23-88) cuts the base sampler's order into batches of one group each.  Batches of one
group are what makes the padded batch of the training step (``GeneralizedRCNNTransform.batch_images``: the largest height
and the largest width of the batch) tight: with VOC's landscape and portrait images mixed a batch pads to 800 x 800, a
one-group batch to about 608 x 800 -- a quarter fewer pixels through every layer.

Same results as the reference on the same sampler order and sizes (tests/golden/group_sampler.npz, made by
oracle/make_golden_group_sampler.py from the imported reference), including the order in which the incomplete groups are
topped up at the end of an epoch.
"""
import bisect
import math

import numpy as np
import torch
from torch.utils.data.sampler import BatchSampler, Sampler


class GroupedBatchSampler(BatchSampler):
    """Batches of ``batch_size`` indices of ONE group each, in an order as close to the base sampler's as that allows; the number of
    batches is ``len(sampler) // batch_size`` whatever the group sizes (detection/group_by_aspect_ratio.py:23-88)."""

    def __init__(self, sampler, group_ids, batch_size):
        if not isinstance(sampler, Sampler):
            raise ValueError("sampler should be an instance of torch.utils.data.Sampler, but got sampler={}".format(sampler))
        self.sampler, self.group_ids, self.batch_size = sampler, group_ids, batch_size

    def __iter__(self):
        waiting = {}        # group -> indices waiting for their batch to fill; a group that has just emitted moves to the END of this
        drawn = {}          # dict with an empty list (the reference's defaultdict re-creates the entry): that order breaks the ties below
        emitted = 0
        for idx in self.sampler:
            g = self.group_ids[idx]
            waiting.setdefault(g, []).append(idx)
            drawn.setdefault(g, []).append(idx)
            if len(waiting[g]) == self.batch_size:
                yield waiting.pop(g)
                emitted += 1
                waiting[g] = []
        # the epoch's last batches: incomplete groups, fullest first (ties in the order above), topped up by cycling through the
        # group's own indices of this epoch, until the sampler's length is reached
        missing = len(self) - emitted
        if missing > 0:
            for g, buf in sorted(waiting.items(), key=lambda kv: len(kv[1]), reverse=True):
                need = self.batch_size - len(buf)
                own = drawn[g]
                buf.extend((own * int(math.ceil(need / float(len(own)))))[:need])
                yield buf
                missing -= 1
                if missing == 0:
                    break
        assert missing <= 0

    def __len__(self):
        return len(self.sampler) // self.batch_size


def _image_size(path):
    """(width, height) from the file header only."""
    from PIL import Image
    with Image.open(path) as im:            # lazy: the pixel data is not decoded
        return im.size


def compute_aspect_ratios(dataset, indices=None):
    """width / height per dataset item without decoding images where the dataset allows it (:163-178): a ``get_height_and_width``
    method, the COCO image table, VOC's JPEG headers, a Subset of any of those; otherwise every item is loaded."""
    if indices is None:
        indices = range(len(dataset))
    if hasattr(dataset, "get_height_and_width"):
        out = []
        for i in indices:
            h, w = dataset.get_height_and_width(i)
            out.append(float(w) / float(h))
        return out
    from . import coco_utils, voc_utils
    if isinstance(dataset, coco_utils.CocoDetection):
        return [float(dataset.imgs[dataset.ids[i]]["width"]) / float(dataset.imgs[dataset.ids[i]]["height"]) for i in indices]
    if isinstance(dataset, voc_utils.VOCDetection):
        out = []
        for i in indices:
            w, h = _image_size(dataset.images[i])
            out.append(float(w) / float(h))
        return out
    if isinstance(dataset, (coco_utils.Subset, torch.utils.data.Subset)):
        return compute_aspect_ratios(dataset.dataset, [dataset.indices[i] for i in indices])
    out = []
    for i in indices:                        # slow path: load every image
        img = dataset[i][0]
        h, w = img.shape[-2:]
        out.append(float(w) / float(h))
    return out


def _quantize(x, bins):
    edges = sorted(bins)
    return [bisect.bisect_right(edges, y) for y in x]


def create_aspect_ratio_groups(dataset, k=0):
    """Group id per item: the bin of its width / height among the edges 2 ** linspace(-1, 1, 2k + 1) (k = 0: the single edge 1.0)."""
    ratios = compute_aspect_ratios(dataset)
    bins = (2 ** np.linspace(-1, 1, 2 * k + 1)).tolist() if k > 0 else [1.0]
    return _quantize(ratios, bins)
