"""Aspect-ratio sampler fixtures from the IMPORTED reference (TEST INFRASTRUCTURE ONLY; build container only).

    python oracle/make_golden_group_sampler.py   ->  tests/golden/group_sampler.npz

detection/group_by_aspect_ratio.py is executed as it lies in /root/reference: ``create_aspect_ratio_groups`` (k = 0, 1, 3) on
datasets that expose ``get_height_and_width`` (VOC- and COCO-like size mixes, plus ratios sitting exactly on bin edges), and
``GroupedBatchSampler`` over fixed index orders with batch sizes 2, 4 and 7 -- group populations that do and do not divide by
the batch size, groups that never fill a batch, ties between incomplete groups.  Stored: the sizes, the sampler orders, the
group ids and every batch the reference yielded.
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


class _Sizes(object):
    def __init__(self, hw):
        self.hw = hw

    def __len__(self):
        return len(self.hw)

    def get_height_and_width(self, i):
        return int(self.hw[i][0]), int(self.hw[i][1])


def main():
    import torch
    from torch.utils.data.sampler import Sampler
    ref_harness.load_reference()
    G = importlib.import_module("detection.group_by_aspect_ratio")

    class _Order(Sampler):
        def __init__(self, order):
            self.order = list(order)

        def __iter__(self):
            return iter(self.order)

        def __len__(self):
            return len(self.order)

    rs = np.random.RandomState(17)
    voc = [(375, 500), (500, 375), (333, 500), (500, 333), (500, 500), (281, 500), (500, 400), (374, 500), (250, 500), (500, 250), (200, 500)]
    coco = [(480, 640), (640, 480), (427, 640), (640, 427), (640, 640), (500, 375), (360, 640), (612, 612), (333, 500), (640, 318)]
    cases = {"voc": np.array([voc[i] for i in rs.randint(0, len(voc), 203)]),
             "coco": np.array([coco[i] for i in rs.randint(0, len(coco), 97)]),
             "edges": np.array([(100, 50), (100, 63), (100, 100), (100, 200), (200, 100), (1000, 1260), (400, 200), (300, 300), (100, 126), (126, 100)]),
             "one_group": np.array([(375, 500)] * 10)}
    blob = {}
    for name, hw in cases.items():
        blob["hw_" + name] = hw
        for k in (0, 1, 3):
            groups = G.create_aspect_ratio_groups(_Sizes(hw), k=k)
            blob["groups_%s_k%d" % (name, k)] = np.array(groups, np.int64)
            for bs in (2, 4, 7):
                for oi, order in enumerate((np.arange(len(hw)), rs.permutation(len(hw)))):
                    if name == "edges" and len(hw) < bs:
                        continue
                    sampler = G.GroupedBatchSampler(_Order(order.tolist()), groups, bs)
                    batches = [list(b) for b in sampler]
                    assert len(batches) == len(sampler)
                    key = "%s_k%d_b%d_o%d" % (name, k, bs, oi)
                    blob["order_" + key] = np.array(order, np.int64)
                    blob["batches_" + key] = np.array(batches, np.int64).reshape(len(batches), bs)
    np.savez_compressed(os.path.join(OUT, "group_sampler.npz"), **blob)
    print("group_sampler.npz:", len(blob), "arrays")


if __name__ == "__main__":
    main()
