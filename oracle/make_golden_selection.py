"""Extra selection-stage fixtures from the IMPORTED reference (TEST INFRASTRUCTURE ONLY; build container only).

    python oracle/make_golden_selection.py   ->  tests/golden/selection_more.npz

cald_train.cls_kldiv (cald_train.py:234-271) is executed as it lies in /root/reference on randomised candidate sets that
cover what tests/golden/selection.npz does not: exact ties in the JS vector, all-zero candidate rows beyond the budget,
`budget` larger than the candidate list (the reference then appends index 0 repeatedly), 20- and 90-wide class vectors,
both settings of --uniform.  Only inputs and the selected indices are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    ct = ref_harness.load_reference()
    ct = ct[0] if isinstance(ct, tuple) else ct
    rs = np.random.RandomState(7)
    blob = {}
    ncase = 24
    for case in range(ncase):
        n = int(rs.randint(3, 60)); Cm1 = int(rs.choice([20, 90]))
        budget = int(rs.randint(1, n + 5 if case % 6 == 0 else n))
        cls = rs.rand(n, Cm1) * (rs.rand(n, Cm1) < 0.3)
        if case % 3 == 0:
            cls = np.round(cls, 1)                      # exact ties
        for _ in range(int(rs.randint(0, 4))):
            cls[rs.randint(n)] = 0
        if case % 11 == 10:
            cls[:] = 0                                  # more zero-sum candidates than the budget
        labels = [rs.randint(1, Cm1 + 1, rs.randint(1, 6)) for _ in range(int(rs.randint(1, 9)))]
        loader = [(None, [{"labels": torch.from_numpy(l)}]) for l in labels]
        uniform = bool(case % 2)
        ct.args.uniform = uniform
        sel = ct.cls_kldiv(loader, list(cls), budget, 0)
        blob["cls_corrs%d" % case] = cls
        blob["labels%d" % case] = np.array([np.pad(l, (0, 8 - len(l)), constant_values=-1) for l in labels])
        blob["budget%d" % case] = budget
        blob["uniform%d" % case] = uniform
        blob["sel%d" % case] = np.array([int(s) for s in sel], np.int64)
    ct.args.uniform = False
    blob["n_cases"] = ncase
    np.savez_compressed(os.path.join(OUT, "selection_more.npz"), **blob)
    print("selection_more ok:", ncase, "cases")


if __name__ == "__main__":
    main()
