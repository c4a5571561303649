"""make_golden.py -- tests/golden/mfma_f16_hw.npz: a sample of the hardware's own results for v_mfma_f32_32x32x16_f16.

Run after the probe (tools/mfma_model/probe.hip on an MI355X, see gen_cases.py / gen_cases2.py):
    python tools/mfma_model/make_golden.py SET1_DIR OUT1_DIR SET2_DIR OUT2_DIR
SETn_DIR holds cases.bin (the generator's output, regenerated locally: same seed, same bytes -- the md5 is checked on both sides),
OUTn_DIR holds out.bin / index.json as they came back from the GPU box.  1 000 cases per family are kept (every 1 + n // 1000-th), so the
CPU-only suite can hold oracle/mfma_f16_model.h against the instruction without a GPU.
"""
import json
import sys

import numpy as np


def main():
    A, B, C, D, fam = [], [], [], [], []
    for setdir, outdir in ((sys.argv[1], sys.argv[2]), (sys.argv[3], sys.argv[4])):
        raw = np.fromfile(setdir + "/cases.bin", np.uint8)
        n = int(raw[:8].view(np.int64)[0])
        a = raw[8:8 + n * 32].view(np.uint16).reshape(n, 16); b = raw[8 + n * 32:8 + n * 64].view(np.uint16).reshape(n, 16)
        c = raw[8 + n * 64:8 + n * 68].view(np.uint32); d = np.fromfile(outdir + "/out.bin", np.uint32)
        assert d.size == n
        for name, e in json.load(open(outdir + "/index.json")).items():
            sel = np.arange(e["start"], e["start"] + e["n"], 1 + e["n"] // 1000)
            A.append(a[sel]); B.append(b[sel]); C.append(c[sel]); D.append(d[sel]); fam += [name] * sel.size
    np.savez_compressed("tests/golden/mfma_f16_hw.npz", A=np.concatenate(A), B=np.concatenate(B), C=np.concatenate(C), D=np.concatenate(D),
                        family=np.array(fam))
    print("tests/golden/mfma_f16_hw.npz:", len(fam), "cases of", len(set(fam)), "families")


if __name__ == "__main__":
    main()
