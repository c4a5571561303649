"""Golden vectors for the JPEG decode row: TEST INFRASTRUCTURE.

The reference decodes with ``PIL.Image.open(path).convert('RGB')`` (torchvision VOCDetection.__getitem__).  This
script encodes small seeded images with Pillow in the flavours the decoder supports and stores (file bytes,
Pillow's decoded RGB) pairs in tests/golden/jpeg_cases.npz.  Run in the build container:
    python oracle/make_golden_jpeg.py
"""
import io
import os
import sys

import numpy as np
from PIL import Image, ImageFile, __version__ as pil_version, features

ImageFile.MAXBLOCK = 1 << 24
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cald_amd import synth  # noqa: E402


def cases():
    rng = np.random.default_rng(20260928)
    specs = [
        # (H, W, content, save kwargs, grayscale)
        (48, 64, "synth", dict(quality=90, subsampling=2), False),
        (37, 53, "synth", dict(quality=75, subsampling=2, optimize=True), False),
        (40, 40, "noise", dict(quality=95, subsampling=1), False),
        (33, 31, "synth", dict(quality=60, subsampling=0), False),
        (17, 23, "noise", dict(quality=100, subsampling=2), False),
        (24, 56, "synth", dict(quality=85), True),
        (1, 1, "noise", dict(quality=90, subsampling=2), False),
        (2, 3, "noise", dict(quality=90, subsampling=2), False),      # chroma width <= 2: replication branch
        (5, 40, "synth", dict(quality=80, subsampling=1), False),
        (50, 70, "synth", dict(quality=85, subsampling=2, restart_marker_blocks=3), False),
        (32, 48, "noise", dict(quality=30, subsampling=2, restart_marker_rows=1, optimize=True), False),
        (64, 64, "smooth", dict(quality=98, subsampling=2), False),
    ]
    out = []
    for k, (H, W, kind, kw, gray) in enumerate(specs):
        if kind == "noise":
            a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        elif kind == "smooth":
            y, x = np.mgrid[0:H, 0:W]
            a = np.stack([x * 255 // max(W - 1, 1), y * 255 // max(H - 1, 1), (x + y) * 255 // max(H + W - 2, 1)], -1).astype(np.uint8)
        else:
            a = np.ascontiguousarray(synth.synth_image(900 + k, max(H, 33), max(W, 33))[:H, :W])
        im = Image.fromarray(a)
        if gray:
            im = im.convert("L")
        bio = io.BytesIO()
        im.save(bio, "JPEG", **kw)
        data = bio.getvalue()
        ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        out.append((data, ref))
    return out


if __name__ == "__main__":
    d = {"pillow_version": np.array(pil_version), "libjpeg": np.array(str(features.version("jpg"))),
         "libjpeg_turbo": np.array(bool(features.check_feature("libjpeg_turbo")))}
    cs = cases()
    d["n"] = np.array(len(cs))
    for i, (data, ref) in enumerate(cs):
        d["file_%d" % i] = np.frombuffer(data, np.uint8)
        d["rgb_%d" % i] = ref
    path = os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path), "bytes,", len(cs), "cases")
