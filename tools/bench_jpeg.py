#!/usr/bin/env python
"""Informational: GPU JPEG decode throughput vs Pillow on this host.
Encodes N VOC-shaped synthetic images with Pillow (quality 90, 4:2:0), then times
  * cald_jpeg_decode_batch into a DevicePool (files already in host memory), and
  * PIL.Image.open(...).convert('RGB') on one core.
Prints one JSON line."""
import io
import json
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cald_amd import pool, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
imgs = synth.make_pool(min(n, 64), "voc", 0)
blobs = []
for i in range(n):
    bio = io.BytesIO()
    Image.fromarray(imgs[i % len(imgs)]).save(bio, "JPEG", quality=90, subsampling=2)
    blobs.append(bio.getvalue())
pool.DevicePool.from_jpeg_bytes(blobs[:8])            # warm up (module load, first launches)
torch.cuda.synchronize()
res = {}
for chunk in (64, 512):
    t0 = time.time()
    dp = pool.DevicePool.from_jpeg_bytes(blobs, chunk=chunk)
    torch.cuda.synchronize()
    res["gpu_images_per_s_chunk%d" % chunk] = n / (time.time() - t0)
t0 = time.time()
m = min(n, 200)
for b in blobs[:m]:
    np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))
res["pillow_images_per_s_1core"] = m / (time.time() - t0)
ok = all(np.array_equal(dp[i].cpu().numpy(), np.asarray(Image.open(io.BytesIO(blobs[i])).convert("RGB"))) for i in range(0, n, max(1, n // 16)))
res.update(n=n, file_MB=sum(len(b) for b in blobs) / 1e6, decoded_MB=dp.nbytes / 1e6, bit_identical_to_pillow=bool(ok))
print(json.dumps(res))
