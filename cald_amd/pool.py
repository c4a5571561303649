"""Input side of the sweep: the unlabeled pool kept resident in HBM.

The reference re-reads and re-decodes every pool image on the CPU in every active-learning cycle
(``DataLoader(dataset_aug, batch_size=1, sampler=SubsetSequentialSampler(subset), num_workers=...)``,
cald_train.py:434; ``Image.open(path).convert('RGB')`` in torchvision's VOCDetection.__getitem__ reached from
detection/voc_utils.py:47-58; ``ToTensor`` in detection/train.py:54-59).  A MI355X has 288 GB of HBM: all of
VOC07+12 trainval as uint8 RGB is 9 GB, COCO train2017 about 90 GB.  ``DevicePool`` therefore decodes each JPEG
ONCE, on the GPU (``cald_jpeg_decode_batch``: bit-identical to Pillow), keeps the uint8 HWC images in one HBM arena,
and hands the sweep device pointers; later cycles only change the subset of positions.

``pool.loader(subset)`` yields ``([image], [None])`` batches of one, i.e. it can be passed wherever the reference
passes ``unlabeled_loader`` (``cald_amd.sweep.get_uncertainty`` takes uint8 HWC CUDA tensors as they are).
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi


def jpeg_info(data):
    """(H, W, ncomp) of a JPEG byte string (host-only header parse)."""
    buf = np.frombuffer(data, np.uint8)
    H, W, nc = C.c_int(), C.c_int(), C.c_int()
    _ffi.check(_ffi.lib().cald_jpeg_info(buf.ctypes.data, buf.size, C.byref(H), C.byref(W), C.byref(nc)))
    return H.value, W.value, nc.value


def decode_jpeg_batch(blobs, outs=None, ctx=None):
    """Decodes JPEG byte strings on the GPU.  Returns a list of uint8 [H][W][3] CUDA tensors (RGB), equal to
    ``np.asarray(Image.open(io.BytesIO(b)).convert('RGB'))``.  ``outs`` (optional): preallocated tensors."""
    from .detector import get_ctx
    L = _ffi.lib()
    n = len(blobs)
    if n == 0:
        return []
    bufs = [np.frombuffer(b, np.uint8) for b in blobs]
    if outs is None:
        dev = torch.device("cuda", torch.cuda.current_device())
        outs = []
        for b in blobs:
            H, W, _ = jpeg_info(b)
            outs.append(torch.empty((H, W, 3), dtype=torch.uint8, device=dev))
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    sizes = (C.c_size_t * n)(*[b.size for b in bufs])
    optrs = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    _ffi.check(L.cald_jpeg_decode_batch(ctx if ctx is not None else get_ctx(), n, ptrs, sizes, optrs))
    return outs


class DevicePool:
    """uint8 HWC images resident in one HBM arena, addressed by pool position."""

    def __init__(self, arena, offsets, shapes):
        self.arena = arena            # 1-D uint8 CUDA tensor
        self.offsets = offsets        # int64 [n]
        self.shapes = shapes          # [(H, W)]

    def __len__(self):
        return len(self.shapes)

    def __getitem__(self, i):
        H, W = self.shapes[i]
        o = int(self.offsets[i])
        return self.arena[o:o + H * W * 3].view(H, W, 3)

    @property
    def nbytes(self):
        return int(self.arena.numel())

    @staticmethod
    def _layout(shapes):
        offsets = np.zeros(len(shapes), np.int64)
        off = 0
        for i, (H, W) in enumerate(shapes):
            offsets[i] = off
            off += (H * W * 3 + 255) & ~255            # 256-byte aligned images
        return offsets, off

    @classmethod
    def from_jpeg_bytes(cls, blobs, chunk=512, device=None):
        """Decode-once constructor: header parse on the host, everything else on the GPU, ``chunk`` files per call
        (bounds the coefficient workspace: about 0.9 MB per VOC-sized image)."""
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        shapes = [jpeg_info(b)[:2] for b in blobs]
        offsets, total = cls._layout(shapes)
        arena = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
        pool = cls(arena, offsets, shapes)
        for s in range(0, len(blobs), chunk):
            idx = range(s, min(s + chunk, len(blobs)))
            decode_jpeg_batch([blobs[i] for i in idx], [pool[i] for i in idx])
        return pool

    @classmethod
    def from_files(cls, paths, chunk=512, device=None):
        blobs = []
        for p in paths:
            with open(p, "rb") as f:
                blobs.append(f.read())
        return cls.from_jpeg_bytes(blobs, chunk, device)

    @classmethod
    def from_arrays(cls, arrays, device=None):
        """Already-decoded uint8 HWC arrays (e.g. files in a format other than baseline JPEG, decoded by the caller)."""
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        arrays = [np.array(a, dtype=np.uint8, order="C") for a in arrays]
        shapes = [a.shape[:2] for a in arrays]
        offsets, total = cls._layout(shapes)
        arena = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
        pool = cls(arena, offsets, shapes)
        for i, a in enumerate(arrays):
            pool[i].copy_(torch.from_numpy(a), non_blocking=True)
        return pool

    def loader(self, subset=None, rank=0, world_size=1):
        """Iterable with the reference loader's batch shape: (images: list of 1, targets: list of 1).  With
        world_size > 1 only this rank's strided shard subset[rank::world_size] is yielded (pass
        ``loader_is_sharded=True`` to get_uncertainty)."""
        order = range(len(self)) if subset is None else subset
        for i in list(order)[rank::world_size]:
            yield [self[int(i)]], [None]
