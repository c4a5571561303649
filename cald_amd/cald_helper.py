"""Mirror of the reference helper API (cald/cald_helper.py) so `from cald.cald_helper import *` users keep
working: same names and signatures; the box transforms run on the caller's tensors, the pixel work is what
libcaldhip's preprocess / PIL-resize kernels do inside the sweep.

HorizontalFlip(image, bbox) -> (image, boxes)      cald_helper.py:23-30
resize(img, boxes, ratio)   -> (image, boxes)      cald_helper.py:47-53   (Pillow BILINEAR on uint8, on the GPU)
cutout(image, boxes, labels, cut_num=2, ...) -> image   cald_helper.py:88-132 (rectangles from Python `random`)
intersect(boxes1, boxes2)   -> [n1, n2]            cald_helper.py:226-243
"""
import ctypes as C
import random

import numpy as np
import torch

from . import _ffi


def _to_u8_hwc(image):
    if isinstance(image, torch.Tensor):
        t = image
        if t.dtype != torch.uint8:
            t = (t * 255.0).round().clamp(0, 255).to(torch.uint8)
        if t.shape[0] == 3 and t.shape[-1] != 3:
            t = t.permute(1, 2, 0)
        return t.contiguous()
    return torch.from_numpy(np.array(image, dtype=np.uint8))


def _to_float_chw(u8):
    return u8.permute(2, 0, 1).float().div(255)


def HorizontalFlip(image, bbox):
    u8 = _to_u8_hwc(image)
    width = u8.shape[1]
    b = bbox.clone()
    b[:, [0, 2]] = width - bbox[:, [2, 0]]
    return _to_float_chw(u8.flip(1)), b


def resize(img, boxes, ratio):
    from .detector import get_ctx
    u8 = _to_u8_hwc(img).cuda()
    H, W = u8.shape[:2]
    ow, oh = int(W * ratio), int(H * ratio)
    dst = torch.empty((oh, ow, 3), dtype=torch.uint8, device=u8.device)
    _ffi.check(_ffi.lib().cald_op_pil_resize(get_ctx(u8.device.index), u8.data_ptr(), H, W, dst.data_ptr(), oh, ow))
    return _to_float_chw(dst), boxes * ratio


def cutout(image, boxes, labels, cut_num=2, fill_val=0, bbox_remove_thres=0.4, bbox_min_thres=0.1, seed=None):
    """`seed`: the per-image seed used by the sweep; when None one is drawn from Python's global `random`."""
    if fill_val != 0 or bbox_remove_thres != 0.4 or bbox_min_thres != 0.1:
        raise NotImplementedError("only the reference's default fill/thresholds are implemented")
    u8 = _to_u8_hwc(image).clone()
    H, W = u8.shape[:2]
    if seed is None:
        seed = random.getrandbits(63)
    b = np.ascontiguousarray(boxes.detach().cpu().numpy(), dtype=np.float32).reshape(-1, 4)
    rects = np.zeros(16, np.int32)
    n = C.c_int()
    _ffi.check(_ffi.lib().cald_op_cutout_rects(int(seed), H, W, b.shape[0], _ffi.ptr(b), cut_num, _ffi.ptr(rects, _ffi.c_i), C.byref(n)))
    for (l, t, r, bt) in rects[:4 * n.value].reshape(-1, 4):
        u8[t:bt, l:r] = 0
    return _to_float_chw(u8)


def intersect(boxes1, boxes2):
    n1, n2 = boxes1.size(0), boxes2.size(0)
    max_xy = torch.min(boxes1[:, 2:].unsqueeze(1).expand(n1, n2, 2), boxes2[:, 2:].unsqueeze(0).expand(n1, n2, 2))
    min_xy = torch.max(boxes1[:, :2].unsqueeze(1).expand(n1, n2, 2), boxes2[:, :2].unsqueeze(0).expand(n1, n2, 2))
    inter = torch.clamp(max_xy - min_xy, min=0)
    return inter[:, :, 0] * inter[:, :, 1]
