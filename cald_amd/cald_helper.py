"""Mirror of the reference helper API (cald/cald_helper.py) so `from cald.cald_helper import *` users keep
working: same names and signatures; the box transforms run on the caller's tensors, the pixel work is what
libcaldhip's preprocess / PIL-resize kernels do inside the sweep.

HorizontalFlip(image, bbox) -> (image, boxes)      cald_helper.py:23-30
resize(img, boxes, ratio)   -> (image, boxes)      cald_helper.py:47-53   (Pillow BILINEAR on uint8, on the GPU)
ColorSwap(image)            -> image               cald_helper.py:56-62   (permutation from Python `random`)
ColorAdjust(image, factor)  -> image               cald_helper.py:65-69   (PIL ImageEnhance x3, on the GPU)
GaussianNoise(image, std=1) -> image               cald_helper.py:72-75   (torch.randn stream, on the GPU)
SaltPepperNoise(image, prob)-> image               cald_helper.py:78-85   (torch.rand stream, on the GPU)
cutout(image, boxes, labels, cut_num=2, ...) -> image   cald_helper.py:88-132 (rectangles from Python `random`)
rotate(image, boxes, angle) -> (image, boxes)      cald_helper.py:135-223 (PIL rotate + bicubic resize, on the GPU)
intersect(boxes1, boxes2)   -> [n1, n2]            cald_helper.py:226-243

The random helpers take an optional ``seed``: the generator is re-seeded with it (what the sweep does per image).
With ``seed=None`` one is drawn from the global generator the reference would have used (torch's for the noise
helpers, Python's ``random`` for ColorSwap / cutout).
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


def _augment(kind, param, seed, u8, dst, boxes=None):
    from .detector import get_ctx
    H, W = u8.shape[:2]
    b = None if boxes is None else np.ascontiguousarray(boxes.detach().cpu().numpy(), dtype=np.float32).reshape(-1, 4)
    bo = None if b is None else np.empty_like(b)
    aux = np.zeros(4, np.int32)
    _ffi.check(_ffi.lib().cald_op_augment(get_ctx(u8.device.index if u8.is_cuda else None), kind, float(param), int(seed),
                                          u8.data_ptr(), H, W, 0 if b is None else b.shape[0], _ffi.ptr(b),
                                          None if dst is None else dst.data_ptr(), _ffi.ptr(bo), _ffi.ptr(aux, _ffi.c_i)))
    return bo, aux


def _torch_seed(seed):
    return int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)


COLOR_PERMS = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))


def ColorSwap(image, seed=None):
    u8 = _to_u8_hwc(image)
    if seed is None:
        k = random.randint(0, len(COLOR_PERMS) - 1)
    else:
        _, aux = _augment(_ffi.AUG_COLOR_SWAP, 0.0, seed, u8, None)
        k = int(aux[0])
    return _to_float_chw(u8)[list(COLOR_PERMS[k]), :, :]


def ColorAdjust(image, factor):
    u8 = _to_u8_hwc(image).cuda()
    dst = torch.empty_like(u8)
    _augment(_ffi.AUG_COLOR_ADJUST, factor, 0, u8, dst)
    return _to_float_chw(dst)


def GaussianNoise(image, std=1, seed=None):
    u8 = _to_u8_hwc(image).cuda()
    noise = torch.empty((3, u8.shape[0], u8.shape[1]), dtype=torch.float32, device=u8.device)
    _augment(_ffi.AUG_GAUSS, std, _torch_seed(seed), u8, noise)
    return _to_float_chw(u8) + noise


def SaltPepperNoise(image, prob, seed=None):
    u8 = _to_u8_hwc(image).cuda()
    dst = torch.empty_like(u8)
    _augment(_ffi.AUG_SALT_PEPPER, prob, _torch_seed(seed), u8, dst)
    return _to_float_chw(dst)


def rotate(image, boxes, angle):
    u8 = _to_u8_hwc(image).cuda()
    dst = torch.empty_like(u8)
    bo, _ = _augment(_ffi.AUG_ROTATE, angle, 0, u8, dst, boxes)
    return _to_float_chw(dst), torch.from_numpy(bo).to(boxes.device)


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
    """Pairwise intersection areas [n1, n2] of two (x1, y1, x2, y2) box sets (cald_helper.py:226-243)."""
    a, b = boxes1[:, None, :], boxes2[None, :, :]
    w = (torch.minimum(a[..., 2], b[..., 2]) - torch.maximum(a[..., 0], b[..., 0])).clamp(min=0)
    h = (torch.minimum(a[..., 3], b[..., 3]) - torch.maximum(a[..., 1], b[..., 1])).clamp(min=0)
    return w * h
