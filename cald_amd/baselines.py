"""Baseline active-learning sweeps of the same repo on the HIP detector.

``lt_c_get_uncertainty(task_model, unlabeled_loader)``  -- lt_c_train.py:105-121 (localization tightness + classification)
``ls_c_get_uncertainty(task_model, unlabeled_loader)``  -- ls_c_train.py:108-155 (localization stability + classification;
    six GaussianNoise views per image, torch.randn stream re-seeded per pool position like the CALD sweep)
Same positional signatures and return types (list of floats in loader order) as the reference functions.
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi
from .sweep import _to_u8_cuda


def _collect(unlabeled_loader):
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    images, positions = [], []
    for pos, (imgs, _) in enumerate(unlabeled_loader):
        for image in imgs:
            images.append(_to_u8_cuda(image, dev)); positions.append(pos)
    return images, positions


def _arrays(images):
    n = len(images)
    ptrs = (C.c_void_p * n)(*[im.data_ptr() for im in images])
    Hs = np.array([im.shape[0] for im in images], np.int32)
    Ws = np.array([im.shape[1] for im in images], np.int32)
    return ptrs, Hs, Ws


def lt_c_get_uncertainty(task_model, unlabeled_loader, batch_images=64):
    task_model.eval()
    images, _ = _collect(unlabeled_loader)
    out = np.zeros(len(images), np.float64)
    if len(images):
        ptrs, Hs, Ws = _arrays(images)
        _ffi.check(_ffi.lib().cald_sweep_ltc(task_model.handle(), len(images), ptrs, _ffi.ptr(Hs, _ffi.c_i), _ffi.ptr(Ws, _ffi.c_i),
                                             batch_images, _ffi.ptr(out, _ffi.c_d)))
    return [float(v) for v in out]


def ls_c_get_uncertainty(task_model, unlabeled_loader, aves=None, base_seed=0, batch_images=32):
    task_model.eval()
    images, positions = _collect(unlabeled_loader)
    out = np.zeros(len(images), np.float64)
    if len(images):
        ptrs, Hs, Ws = _arrays(images)
        pos = np.ascontiguousarray(positions, dtype=np.int64)
        _ffi.check(_ffi.lib().cald_sweep_lsc(task_model.handle(), len(images), ptrs, _ffi.ptr(Hs, _ffi.c_i), _ffi.ptr(Ws, _ffi.c_i),
                                             _ffi.ptr(pos, _ffi.c_i64), int(base_seed), batch_images, _ffi.ptr(out, _ffi.c_d)))
    return [float(v) for v in out]
