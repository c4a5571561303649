"""Thin wrappers of the cald_train_* device operators (include/cald_hip.h, cald_amd/csrc/train.hip) on torch CUDA tensors.

torch supplies device memory and the stream only; every arithmetic step is a libcaldhip kernel.  Activations are NHWC
float32 [N, H, W, C].  No fallback: without the library or an MI355X every call raises.
"""
import ctypes as C

import torch

from . import _ffi
from .detector import get_ctx

FLAG_BIAS, FLAG_BN, FLAG_RELU = 1, 2, 4


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _chk(t, name):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), name


def round_up(x, m):
    return (x + m - 1) // m * m


def packed_floats(Cout, Cin, KH, KW, CinK, mode):
    n = C.c_int64()
    _ffi.check(_ffi.lib().cald_train_packed_floats(Cout, Cin, KH, KW, CinK, mode, C.byref(n)))
    return int(n.value)


class PackedConv(object):
    """A torch-layout weight [Cout, Cin, KH, KW] (+ bias / frozen-BN scale, shift) in the layout the MFMA kernels read."""

    def __init__(self, weight, bias=None, scale=None, shift=None, CinK=None, mode=0, taps=None, out=None):
        _chk(weight, "weight")
        if mode == 2:                                    # linear on [tap][Cin] rows, torch weight [Cout, Cin * taps]
            self.Cout, self.Cin, self.KH, self.KW = weight.shape[0], weight.shape[1] // taps, taps, 1
            CinK = self.Cin
        elif weight.dim() == 2:                          # plain linear layer = 1x1 conv
            self.Cout, self.Cin, self.KH, self.KW = weight.shape[0], weight.shape[1], 1, 1
        else:
            self.Cout, self.Cin, self.KH, self.KW = weight.shape
        self.mode = mode
        self.CinK = CinK if CinK is not None else (round_up(self.Cout, 4) if mode == 1 else self.Cin)
        n = packed_floats(self.Cout, self.Cin, self.KH, self.KW, self.CinK, mode)
        self.buf = out if out is not None else torch.empty(n, dtype=torch.float32, device=weight.device)
        assert self.buf.numel() >= n
        self.flags = (FLAG_BIAS if bias is not None else 0) | (FLAG_BN if scale is not None else 0)
        _ffi.check(_ffi.lib().cald_train_pack_conv(get_ctx(weight.device.index), _p(weight), _p(bias), _p(scale), _p(shift), self.Cout,
                                                   self.Cin, self.KH, self.KW, self.CinK, mode, _p(self.buf)))


def conv(x, pk, stride=1, pad=0, relu=False, residual=None, up=None, out=None, out_ld=None):
    """Forward conv / linear (pk.mode 0 or 2) or stride-1 data gradient (pk.mode 1) of a dense NHWC batch."""
    _chk(x, "x")
    N, H, W, Cx = x.shape
    n_out = pk.Cin if pk.mode == 1 else pk.Cout
    if pk.mode == 2:
        assert Cx == pk.Cin * pk.KH
        kh = kw = 1
    else:
        assert Cx == pk.CinK, (Cx, pk.CinK)
        kh, kw = pk.KH, pk.KW
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    ld = out_ld if out_ld is not None else n_out
    if out is None:
        out = (torch.zeros if ld != n_out else torch.empty)((N, Ho, Wo, ld), dtype=torch.float32, device=x.device)
    Hup, Wup = (up.shape[1], up.shape[2]) if up is not None else (0, 0)
    flags = pk.flags | (FLAG_RELU if relu else 0)
    _ffi.check(_ffi.lib().cald_train_conv(get_ctx(x.device.index), N, H, W, _p(x), Cx if pk.mode != 2 else pk.Cin, _p(pk.buf), pk.Cout,
                                          pk.Cin, pk.KH, pk.KW, stride, pad, pk.mode, flags, _p(residual), _p(up), Hup, Wup, _p(out), ld))
    return out


def dilate(g, s, Hd, Wd):
    N, Ho, Wo, Cc = g.shape
    out = torch.empty((N, Hd, Wd, Cc), dtype=torch.float32, device=g.device)
    _ffi.check(_ffi.lib().cald_train_dilate(get_ctx(g.device.index), N, Ho, Wo, Cc, s, Hd, Wd, _p(g), _p(out)))
    return out


def conv_dgrad(g, pk_d, H, W, stride, pad, residual=None):
    """dX [N, H, W, Cin] of a conv whose forward had (stride, pad); g [N, Ho, Wo, CinK(pk_d)], pk_d packed with mode 1."""
    K = pk_d.KH
    if stride > 1:
        g = dilate(g, stride, H + 2 * pad - K + 1, W + 2 * pad - pk_d.KW + 1)
    out = conv(g, pk_d, stride=1, pad=K - 1 - pad, residual=residual)
    assert out.shape[1] == H and out.shape[2] == W, (out.shape, H, W)
    return out


def conv_wgrad(x, g, Cin, Cout, KH, KW, stride, pad, dw, db=None, accumulate=False):
    _chk(x, "x"); _chk(g, "g")
    N, H, W, ldx = x.shape
    _ffi.check(_ffi.lib().cald_train_conv_wgrad(get_ctx(x.device.index), N, H, W, _p(x), Cin, ldx, _p(g), Cout, g.shape[-1], KH, KW, stride,
                                                pad, _p(dw), _p(db), int(accumulate)))
    return dw


def linear_wgrad(x, g, Cout, dw, db=None, taps=1, accumulate=False):
    _chk(x, "x"); _chk(g, "g")
    R, K = x.shape[0], x[0].numel()
    _ffi.check(_ffi.lib().cald_train_linear_wgrad(get_ctx(x.device.index), R, _p(x), K, _p(g), Cout, g.shape[-1], taps, _p(dw), _p(db),
                                                  int(accumulate)))
    return dw


def relu_bwd_(g, act=None, scale=None):
    Cc = g.shape[-1]
    _ffi.check(_ffi.lib().cald_train_relu_bwd(get_ctx(g.device.index), g.numel() // Cc, Cc, _p(g), _p(act), _p(scale)))
    return g


def add(a, b=None, out=None):
    out = out if out is not None else torch.empty_like(a)
    _ffi.check(_ffi.lib().cald_train_add(get_ctx(a.device.index), a.numel(), _p(out), _p(a), _p(b)))
    return out


def upsample_bwd_(fine, coarse):
    N, Hf, Wf, Cc = fine.shape
    _ffi.check(_ffi.lib().cald_train_upsample_bwd(get_ctx(fine.device.index), N, Hf, Wf, coarse.shape[1], coarse.shape[2], Cc, _p(fine), _p(coarse)))
    return coarse


def sgd_(param, grad, buf, lr, momentum, weight_decay, first_step):
    _ffi.check(_ffi.lib().cald_train_sgd(get_ctx(param.device.index), param.numel(), _p(param), _p(grad), _p(buf), lr, momentum, weight_decay,
                                         int(first_step)))
