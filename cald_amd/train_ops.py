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


_WGRAD_CTX = [None]     # set by train.py while weight gradients / weight packing / target matching are issued on another stream


def _wctx(t):
    return _WGRAD_CTX[0] if _WGRAD_CTX[0] is not None else get_ctx(t.device.index)


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

    def __init__(self, weight, bias=None, scale=None, shift=None, CinK=None, mode=0, taps=None, out=None, pack=True):
        _chk(weight, "weight")
        if mode in (2, 3):                               # linear on [tap][Cin] rows, torch weight [Cout, Cin * taps] (3: its data gradient)
            self.Cout, self.Cin, self.KH, self.KW = weight.shape[0], weight.shape[1] // taps, taps, 1
            CinK = self.Cin if mode == 2 else (CinK if CinK is not None else round_up(self.Cout, 4))
        elif weight.dim() == 2:                          # plain linear layer = 1x1 conv
            self.Cout, self.Cin, self.KH, self.KW = weight.shape[0], weight.shape[1], 1, 1
        else:
            self.Cout, self.Cin, self.KH, self.KW = weight.shape
        self.mode = mode
        self.CinK = CinK if CinK is not None else (round_up(self.Cout, 4) if mode == 1 else self.Cin)
        n = packed_floats(self.Cout, self.Cin, self.KH, self.KW, self.CinK, mode)
        self.buf = out if out is not None else torch.empty(n, dtype=torch.float32, device=weight.device)
        assert self.buf.numel() >= n
        # gradient packs (modes 1, 3) carry no epilogue vectors: a scale given there is folded into the filter rows
        self.flags = ((FLAG_BIAS if bias is not None else 0) | (FLAG_BN if scale is not None else 0)) if mode in (0, 2) else 0
        # what a PackPlan needs to repeat this pack (the tensors are kept alive: the plan holds their device pointers)
        self.src = (weight, bias, scale, shift)
        if pack:
            _ffi.check(_ffi.lib().cald_train_pack_conv(_wctx(weight), _p(weight), _p(bias), _p(scale), _p(shift), self.Cout,
                                                       self.Cin, self.KH, self.KW, self.CinK, mode, _p(self.buf)))

    def job(self):
        w, b, sc, sh = self.src
        q = _ffi.PackJob()
        q.weight, q.bias, q.bn_scale, q.bn_shift = [t.data_ptr() if t is not None else None for t in (w, b, sc, sh)]
        q.Cout, q.Cin, q.KH, q.KW, q.CinK, q.mode, q.packed = self.Cout, self.Cin, self.KH, self.KW, self.CinK, self.mode, self.buf.data_ptr()
        return q


class PackPlan(object):
    """All of a model's PackedConv buffers re-packed in two launches (cald_train_pack_plan_*): after an optimizer step every trainable
    weight has changed, and one cald_train_pack_conv per layer and form is ~220 launches of a few microseconds each."""

    def __init__(self, packs, device):
        self.packs = list(packs)                         # keeps weights and packed buffers alive
        n = len(self.packs)
        self.jobs = (_ffi.PackJob * n)(*[pk.job() for pk in self.packs])
        need = C.c_int64()
        _ffi.check(_ffi.lib().cald_train_pack_plan_scratch_floats(n, self.jobs, C.byref(need)))
        self.scratch = torch.empty(max(int(need.value), 1), dtype=torch.float32, device=device)
        self.handle = C.c_void_p()
        # (the context whose stream run() will use: the scratch is zeroed on that stream)
        _ffi.check(_ffi.lib().cald_train_pack_plan_create(_wctx(self.scratch), n, self.jobs, _p(self.scratch), int(need.value),
                                                          C.byref(self.handle)))

    def run(self):
        _ffi.check(_ffi.lib().cald_train_pack_plan_run(_wctx(self.scratch), self.handle))

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _ffi.lib().cald_train_pack_plan_destroy(h)
            except Exception:
                pass


def conv(x, pk, stride=1, pad=0, relu=False, residual=None, up=None, out=None, out_ld=None, mask=None):
    """Forward conv / linear (pk.mode 0 or 2) or stride-1 data gradient (pk.mode 1) of a dense NHWC batch."""
    _chk(x, "x")
    N, H, W, Cx = x.shape
    n_out = pk.Cin if pk.mode == 1 else (pk.Cin * pk.KH if pk.mode == 3 else pk.Cout)
    if pk.mode == 2:
        assert Cx == pk.Cin * pk.KH
        kh = kw = 1
    elif pk.mode == 3:
        assert Cx == pk.CinK
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
    _ffi.check(_ffi.lib().cald_train_conv(_wctx(x), N, H, W, _p(x), pk.CinK, _p(pk.buf), pk.Cout,
                                          pk.Cin, pk.KH, pk.KW, stride, pad, pk.mode, flags, _p(residual), _p(up), Hup, Wup, _p(mask), _p(out), ld))
    return out


def conv_group(xs, pks, stride=1, pad=0, relu=False, outs=None, out_ld=None, masks=None):
    """The same layer shape on several NHWC tensors in one launch; pks: one PackedConv per tensor (all of one shape) or a single one."""
    if not isinstance(pks, (list, tuple)):
        pks = [pks] * len(xs)
    pk = pks[0]
    n_out = pk.Cin if pk.mode == 1 else pk.Cout
    ld = out_ld if out_ld is not None else n_out
    res = []
    for i, x in enumerate(xs):
        _chk(x, "x")
        assert x.shape[3] == pk.CinK and x.shape[0] == xs[0].shape[0]
        Ho, Wo = (x.shape[1] + 2 * pad - pk.KH) // stride + 1, (x.shape[2] + 2 * pad - pk.KW) // stride + 1
        o = outs[i] if outs is not None else (torch.zeros if ld != n_out else torch.empty)((x.shape[0], Ho, Wo, ld), dtype=torch.float32, device=x.device)
        res.append(o)
    hw = [v for x in xs for v in (x.shape[1], x.shape[2])]
    flags = pk.flags | (FLAG_RELU if relu else 0)
    _ffi.check(_ffi.lib().cald_train_conv_group(_wctx(xs[0]), len(xs), xs[0].shape[0], _int_array(hw), _ptr_array(xs), pk.CinK,
                                                _ptr_array([p.buf for p in pks]), pk.Cout, pk.Cin, pk.KH, pk.KW, stride, pad, pk.mode, flags,
                                                _ptr_array(masks) if masks is not None else None, _ptr_array(res), ld))
    return res


def dilate(g, s, Hd, Wd):
    N, Ho, Wo, Cc = g.shape
    out = torch.empty((N, Hd, Wd, Cc), dtype=torch.float32, device=g.device)
    _ffi.check(_ffi.lib().cald_train_dilate(_wctx(g), N, Ho, Wo, Cc, s, Hd, Wd, _p(g), _p(out)))
    return out


def conv_dgrad(g, pk_d, H, W, stride, pad, residual=None, mask=None):
    """dX [N, H, W, Cin] of a conv whose forward had (stride, pad); g [N, Ho, Wo, CinK(pk_d)], pk_d packed with mode 1."""
    K = pk_d.KH
    if stride > 1 and K == 1 and pk_d.KW == 1 and pad == 0 and residual is None and mask is None:
        # strided 1x1 (the bottleneck's downsample path): only every stride-th input pixel receives a gradient -- multiply on the
        # coarse grid and scatter the result, instead of scattering dY first and multiplying 3/4 zeros
        return dilate(conv(g, pk_d), stride, H, W)
    if stride > 1:
        g = dilate(g, stride, H + 2 * pad - K + 1, W + 2 * pad - pk_d.KW + 1)
    out = conv(g, pk_d, stride=1, pad=K - 1 - pad, residual=residual, mask=mask)
    assert out.shape[1] == H and out.shape[2] == W, (out.shape, H, W)
    return out


_S2_ROWS = ([1], [0, 2])       # filter taps feeding even / odd input positions of a 3-tap, stride-2, pad-1 layer


def pack_s2_grads(weight, scale=None, CinK=None, outs=None):
    """The data gradient of a 3x3 stride-2 pad-1 conv as FOUR small convolutions on the un-dilated dY: input pixels (2m + a, 2n + b)
    receive contributions from disjoint tap sets (1 tap where the coordinate is even, 2 where it is odd), so each phase gets its own
    (1 + a) x (1 + b) sub-filter -- 9 taps per dY pixel in all instead of 36 on the zero-stuffed grid."""
    packs = []
    for a in (0, 1):
        for b in (0, 1):
            sub = weight[:, :, _S2_ROWS[a]][:, :, :, _S2_ROWS[b]].contiguous()
            packs.append(PackedConv(sub, scale=scale, CinK=CinK, mode=1, out=None if outs is None else outs[2 * a + b].buf))
    return packs


def conv_dgrad_s2(g, packs, H, W, mask=None):
    """dX [N, H, W, Cin] of a 3x3 stride-2 pad-1 conv from dY g [N, Ho, Wo, CinK]; packs from pack_s2_grads."""
    N, Ho, Wo, _ = g.shape
    phases, hw, off = [], [], []
    for k, pk in enumerate(packs):
        p = 1 if k else 0
        o = conv(g, pk, stride=1, pad=p)
        phases.append(o); hw += [o.shape[1], o.shape[2]]; off.append(p)
    out = torch.empty((N, H, W, packs[0].Cin), dtype=torch.float32, device=g.device)
    _ffi.check(_ffi.lib().cald_train_weave2(_wctx(g), N, H, W, packs[0].Cin, _ptr_array(phases), _int_array(hw), _int_array(off), _p(mask), _p(out)))
    return out


def conv_wgrad(x, g, Cin, Cout, KH, KW, stride, pad, dw, db=None, accumulate=False, row_scale=None):
    _chk(x, "x"); _chk(g, "g")
    N, H, W, ldx = x.shape
    _ffi.check(_ffi.lib().cald_train_conv_wgrad(_wctx(x), N, H, W, _p(x), Cin, ldx, _p(g), Cout, g.shape[-1], KH, KW, stride,
                                                pad, _p(row_scale), _p(dw), _p(db), int(accumulate)))
    return dw


def linear_wgrad(x, g, Cout, dw, db=None, taps=1, accumulate=False):
    _chk(x, "x"); _chk(g, "g")
    R, K = x.shape[0], x[0].numel()
    _ffi.check(_ffi.lib().cald_train_linear_wgrad(_wctx(x), R, _p(x), K, _p(g), Cout, g.shape[-1], taps, _p(dw), _p(db),
                                                  int(accumulate)))
    return dw


def relu_bwd_(g, act=None, scale=None):
    Cc = g.shape[-1]
    _ffi.check(_ffi.lib().cald_train_relu_bwd(_wctx(g), g.numel() // Cc, Cc, _p(g), _p(act), _p(scale)))
    return g


def add(a, b=None, out=None):
    out = out if out is not None else torch.empty_like(a)
    _ffi.check(_ffi.lib().cald_train_add(_wctx(a), a.numel(), _p(out), _p(a), _p(b)))
    return out


def upsample_bwd_(fine, coarse):
    N, Hf, Wf, Cc = fine.shape
    _ffi.check(_ffi.lib().cald_train_upsample_bwd(_wctx(fine), N, Hf, Wf, coarse.shape[1], coarse.shape[2], Cc, _p(fine), _p(coarse)))
    return coarse


def sgd_(param, grad, buf, lr, momentum, weight_decay, first_step):
    _ffi.check(_ffi.lib().cald_train_sgd(_wctx(param), param.numel(), _p(param), _p(grad), _p(buf), lr, momentum, weight_decay,
                                         int(first_step)))


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _int_array(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def rpn_proposals(heads, Hp, Wp, image_sizes, pre_n=2000, post_n=2000, nms_thr=0.7, min_size=1e-3):
    """heads: five [N, Hl, Wl, 16] tensors (3 logits + 12 deltas + 1 pad).  Returns (proposals [N, post_n, 4], counts [N] int32)."""
    N = heads[0].shape[0]
    dev = heads[0].device
    props = torch.zeros((N, post_n, 4), dtype=torch.float32, device=dev)
    counts = torch.zeros(N, dtype=torch.int32, device=dev)
    hw = [v for h in heads for v in (h.shape[1], h.shape[2])]
    _ffi.check(_ffi.lib().cald_train_rpn_proposals(get_ctx(dev.index), N, Hp, Wp, _int_array([v for s in image_sizes for v in s]), _ptr_array(heads),
                                                   _int_array(hw), heads[0].shape[3], pre_n, post_n, nms_thr, min_size, _p(props), _p(counts)))
    return props, counts


def anchors(Hp, Wp, level_hw, device, kind=0):
    """kind 0: Faster R-CNN (3 per location), 1: RetinaNet (9 per location)."""
    n = sum(h * w * (9 if kind else 3) for h, w in level_hw)
    out = torch.empty((n, 4), dtype=torch.float32, device=device)
    _ffi.check(_ffi.lib().cald_train_anchors(get_ctx(device.index), kind, Hp, Wp, _int_array([v for s in level_hw for v in s]), _p(out)))
    return out


def match(boxes, gt, hi, lo, allow_low_quality, out=None):
    if out is None:
        out = torch.empty(boxes.shape[0], dtype=torch.int32, device=boxes.device)
    assert boxes.is_contiguous() and out.is_contiguous()
    _ffi.check(_ffi.lib().cald_train_match(_wctx(boxes), boxes.shape[0], _p(boxes), gt.shape[0], _p(gt), hi, lo, int(allow_low_quality),
                                           _p(out), None))
    return out


def roi_sample_host(slots, n_gt, counts, matched, gt_labels_cat, keys, batch, pos_fraction, pred_ld, num_classes, out=None):
    """cald_train_roi_sample_host: RoI labels + balanced sampling + loss index lists of a whole batch on the host, one call.
    slots / n_gt: sequences or ctypes int arrays; counts: None, a sequence, or an int32 numpy view (the device's counts as copied back).
    out: int64 numpy array of 6 * N * batch elements (e.g. pinned memory) that receives the six lists with stride cap = N * batch:
    table rows | ground-truth rows | labels | pred_idx | pos_rows | image index (float32 in the first 4 R bytes).
    Returns (out, cap, R, n_pos, RoIs per image)."""
    import numpy as np
    N = len(slots)
    cap = N * int(batch)
    assert matched.dtype == np.int32 and matched.flags.c_contiguous and keys.dtype == np.float64 and gt_labels_cat.dtype == np.int64
    if out is None:
        out = np.empty(6 * cap, np.int64)
    assert out.dtype == np.int64 and out.size >= 6 * cap and out.flags.c_contiguous
    R, n_pos, per = C.c_int(), C.c_int(), (C.c_int * N)()
    base = out.ctypes.data
    if counts is None:
        cp = None
    elif isinstance(counts, np.ndarray):
        assert counts.dtype == np.int32 and counts.size >= N
        cp = C.cast(C.c_void_p(counts.ctypes.data), C.POINTER(C.c_int))
    else:
        cp = _int_array(counts)
    ia = lambda v: v if isinstance(v, C.Array) else _int_array(v)
    _ffi.check(_ffi.lib().cald_train_roi_sample_host(N, ia(slots), ia(n_gt), cp, C.c_void_p(matched.ctypes.data), C.c_void_p(gt_labels_cat.ctypes.data),
                                                     C.c_void_p(keys.ctypes.data), int(batch), float(pos_fraction), int(pred_ld), int(num_classes),
                                                     C.c_void_p(base), C.c_void_p(base + 8 * cap), C.c_void_p(base + 16 * cap), C.c_void_p(base + 40 * cap),
                                                     C.c_void_p(base + 32 * cap), C.c_void_p(base + 24 * cap), C.byref(R), C.byref(n_pos), per))
    return out, cap, int(R.value), int(n_pos.value), list(per)


def roi_gather(table, gts_all, idx_dev, cap, R, n_pos, weights):
    """cald_train_roi_gather: RoIAlign rows [R, 5] and the foreground rows' regression targets [n_pos, 4] from the uploaded sampler block."""
    rois = torch.empty((R, 5), dtype=torch.float32, device=table.device)
    box_tgt = torch.empty((n_pos, 4), dtype=torch.float32, device=table.device)
    _ffi.check(_ffi.lib().cald_train_roi_gather(_wctx(table), _p(table), _p(gts_all), _p(idx_dev), int(cap), int(R), int(n_pos),
                                                *[float(w) for w in weights], _p(rois), _p(box_tgt)))
    return rois, box_tgt


def box_encode(reference, proposals, weights):
    out = torch.empty_like(proposals)
    _ffi.check(_ffi.lib().cald_train_box_encode(_wctx(proposals), proposals.shape[0], _p(reference), _p(proposals), *[float(w) for w in weights],
                                                _p(out)))
    return out


def roi_align(feats, rois):
    """feats: four [N, Hl, Wl, C]; rois [R, 5] (image index, box).  Returns [R, 49, C]."""
    R, Cc = rois.shape[0], feats[0].shape[3]
    out = torch.empty((R, 49, Cc), dtype=torch.float32, device=rois.device)
    hw = [v for f in feats for v in (f.shape[1], f.shape[2])]
    _ffi.check(_ffi.lib().cald_train_roi_align(_wctx(rois), _ptr_array(feats), _int_array(hw), Cc, R, _p(rois), _p(out)))
    return out


def roi_align_bwd_(gfeats, rois, gout):
    hw = [v for f in gfeats for v in (f.shape[1], f.shape[2])]
    _ffi.check(_ffi.lib().cald_train_roi_align_bwd(_wctx(rois), gfeats[0].shape[0], _ptr_array(gfeats), _int_array(hw), gfeats[0].shape[3],
                                                   rois.shape[0], _p(rois), _p(gout)))
    return gfeats


def softmax_ce(logits, labels, Ccls, grad=None, gscale=1.0):
    """logits [R, ld] (first Ccls columns are the class logits), labels int64 [R].  Returns the loss as a 1-element tensor."""
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    _ffi.check(_ffi.lib().cald_train_softmax_ce(_wctx(logits), logits.shape[0], Ccls, logits.shape[1], _p(logits), _p(labels), gscale,
                                                _p(loss), _p(grad)))
    return loss


def smooth_l1(pred, idx, target, beta, denom, grad=None, gscale=1.0, weights=None):
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    _ffi.check(_ffi.lib().cald_train_smooth_l1(_wctx(pred), idx.numel(), _p(pred), _p(idx), _p(target), beta, float(denom), _p(weights),
                                               gscale, _p(loss), _p(grad)))
    return loss


def focal_loss(logits_flat, level_pix, N, A, K, ld, matched, gt_labels, gt_off, img_weight, grad=None, gscale=1.0, alpha=0.25):
    loss = torch.empty(1, dtype=torch.float32, device=logits_flat.device)
    _ffi.check(_ffi.lib().cald_train_focal_loss(_wctx(logits_flat), N, _int_array(level_pix), A, K, ld, _p(logits_flat), _p(matched),
                                                _p(gt_labels), _p(gt_off), _p(img_weight), alpha, gscale, _p(loss), _p(grad)))
    return loss


def bce_logits(logits, idx, labels, grad=None, gscale=1.0):
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    _ffi.check(_ffi.lib().cald_train_bce_logits(_wctx(logits), idx.numel(), _p(logits), _p(idx), _p(labels), gscale, _p(loss), _p(grad)))
    return loss


def preprocess(images_u8, sizes, Hp, Wp, remainders=None):
    """images_u8: list of uint8 [H, W, 3] cuda tensors; sizes: list of (Hr, Wr).  Returns [N, Hp, Wp, 4] float32."""
    N = len(images_u8)
    dev = images_u8[0].device
    out = torch.empty((N, Hp, Wp, 4), dtype=torch.float32, device=dev)
    hw = [v for im, (hr, wr) in zip(images_u8, sizes) for v in (im.shape[0], im.shape[1], hr, wr)]
    rem = None
    if remainders is not None and any(r is not None for r in remainders):
        rem = (C.c_void_p * N)(*[(r.data_ptr() if r is not None else None) for r in remainders])
    _ffi.check(_ffi.lib().cald_train_preprocess(get_ctx(dev.index), N, _ptr_array(images_u8), rem, _int_array(hw), Hp, Wp, _p(out)))
    return out


def maxpool(x):
    N, H, W, Cc = x.shape
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), dtype=torch.float32, device=x.device)
    _ffi.check(_ffi.lib().cald_train_maxpool(_wctx(x), N, H, W, Cc, _p(x), _p(out)))
    return out


def subsample2(x):
    N, H, W, Cc = x.shape
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), dtype=torch.float32, device=x.device)
    _ffi.check(_ffi.lib().cald_train_subsample2(_wctx(x), N, H, W, Cc, _p(x), _p(out)))
    return out


def transform_size(H, W, min_size, max_size):
    hr, wr, hp, wp = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _ffi.check(_ffi.lib().cald_op_transform_size(H, W, min_size, max_size, C.byref(hr), C.byref(wr), C.byref(hp), C.byref(wp)))
    return hr.value, wr.value, hp.value, wp.value
