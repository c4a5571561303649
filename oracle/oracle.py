"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  It orchestrates ``libcald_oracle.so`` (oracle/cald_oracle.c) into
  * the scoring half of ``get_uncertainty`` (reference cald_train.py:91-231), and
  * a Faster R-CNN ResNet-FPN forward (reference detection/frcnn_la.py:237-275 on top of
    torchvision 0.8.2 semantics, SURVEY.md Appendix A -- "parity unpinned" for that half).
Every function cites the reference lines it restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f = C.POINTER(C.c_float)
c_i = C.POINTER(C.c_int)
c_u8 = C.POINTER(C.c_uint8)


def build(force=False):
    so = os.path.join(_HERE, "libcald_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("cald_oracle.c", "jpeg_oracle.c", "orc_math.h", "f16x3_oracle.c", "f16x3_conv_body.inc", "mfma_f16_model.h")]
    if force or not os.path.exists(so) or any(os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(so) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcald_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libcald_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_exp.restype = C.c_float
        _LIB.orc_exp.argtypes = [C.c_float]
        _LIB.orc_log.restype = C.c_float
        _LIB.orc_log.argtypes = [C.c_float]
        _LIB.orc_js_divergence.restype = C.c_float
        _LIB.orc_consistency_view.restype = C.c_float
        _LIB.orc_f16x3_join.restype = C.c_float
        _LIB.orc_f16x3_split_word.restype = C.c_uint32
        _LIB.orc_f16x3_split_word.argtypes = [C.c_float]
        _LIB.orc_f16x3_join.argtypes = [C.c_uint32]
        _LIB.orc_set_threads(C.c_int(min(32, os.cpu_count() or 1)))
        _LIB.orc_f16x3_set_threads(C.c_int(min(32, os.cpu_count() or 1)))
    return _LIB


def set_threads(n):
    """OpenMP threads of the conv / linear / RoIAlign loops (results do not depend on it: every output is one chain)."""
    lib().orc_set_threads(C.c_int(max(1, int(n))))
    lib().orc_f16x3_set_threads(C.c_int(max(1, int(n))))


def _p(a, t=c_f):
    return a.ctypes.data_as(t) if a is not None else None


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ----------------------------------------------------------------------------- math
def exp_array(x):
    x = f32(x); y = np.empty_like(x)
    lib().orc_exp_array(_p(x), _p(y), C.c_int(x.size))
    return y


def log_array(x):
    x = f32(x); y = np.empty_like(x)
    lib().orc_log_array(_p(x), _p(y), C.c_int(x.size))
    return y


def py_random(seed, n):
    out = np.empty(n, np.float64)
    lib().orc_py_random(C.c_uint64(seed), C.c_int(n), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


# ----------------------------------------------------------------------------- scoring (A2-A7)
def js_divergence(p, q):
    p = f32(p); q = f32(q)
    return float(lib().orc_js_divergence(_p(p), _p(q), C.c_int(p.size)))


def subsample_indices(n):
    """cald_train.py:110-113"""
    inds = np.empty(50, np.int32)
    k = lib().orc_subsample_indices(C.c_int(n), _p(inds, c_i))
    return inds[:k].copy()


def cls_corr_view(scores, labels, num_cls):
    """cald_train.py:114-117, :194-197"""
    scores = f32(scores); labels = np.ascontiguousarray(labels, dtype=np.int64)
    out = np.empty(num_cls - 1, np.float32)
    lib().orc_cls_corr_view(C.c_int(scores.size), _p(scores), labels.ctypes.data_as(C.POINTER(C.c_int64)),
                            C.c_int(num_cls), _p(out))
    return out


def consistency_view(aug_box, ref_scores_cls, ref_pm, boxes, scores_cls, pm, bp, detail=False):
    """cald_train.py:189-224 for one augmented view."""
    aug_box = f32(aug_box).reshape(-1, 4); ref_scores_cls = f32(ref_scores_cls); ref_pm = f32(ref_pm)
    boxes = f32(boxes).reshape(-1, 4); scores_cls = f32(scores_cls); pm = f32(pm)
    N, M = aug_box.shape[0], boxes.shape[0]
    Ccls = ref_scores_cls.shape[1] if ref_scores_cls.ndim == 2 else 0
    d = None
    if detail:
        d = (np.zeros(N, np.float32), np.zeros(N, np.int32), np.zeros(N, np.float32), np.zeros(N, np.float32))
    r = lib().orc_consistency_view(C.c_int(N), _p(aug_box), _p(ref_scores_cls), _p(ref_pm), C.c_int(M), _p(boxes),
                                   _p(scores_cls), _p(pm), C.c_int(Ccls), C.c_float(bp),
                                   _p(d[0]) if d else None, _p(d[1], c_i) if d else None,
                                   _p(d[2]) if d else None, _p(d[3]) if d else None)
    return (float(r), d) if detail else float(r)


# ----------------------------------------------------------------------------- augmentations (A8-A10)
def flip_boxes(boxes, W):
    """cald_helper.py:23-30: b[:, [0, 2]] = width - bbox[:, [2, 0]]"""
    b = f32(boxes).copy().reshape(-1, 4)
    src = f32(boxes).reshape(-1, 4)
    b[:, 0] = np.float32(W) - src[:, 2]
    b[:, 2] = np.float32(W) - src[:, 0]
    return b


def cutout_rects(seed, H, W, boxes, cut_num=2, remove_thres=0.4, min_thres=0.1):
    """cald_helper.py:88-132 (rectangle selection; fill is applied by preprocess_view)."""
    boxes = f32(boxes).reshape(-1, 4)
    rects = np.zeros((max(cut_num, 1), 4), np.int32)
    n = lib().orc_cutout_rects(C.c_uint64(seed), C.c_int(H), C.c_int(W), C.c_int(boxes.shape[0]), _p(boxes),
                               C.c_int(cut_num), C.c_float(remove_thres), C.c_float(min_thres), _p(rects, c_i))
    return rects[:n].copy()


def pil_resize_bilinear(img, oh, ow):
    """cald_helper.py:47-53: PIL.Image.resize((ow, oh), BILINEAR) on uint8 RGB."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, _ = img.shape
    out = np.empty((oh, ow, 3), np.uint8)
    lib().orc_pil_resize_bilinear(_p(img, c_u8), C.c_int(H), C.c_int(W), _p(out, c_u8), C.c_int(oh), C.c_int(ow))
    return out


def pil_resize_bicubic(img, oh, ow):
    """PIL.Image.resize((ow, oh)) with the default filter (BICUBIC), cald_helper.py:215."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, _ = img.shape
    out = np.empty((oh, ow, 3), np.uint8)
    lib().orc_pil_resize_bicubic(_p(img, c_u8), C.c_int(H), C.c_int(W), _p(out, c_u8), C.c_int(oh), C.c_int(ow))
    return out


def pil_rotate_expand(img, angle):
    """PIL.Image.rotate(angle, expand=True) (NEAREST), cald_helper.py:153.  Returns the rotated uint8 image."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, _ = img.shape
    m = np.zeros(6, np.float64); nh, nw = C.c_int(), C.c_int()
    lib().orc_pil_rotate_matrix(C.c_int(H), C.c_int(W), C.c_double(angle), m.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nh), C.byref(nw))
    out = np.empty((nh.value, nw.value, 3), np.uint8)
    lib().orc_pil_affine_nearest(_p(img, c_u8), C.c_int(H), C.c_int(W), m.ctypes.data_as(C.POINTER(C.c_double)), _p(out, c_u8),
                                 C.c_int(nh.value), C.c_int(nw.value))
    return out


def torch_rand(seed, n):
    out = np.empty(n, np.float32)
    lib().orc_torch_rand(C.c_uint64(seed), C.c_int(n), _p(out))
    return out


def salt_pepper(img, prob, seed):
    """cald_helper.py:78-85 on the uint8 image (torch.rand stream re-seeded with `seed`)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, _ = img.shape
    out = np.empty_like(img)
    lib().orc_salt_pepper(_p(img, c_u8), C.c_int(H), C.c_int(W), C.c_float(prob), C.c_uint64(seed), _p(out, c_u8))
    return out


def rotate_aug(img, boxes, angle):
    """cald_helper.py:135-223: returns (uint8 image of the original size, rotated boxes)."""
    H, W, _ = img.shape
    rot = pil_rotate_expand(img, angle)
    back = pil_resize_bicubic(rot, H, W)
    b = f32(boxes).reshape(-1, 4)
    out = np.empty_like(b)
    lib().orc_rotate_boxes(_p(b), C.c_int(b.shape[0]), C.c_int(H), C.c_int(W), C.c_double(angle), C.c_int(rot.shape[1]),
                           C.c_int(rot.shape[0]), _p(out))
    return back, out


def gaussian_noise(seed, H, W, std=16):
    """cald_helper.py:72-75: the additive term torch.randn(3, H, W) * std / 255.0 (CHW float32)."""
    out = np.empty(3 * H * W, np.float32)
    lib().orc_gaussian_noise(C.c_uint64(seed), C.c_int(3 * H * W), C.c_float(std), _p(out))
    return out.reshape(3, H, W)


def resize_aug(img, ratio):
    H, W, _ = img.shape
    ow, oh = int(W * ratio), int(H * ratio)
    return pil_resize_bilinear(img, oh, ow)


# ----------------------------------------------------------------------------- detector pieces
def transform_size(H, W, min_size, max_size):
    v = [C.c_int() for _ in range(4)]
    lib().orc_transform_size(C.c_int(H), C.c_int(W), C.c_int(min_size), C.c_int(max_size), *[C.byref(x) for x in v])
    return tuple(x.value for x in v)  # Hr, Wr, Hp, Wp


def preprocess_view(img, min_size, max_size, flip=False, rects=None, noise=None):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, _ = img.shape
    Hr, Wr, Hp, Wp = transform_size(H, W, min_size, max_size)
    out = np.empty((Hp, Wp, 4), np.float32)
    r = np.ascontiguousarray(rects, dtype=np.int32).reshape(-1, 4) if rects is not None and len(rects) else None
    lib().orc_preprocess_view(_p(img, c_u8), C.c_int(H), C.c_int(W), C.c_int(int(flip)),
                              C.c_int(0 if r is None else r.shape[0]), _p(r, c_i) if r is not None else None,
                              C.c_int(Hr), C.c_int(Wr), C.c_int(Hp), C.c_int(Wp), _p(out), _p(f32(noise)) if noise is not None else None)
    return out, (Hr, Wr, Hp, Wp)


def conv2d(x, wk, KH, KW, stride, pad, bias=None, bn=None, residual=None, up=None, relu=False):
    """x: [H][W][Cin]; wk: K-major [KH*KW*Cin][Cout]."""
    x = f32(x); H, W, Cin = x.shape
    Cout = wk.shape[1]
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    out = np.empty((Ho, Wo, Cout), np.float32)
    upH = upW = 0
    if up is not None:
        up = f32(up); upH, upW = up.shape[:2]
    lib().orc_conv2d_nhwc(_p(x), C.c_int(H), C.c_int(W), C.c_int(Cin), _p(wk), C.c_int(Cout), C.c_int(KH), C.c_int(KW),
                          C.c_int(stride), C.c_int(pad), _p(bias), _p(bn[0]) if bn else None, _p(bn[1]) if bn else None,
                          _p(f32(residual)) if residual is not None else None, _p(up), C.c_int(upH), C.c_int(upW),
                          C.c_int(int(relu)), _p(out), C.c_int(Ho), C.c_int(Wo))
    return out


def linear(x, wk, bias=None, relu=False):
    x = f32(x); M, K = x.shape
    N = wk.shape[1]
    out = np.empty((M, N), np.float32)
    lib().orc_linear(_p(x), C.c_int(M), C.c_int(K), _p(wk), C.c_int(N), _p(bias), C.c_int(int(relu)), _p(out))
    return out


# ----------------------------------------------------------------------------- CALD_PRECISION_F16X3 (oracle/f16x3_oracle.c, mfma_f16_model.h)
c_u16 = C.POINTER(C.c_uint16)
c_u32 = C.POINTER(C.c_uint32)
c_i16 = C.POINTER(C.c_int16)


def mfma_f16_dot16(A, B, Cin, fast=None):
    """D = C + sum_k A[k] B[k] as v_mfma_f32_32x32x16_f16 computes one output element (gfx950).  A, B: [n][16] fp16 bit patterns
    (uint16), Cin: [n] fp32 bit patterns (uint32).  fast=None: the integer statement of mfma_f16_model.h; 0 / 1: the double-precision
    evaluation the convolution uses (portable loops / AVX-512)."""
    A = np.ascontiguousarray(A, np.uint16).reshape(-1, 16); B = np.ascontiguousarray(B, np.uint16).reshape(-1, 16)
    Cin = np.ascontiguousarray(Cin, np.uint32).reshape(-1)
    assert A.shape == B.shape and A.shape[0] == Cin.shape[0]
    D = np.empty_like(Cin)
    n = C.c_long(A.shape[0])
    if fast is None:
        lib().orc_mfma_f16_dot16(_p(A, c_u16), _p(B, c_u16), _p(Cin, c_u32), _p(D, c_u32), n)
    else:
        if lib().orc_mfma_f16_dot16_fast(_p(A, c_u16), _p(B, c_u16), _p(Cin, c_u32), _p(D, c_u32), n, C.c_int(int(fast))) != 0:
            return None            # this CPU lacks the instruction set
    return D


def uses_f16x3(Cin, Cout, KH, KW):
    """Which layers of a CALD_PRECISION_F16X3 model run on the fp16 matrix pipe (cald_amd/csrc/api.hip make_conv: w16 is packed iff ...);
    the others (the 15-channel RPN head) run the exact fp32 chain in that mode too."""
    coutpad = -(-Cout // 128) * 128 if Cout >= 128 else (-(-Cout // 64) * 64 if Cout >= 64 else -(-Cout // 32) * 32)
    return coutpad % 64 == 0 and ((Cin % 16 == 0 and KH * KW <= 32) or Cin == 4)


def f16x3_weights(wk, KH, KW, Cin):
    """Split weights of one layer (api.hip pack_w16): wk K-major [(kh, kw, cin)][Cout] -> prepared arrays in chain order."""
    wk = f32(wk); K, N = wk.shape
    assert K == KH * KW * Cin
    Kpad = -(-K // 16) * 16; Npad = -(-N // 16) * 16
    vh = np.empty((Kpad, Npad), np.float32); vl = np.empty((Kpad, Npad), np.float32)
    eh = np.empty((Kpad, Npad), np.int16); el = np.empty((Kpad, Npad), np.int16)
    uns = C.c_float()
    S = lib().orc_f16x3_weights(_p(wk), C.c_int(KH), C.c_int(KW), C.c_int(Cin), C.c_int(N), C.c_int(Kpad), C.c_int(Npad),
                                _p(vh), _p(vl), _p(eh, c_i16), _p(el, c_i16), C.byref(uns))
    return dict(vh=vh, vl=vl, eh=eh, el=el, Kpad=Kpad, Npad=Npad, N=N, unscale=uns.value, S=S, KH=KH, KW=KW, Cin=Cin)


def conv2d_f16x3(x, w, KH, KW, stride, pad, bias=None, bn=None, residual=None, up=None, relu=False, in_relu=False):
    """The convolution of CALD_PRECISION_F16X3 (conv_h3.hip / conv_h4.hip), bit for bit.  w: K-major weights or f16x3_weights(...)."""
    x = f32(x); H, W, Cin = x.shape
    if not isinstance(w, dict):
        w = f16x3_weights(w, KH, KW, Cin)
    Cout = w["N"]
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    out = np.empty((Ho, Wo, Cout), np.float32)
    upH = upW = 0
    if up is not None:
        up = f32(up); upH, upW = up.shape[:2]
    lib().orc_conv2d_f16x3_nhwc(_p(x), C.c_int(H), C.c_int(W), C.c_int(Cin), _p(w["vh"]), _p(w["vl"]), _p(w["eh"], c_i16), _p(w["el"], c_i16),
                                C.c_int(w["Kpad"]), C.c_int(w["Npad"]), C.c_float(w["unscale"]), C.c_int(Cout), C.c_int(KH), C.c_int(KW),
                                C.c_int(stride), C.c_int(pad), C.c_int(int(in_relu)), _p(f32(bias)) if bias is not None else None,
                                _p(bn[0]) if bn else None, _p(bn[1]) if bn else None,
                                _p(f32(residual)) if residual is not None else None, _p(up), C.c_int(upH), C.c_int(upW),
                                C.c_int(int(relu)), _p(out), C.c_int(Ho), C.c_int(Wo))
    return out


def f16x3_requantize(x):
    """What an epilogue reads back from a tensor the mode keeps in split form only (h16.h h16_join of split16_word): 22 bits of x."""
    x = f32(x); y = np.empty_like(x)
    lib().orc_f16x3_requantize(_p(x), _p(y), C.c_long(x.size))
    return y


def maxpool3x3s2(x):
    x = f32(x); H, W, Cc = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    out = np.empty((Ho, Wo, Cc), np.float32)
    lib().orc_maxpool3x3s2(_p(x), C.c_int(H), C.c_int(W), C.c_int(Cc), _p(out), C.c_int(Ho), C.c_int(Wo))
    return out


def subsample2(x):
    x = f32(x); H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = np.empty((Ho, Wo, Cc), np.float32)
    lib().orc_subsample2(_p(x), C.c_int(H), C.c_int(W), C.c_int(Cc), _p(out), C.c_int(Ho), C.c_int(Wo))
    return out


def base_anchors(sizes, ratios):
    sizes = f32(sizes); ratios = f32(ratios)
    out = np.empty((ratios.size * sizes.size, 4), np.float32)
    lib().orc_base_anchors(_p(sizes), C.c_int(sizes.size), _p(ratios), C.c_int(ratios.size), _p(out))
    return out


def batched_nms(boxes, scores, groups, thr, max_keep=1 << 30):
    boxes = f32(boxes).reshape(-1, 4); scores = f32(scores); groups = np.ascontiguousarray(groups, dtype=np.int32)
    n = boxes.shape[0]
    keep = np.empty(n + 1, np.int32)
    k = lib().orc_batched_nms(_p(boxes), _p(scores), _p(groups, c_i), C.c_int(n), C.c_float(thr),
                              C.c_int(min(max_keep, n + 1)), _p(keep, c_i))
    return keep[:k].copy()


def rpn_proposals(heads, base, Hp, Wp, Hr, Wr, A=3, pre_n=1000, post_n=1000, nms_thr=0.7, min_size=1e-3):
    L = len(heads)
    heads = [f32(h) for h in heads]
    hp = (c_f * L)(*[_p(h) for h in heads])
    fh = np.array([h.shape[0] for h in heads], np.int32)
    fw = np.array([h.shape[1] for h in heads], np.int32)
    hc = heads[0].shape[2]
    base = f32(base)
    props = np.empty((post_n, 4), np.float32); ps = np.empty(post_n, np.float32)
    n = lib().orc_rpn_proposals(C.c_int(L), hp, _p(fh, c_i), _p(fw, c_i), C.c_int(hc), C.c_int(A), _p(base),
                                C.c_int(Hp), C.c_int(Wp), C.c_int(Hr), C.c_int(Wr), C.c_int(pre_n), C.c_int(post_n),
                                C.c_float(nms_thr), C.c_float(min_size), _p(props), _p(ps))
    return props[:n].copy(), ps[:n].copy()


def roi_align(feats, rois):
    L = len(feats)
    feats = [f32(f) for f in feats]
    fp = (c_f * L)(*[_p(f) for f in feats])
    fh = np.array([f.shape[0] for f in feats], np.int32)
    fw = np.array([f.shape[1] for f in feats], np.int32)
    Cc = feats[0].shape[2]
    rois = f32(rois).reshape(-1, 4)
    R = rois.shape[0]
    out = np.empty((R, 49, Cc), np.float32)
    lib().orc_roi_align(C.c_int(L), fp, _p(fh, c_i), _p(fw, c_i), C.c_int(Cc), _p(rois), C.c_int(R), _p(out))
    return out


def resize_boxes(boxes, Hr, Wr, Ho, Wo):
    """frcnn_la.py:307-315 (resized (Hr, Wr) -> original (Ho, Wo)); the function orc_frcnn_postprocess applies to boxes and props."""
    boxes = f32(boxes).reshape(-1, 4)
    out = np.empty_like(boxes)
    lib().orc_resize_boxes(_p(boxes), C.c_int(boxes.shape[0]), C.c_int(Hr), C.c_int(Wr), C.c_int(Ho), C.c_int(Wo), _p(out))
    return out


def frcnn_postprocess(logits, deltas, proposals, Hr, Wr, Ho, Wo, score_thr=0.05, nms_thr=0.5, det_max=100):
    logits = f32(logits); deltas = f32(deltas); proposals = f32(proposals).reshape(-1, 4)
    R, Cc = logits.shape
    ob = np.empty((det_max, 4), np.float32); os_ = np.empty(det_max, np.float32); ol = np.empty(det_max, np.int64)
    op = np.empty((det_max, 4), np.float32); opm = np.empty(det_max, np.float32); osc = np.empty((det_max, Cc), np.float32)
    n = lib().orc_frcnn_postprocess(C.c_int(R), C.c_int(Cc), _p(logits), _p(deltas), _p(proposals), C.c_int(Hr), C.c_int(Wr),
                                    C.c_int(Ho), C.c_int(Wo), C.c_float(score_thr), C.c_float(nms_thr), C.c_int(det_max),
                                    _p(ob), _p(os_), ol.ctypes.data_as(C.POINTER(C.c_int64)), _p(op), _p(opm), _p(osc))
    return dict(boxes=ob[:n].copy(), scores=os_[:n].copy(), labels=ol[:n].copy(), props=op[:n].copy(),
                prob_max=opm[:n].copy(), scores_cls=osc[:n].copy())


# ----------------------------------------------------------------------------- model preparation
def _kmajor_conv(w):
    """torch conv weight [Cout][Cin][KH][KW] -> K-major [(kh,kw,cin)][Cout]"""
    w = f32(w)
    return np.ascontiguousarray(w.transpose(2, 3, 1, 0).reshape(-1, w.shape[0]))


def _frozen_bn(sd, prefix, eps=1e-5):
    """torchvision FrozenBatchNorm2d: scale = w * rsqrt(var + eps); bias = b - mean * scale"""
    w, b = f32(sd[prefix + ".weight"]), f32(sd[prefix + ".bias"])
    rm, rv = f32(sd[prefix + ".running_mean"]), f32(sd[prefix + ".running_var"])
    scale = (w * (np.float32(1.0) / np.sqrt(rv + np.float32(eps)))).astype(np.float32)
    shift = (b - rm * scale).astype(np.float32)
    return scale, shift


RESNET_LAYERS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}


def prepare_frcnn(sd, num_classes, depth=50):
    """sd: torchvision-layout state dict of numpy arrays (SURVEY section 8b key layout)."""
    sd = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in sd.items()}
    P = {"num_classes": num_classes, "depth": depth}
    w1 = f32(sd["backbone.body.conv1.weight"])
    w1p = np.zeros((64, 4, 7, 7), np.float32); w1p[:, :3] = w1
    P["conv1"] = (_kmajor_conv(w1p), _frozen_bn(sd, "backbone.body.bn1"))
    blocks = []
    for li, nb in enumerate(RESNET_LAYERS[depth]):
        for bi in range(nb):
            pre = "backbone.body.layer%d.%d" % (li + 1, bi)
            blk = {"stride": 2 if (bi == 0 and li > 0) else 1, "name": pre}
            for ci in (1, 2, 3):
                blk["conv%d" % ci] = (_kmajor_conv(sd[pre + ".conv%d.weight" % ci]), _frozen_bn(sd, pre + ".bn%d" % ci))
            if pre + ".downsample.0.weight" in sd:
                blk["down"] = (_kmajor_conv(sd[pre + ".downsample.0.weight"]), _frozen_bn(sd, pre + ".downsample.1"))
            blk["layer_end"] = (bi == nb - 1)
            blocks.append(blk)
    P["blocks"] = blocks
    P["fpn_inner"] = [(_kmajor_conv(sd["backbone.fpn.inner_blocks.%d.weight" % i]), f32(sd["backbone.fpn.inner_blocks.%d.bias" % i])) for i in range(4)]
    P["fpn_layer"] = [(_kmajor_conv(sd["backbone.fpn.layer_blocks.%d.weight" % i]), f32(sd["backbone.fpn.layer_blocks.%d.bias" % i])) for i in range(4)]
    P["rpn_conv"] = (_kmajor_conv(sd["rpn.head.conv.weight"]), f32(sd["rpn.head.conv.bias"]))
    wc, bc = f32(sd["rpn.head.cls_logits.weight"]), f32(sd["rpn.head.cls_logits.bias"])
    wb, bb = f32(sd["rpn.head.bbox_pred.weight"]), f32(sd["rpn.head.bbox_pred.bias"])
    P["rpn_head"] = (_kmajor_conv(np.concatenate([wc, wb], 0)), np.concatenate([bc, bb]).astype(np.float32))
    w6 = f32(sd["roi_heads.box_head.fc6.weight"])  # [1024][c*49+bin] -> chain order (bin, c)
    w6 = w6.reshape(w6.shape[0], 256, 49).transpose(0, 2, 1).reshape(w6.shape[0], -1)
    P["fc6"] = (np.ascontiguousarray(w6.T), f32(sd["roi_heads.box_head.fc6.bias"]))
    P["fc7"] = (np.ascontiguousarray(f32(sd["roi_heads.box_head.fc7.weight"]).T), f32(sd["roi_heads.box_head.fc7.bias"]))
    wcs, bcs = f32(sd["roi_heads.box_predictor.cls_score.weight"]), f32(sd["roi_heads.box_predictor.cls_score.bias"])
    wbp, bbp = f32(sd["roi_heads.box_predictor.bbox_pred.weight"]), f32(sd["roi_heads.box_predictor.bbox_pred.bias"])
    P["pred"] = (np.ascontiguousarray(np.concatenate([wcs, wbp], 0).T), np.concatenate([bcs, bbp]).astype(np.float32))
    P["anchors"] = np.stack([base_anchors([s], [0.5, 1.0, 2.0]) for s in (32, 64, 128, 256, 512)])  # [5][3][4]
    return P


def _f16x3_layer(P, name, wk, KH, KW, Cin):
    cache = P.setdefault("_f16x3_weights", {})
    if name not in cache:
        cache[name] = f16x3_weights(wk, KH, KW, Cin)
    return cache[name]


def _conv(P, name, x, wk, KH, KW, stride, pad, in_relu=False, **kw):
    """A conv layer of the model.  precision "fp32": the exact fp32 chain.  precision "f16x3" (CALD_PRECISION_F16X3): the layers the
    library runs on the fp16 matrix pipe (uses_f16x3) follow conv2d_f16x3; a residual / top-down operand is then read back from a tensor
    the mode keeps in split form only (api.hip fwd_layout `only(...)`: block outputs, the downsample branch, the FPN laterals), i.e. as
    h16_join(split16_word(value))."""
    Cin, Cout = x.shape[2], wk.shape[1]
    if P.get("precision", "fp32") == "f16x3" and uses_f16x3(Cin, Cout, KH, KW):
        kw = dict(kw)
        for k in ("residual", "up"):
            if kw.get(k) is not None:
                kw[k] = f16x3_requantize(kw[k])
        return conv2d_f16x3(x, _f16x3_layer(P, name, wk, KH, KW, Cin), KH, KW, stride, pad, in_relu=in_relu, **kw)
    if in_relu:
        x = np.maximum(x, np.float32(0.0))
    return conv2d(x, wk, KH, KW, stride, pad, **kw)


def _linear(P, name, x, wk, bias=None, relu=False):
    if P.get("precision", "fp32") == "f16x3" and uses_f16x3(x.shape[1], wk.shape[1], 1, 1):
        y = conv2d_f16x3(f32(x)[None], _f16x3_layer(P, name, wk, 1, 1, x.shape[1]), 1, 1, 1, 0, bias=bias, relu=relu)
        return y[0]
    return linear(x, wk, bias, relu)


def frcnn_backbone(P, x, keep=None):
    """ResNet body + FPN (rows A15, A16).  x: [Hp][Wp][4].  Returns [P2..P5, pool]."""
    wk, bn = P["conv1"]
    y = _conv(P, "backbone.body.conv1.weight", x, wk, 7, 7, 2, 3, bn=bn, relu=True)
    if keep is not None: keep["conv1"] = y
    y = maxpool3x3s2(y)
    if keep is not None: keep["pool1"] = y
    feats = []
    for bi, blk in enumerate(P["blocks"]):
        idn = y
        pre = blk["name"]
        if "down" in blk:
            idn = _conv(P, pre + ".downsample.0.weight", y, blk["down"][0], 1, 1, blk["stride"], 0, bn=blk["down"][1])
        o = _conv(P, pre + ".conv1.weight", y, blk["conv1"][0], 1, 1, 1, 0, bn=blk["conv1"][1], relu=True)
        o = _conv(P, pre + ".conv2.weight", o, blk["conv2"][0], 3, 3, blk["stride"], 1, bn=blk["conv2"][1], relu=True)
        y = _conv(P, pre + ".conv3.weight", o, blk["conv3"][0], 1, 1, 1, 0, bn=blk["conv3"][1], residual=idn, relu=True)
        if blk["layer_end"]:
            feats.append(y)
    if keep is not None: keep["C"] = feats
    inner = [None] * 4
    fi, fl = "backbone.fpn.inner_blocks.%d.weight", "backbone.fpn.layer_blocks.%d.weight"
    inner[3] = _conv(P, fi % 3, feats[3], P["fpn_inner"][3][0], 1, 1, 1, 0, bias=P["fpn_inner"][3][1])
    for i in (2, 1, 0):
        inner[i] = _conv(P, fi % i, feats[i], P["fpn_inner"][i][0], 1, 1, 1, 0, bias=P["fpn_inner"][i][1], up=inner[i + 1])
    outs = [_conv(P, fl % i, inner[i], P["fpn_layer"][i][0], 3, 3, 1, 1, bias=P["fpn_layer"][i][1]) for i in range(4)]
    outs.append(subsample2(outs[3]))
    return outs


def frcnn_forward(P, img, min_size, max_size, flip=False, rects=None, keep=None,
                  score_thr=0.05, nms_thr=0.5, det_max=100, noise=None):
    """frcnn_la.py:237-275 for ONE view (batch 1, like the reference)."""
    H, W, _ = img.shape
    x, (Hr, Wr, Hp, Wp) = preprocess_view(img, min_size, max_size, flip, rects, noise)
    if keep is not None: keep["input"] = x; keep["sizes"] = (Hr, Wr, Hp, Wp)
    feats = frcnn_backbone(P, x, keep)
    if keep is not None: keep["fpn"] = feats
    heads = []
    for f in feats:
        t = _conv(P, "rpn.head.conv.weight", f, P["rpn_conv"][0], 3, 3, 1, 1, bias=P["rpn_conv"][1], relu=True)
        heads.append(conv2d(t, P["rpn_head"][0], 1, 1, 1, 0, bias=P["rpn_head"][1]))
    if keep is not None: keep["rpn_head"] = heads
    props, pscores = rpn_proposals(heads, P["anchors"], Hp, Wp, Hr, Wr)
    if keep is not None: keep["proposals"] = props; keep["proposal_scores"] = pscores
    Cn = P["num_classes"]
    if props.shape[0] == 0:
        z = np.zeros
        return dict(boxes=z((0, 4), np.float32), scores=z(0, np.float32), labels=z(0, np.int64), props=z((0, 4), np.float32),
                    prob_max=z(0, np.float32), scores_cls=z((0, Cn), np.float32))
    roi = roi_align(feats[:4], props)
    if keep is not None: keep["roi"] = roi
    h = _linear(P, "roi_heads.box_head.fc6.weight", roi.reshape(roi.shape[0], -1), P["fc6"][0], P["fc6"][1], relu=True)
    h = _linear(P, "roi_heads.box_head.fc7.weight", h, P["fc7"][0], P["fc7"][1], relu=True)
    pred = _linear(P, "roi_heads.box_predictor.cls_score.weight", h, P["pred"][0], P["pred"][1])
    if keep is not None: keep["fc7"] = h; keep["pred"] = pred
    return frcnn_postprocess(pred[:, :Cn], pred[:, Cn:], props, Hr, Wr, H, W, score_thr, nms_thr, det_max)


# ----------------------------------------------------------------------------- RetinaNet (rows A21, A22)
def retina_anchor_sizes():
    """retinanet_cal.py:346-347"""
    return [(x, int(x * 2 ** (1.0 / 3)), int(x * 2 ** (2.0 / 3))) for x in [32, 64, 128, 256, 512]]


def prepare_retinanet(sd, num_classes, depth=50):
    """torchvision-layout RetinaNet state dict (SURVEY section 8b key layout) -> oracle weights."""
    sd = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in sd.items()}
    P = {"num_classes": num_classes, "depth": depth, "arch": "retinanet"}
    w1 = f32(sd["backbone.body.conv1.weight"])
    w1p = np.zeros((64, 4, 7, 7), np.float32); w1p[:, :3] = w1
    P["conv1"] = (_kmajor_conv(w1p), _frozen_bn(sd, "backbone.body.bn1"))
    blocks = []
    for li, nb in enumerate(RESNET_LAYERS[depth]):
        for bi in range(nb):
            pre = "backbone.body.layer%d.%d" % (li + 1, bi)
            blk = {"stride": 2 if (bi == 0 and li > 0) else 1, "name": pre}
            for ci in (1, 2, 3):
                blk["conv%d" % ci] = (_kmajor_conv(sd[pre + ".conv%d.weight" % ci]), _frozen_bn(sd, pre + ".bn%d" % ci))
            if pre + ".downsample.0.weight" in sd:
                blk["down"] = (_kmajor_conv(sd[pre + ".downsample.0.weight"]), _frozen_bn(sd, pre + ".downsample.1"))
            blk["layer_end"] = (bi == nb - 1)
            blocks.append(blk)
    P["blocks"] = blocks
    P["fpn_inner"] = [(_kmajor_conv(sd["backbone.fpn.inner_blocks.%d.weight" % i]), f32(sd["backbone.fpn.inner_blocks.%d.bias" % i])) for i in range(3)]
    P["fpn_layer"] = [(_kmajor_conv(sd["backbone.fpn.layer_blocks.%d.weight" % i]), f32(sd["backbone.fpn.layer_blocks.%d.bias" % i])) for i in range(3)]
    P["p6"] = (_kmajor_conv(sd["backbone.fpn.extra_blocks.p6.weight"]), f32(sd["backbone.fpn.extra_blocks.p6.bias"]))
    P["p7"] = (_kmajor_conv(sd["backbone.fpn.extra_blocks.p7.weight"]), f32(sd["backbone.fpn.extra_blocks.p7.bias"]))
    for head, last in (("classification_head", "cls_logits"), ("regression_head", "bbox_reg")):
        P[head] = [(_kmajor_conv(sd["head.%s.conv.%d.weight" % (head, 2 * i)]), f32(sd["head.%s.conv.%d.bias" % (head, 2 * i)])) for i in range(4)]
        P[head + "_out"] = (_kmajor_conv(sd["head.%s.%s.weight" % (head, last)]), f32(sd["head.%s.%s.bias" % (head, last)]))
    P["anchors"] = np.stack([base_anchors(list(s), [0.5, 1.0, 2.0]) for s in retina_anchor_sizes()])  # [5][9][4]
    return P


def retina_backbone(P, x, keep=None):
    """ResNet body (C3..C5) + FPN + LastLevelP6P7(256, 256) (retinanet_cal.py:618-619)."""
    wk, bn = P["conv1"]
    y = maxpool3x3s2(_conv(P, "backbone.body.conv1.weight", x, wk, 7, 7, 2, 3, bn=bn, relu=True))
    feats = []
    for blk in P["blocks"]:
        idn = y
        pre = blk["name"]
        if "down" in blk:
            idn = _conv(P, pre + ".downsample.0.weight", y, blk["down"][0], 1, 1, blk["stride"], 0, bn=blk["down"][1])
        o = _conv(P, pre + ".conv1.weight", y, blk["conv1"][0], 1, 1, 1, 0, bn=blk["conv1"][1], relu=True)
        o = _conv(P, pre + ".conv2.weight", o, blk["conv2"][0], 3, 3, blk["stride"], 1, bn=blk["conv2"][1], relu=True)
        y = _conv(P, pre + ".conv3.weight", o, blk["conv3"][0], 1, 1, 1, 0, bn=blk["conv3"][1], residual=idn, relu=True)
        if blk["layer_end"]:
            feats.append(y)
    feats = feats[1:]   # returned_layers=[2, 3, 4]
    if keep is not None: keep["C"] = feats
    inner = [None] * 3
    fi, fl = "backbone.fpn.inner_blocks.%d.weight", "backbone.fpn.layer_blocks.%d.weight"
    inner[2] = _conv(P, fi % 2, feats[2], P["fpn_inner"][2][0], 1, 1, 1, 0, bias=P["fpn_inner"][2][1])
    for i in (1, 0):
        inner[i] = _conv(P, fi % i, feats[i], P["fpn_inner"][i][0], 1, 1, 1, 0, bias=P["fpn_inner"][i][1], up=inner[i + 1])
    outs = [_conv(P, fl % i, inner[i], P["fpn_layer"][i][0], 3, 3, 1, 1, bias=P["fpn_layer"][i][1]) for i in range(3)]
    p6 = _conv(P, "backbone.fpn.extra_blocks.p6.weight", outs[2], P["p6"][0], 3, 3, 2, 1, bias=P["p6"][1])
    p7 = _conv(P, "backbone.fpn.extra_blocks.p7.weight", p6, P["p7"][0], 3, 3, 2, 1, in_relu=True, bias=P["p7"][1])     # ReLU while staging
    return outs + [p6, p7]


def retina_postprocess(cls, reg, anchors, Hp, Wp, Hr, Wr, Ho, Wo, K, A=9, score_thr=0.05, nms_thr=0.5, per_class=300):
    L = len(cls)
    cls = [f32(c) for c in cls]; reg = [f32(r) for r in reg]
    cp = (c_f * L)(*[_p(c) for c in cls]); rp = (c_f * L)(*[_p(r) for r in reg])
    fh = np.array([c.shape[0] for c in cls], np.int32); fw = np.array([c.shape[1] for c in cls], np.int32)
    cap = K * per_class
    ob = np.empty((cap, 4), np.float32); os_ = np.empty(cap, np.float32); ol = np.empty(cap, np.int64)
    opm = np.empty(cap, np.float32); osc = np.empty((cap, K), np.float32)
    anchors = f32(anchors)
    n = lib().orc_retina_postprocess(C.c_int(L), cp, rp, _p(fh, c_i), _p(fw, c_i), C.c_int(A), C.c_int(K), _p(anchors),
                                     C.c_int(Hp), C.c_int(Wp), C.c_int(Hr), C.c_int(Wr), C.c_int(Ho), C.c_int(Wo),
                                     C.c_float(score_thr), C.c_float(nms_thr), C.c_int(per_class), C.c_float(1e-2),
                                     _p(ob), _p(os_), ol.ctypes.data_as(C.POINTER(C.c_int64)), _p(opm), _p(osc))
    return dict(boxes=ob[:n].copy(), scores=os_[:n].copy(), labels=ol[:n].copy(), prob_max=opm[:n].copy(), scores_cls=osc[:n].copy())


def retina_forward(P, img, min_size, max_size, flip=False, rects=None, keep=None, score_thr=0.05, nms_thr=0.5, per_class=300, noise=None):
    """retinanet_cal.py:492-575 for ONE view."""
    H, W, _ = img.shape
    x, (Hr, Wr, Hp, Wp) = preprocess_view(img, min_size, max_size, flip, rects, noise)
    feats = retina_backbone(P, x, keep)
    if keep is not None: keep["fpn"] = feats; keep["sizes"] = (Hr, Wr, Hp, Wp)
    cls, reg = [], []
    for f in feats:
        t = f
        for i, (wk, b) in enumerate(P["classification_head"]):
            t = _conv(P, "head.classification_head.conv.%d.weight" % (2 * i), t, wk, 3, 3, 1, 1, bias=b, relu=True)
        cls.append(_conv(P, "head.classification_head.cls_logits.weight", t, P["classification_head_out"][0], 3, 3, 1, 1, bias=P["classification_head_out"][1]))
        t = f
        for i, (wk, b) in enumerate(P["regression_head"]):
            t = _conv(P, "head.regression_head.conv.%d.weight" % (2 * i), t, wk, 3, 3, 1, 1, bias=b, relu=True)
        reg.append(_conv(P, "head.regression_head.bbox_reg.weight", t, P["regression_head_out"][0], 3, 3, 1, 1, bias=P["regression_head_out"][1]))
    if keep is not None: keep["cls"] = cls; keep["reg"] = reg
    return retina_postprocess(cls, reg, P["anchors"], Hp, Wp, Hr, Wr, H, W, P["num_classes"], 9, score_thr, nms_thr, per_class)


def detector_forward(P, img, min_size, max_size, flip=False, rects=None, keep=None, noise=None):
    if P.get("arch") == "retinanet":
        return retina_forward(P, img, min_size, max_size, flip, rects, keep, noise=noise)
    return frcnn_forward(P, img, min_size, max_size, flip, rects, keep, noise=noise)


# ----------------------------------------------------------------------------- the sweep (A1)
def image_seed(base_seed, pool_pos):
    return (int(base_seed) * 1000003 + int(pool_pos)) & 0xFFFFFFFFFFFFFFFF


AUG_KINDS = ("flip", "gauss", "color_adjust", "color_swap", "salt_pepper", "cutout", "resize", "rotate")
COLOR_PERMS = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))     # cald_helper.py:57-58


def expand_augs(augs):
    """The augmented views get_uncertainty builds for a list of aug names, in ITS order (cald_train.py:123-183;
    the order of `augs` itself is irrelevant there).  Returns [(kind, param)]."""
    out = []
    if "flip" in augs:
        out.append(("flip", 0.0))
    if "ga" in augs:
        out.append(("gauss", 16.0))
    if "multi_ga" in augs:
        out += [("gauss", float(i * 8)) for i in range(1, 7)]
    if "color_adjust" in augs:
        out.append(("color_adjust", 1.5))
    if "color_swap" in augs:
        out.append(("color_swap", 0.0))
    if "multi_color_adjust" in augs:
        raise NameError("name 'reference_boxes' is not defined")          # cald_train.py:148, as the reference does
    if "sp" in augs:
        out.append(("salt_pepper", 0.1))
    if "multi_sp" in augs:
        out += [("salt_pepper", i * 0.05) for i in range(1, 7)]
    if "cut_out" in augs:
        out.append(("cutout", 2.0))
    if "multi_cut_out" in augs:
        out += [("cutout", float(i)) for i in range(1, 5)]
    if "multi_resize" in augs:
        out += [("resize", i * 0.1) for i in range(7, 10)]
    if "larger_resize" in augs:
        out.append(("resize", 1.2))
    if "smaller_resize" in augs:
        out.append(("resize", 0.8))
    if "rotation" in augs:
        out.append(("rotate", 5.0))
    return out


def color_adjust(img, factor):
    """cald_helper.py:65-69 ColorAdjust on the uint8 image (PIL ImageEnhance Brightness -> Contrast -> Color)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty_like(img)
    lib().orc_color_adjust(_p(img, c_u8), C.c_int(img.shape[0]), C.c_int(img.shape[1]), C.c_float(factor), _p(out, c_u8))
    return out


def torch_stream(seed, n, ops):
    """ops: [(kind 0 randn*std/255 | 1 rand, std)] drawn in order from one torch CPU generator seeded with `seed`."""
    kinds = np.array([k for k, _ in ops], np.int32)
    stds = f32([p for _, p in ops])
    out = np.empty((len(ops), n), np.float32)
    if len(ops):
        lib().orc_torch_stream(C.c_uint64(seed), C.c_int(len(ops)), _p(kinds, c_i), C.c_int(n), _p(stds), _p(out))
    return out


def salt_pepper_from_uniforms(img, prob, u):
    """cald_helper.py:78-85 given the torch.rand(3, H, W) draws `u` (CHW order)."""
    H, W, _ = img.shape
    noise = u.reshape(3, H, W).transpose(1, 2, 0)
    lo, hi = np.float32(prob / 2.0), np.float32(1.0 - prob / 2.0)
    out = img.copy()
    mx, mn = img.max(), img.min()
    out[noise < lo] = mx
    out[noise > hi] = mn
    return out


def build_views(img, augs, ref, seed):
    """cald_train.py:123-183.  One torch generator (GaussianNoise / SaltPepperNoise draws) and one Python `random`
    generator (ColorSwap / cutout draws) per image, both seeded with `seed` and consumed in call order.
    Returns list of (src_image, flip, rects, aug_boxes[, noise])."""
    H, W, _ = img.shape
    rb = ref["boxes"]
    specs = expand_augs(augs)
    ops = [((0, p) if k == "gauss" else (1, 0.0)) for k, p in specs if k in ("gauss", "salt_pepper")]
    draws = torch_stream(seed, 3 * H * W, ops)
    lib().orc_pyrandom_new.restype = C.c_void_p
    st = C.c_void_p(lib().orc_pyrandom_new(C.c_uint64(seed)))
    views, d = [], 0
    for kind, p in specs:
        if kind == "flip":
            views.append((img, True, None, flip_boxes(rb, W)))
        elif kind == "gauss":
            views.append((img, False, None, rb, draws[d].reshape(3, H, W))); d += 1
        elif kind == "color_adjust":
            views.append((color_adjust(img, p), False, None, rb))
        elif kind == "color_swap":
            perm = COLOR_PERMS[lib().orc_pyrandom_randbelow(st, C.c_int(6))]
            views.append((np.ascontiguousarray(img[:, :, list(perm)]), False, None, rb))
        elif kind == "salt_pepper":
            views.append((salt_pepper_from_uniforms(img, p, draws[d]), False, None, rb)); d += 1
        elif kind == "cutout":
            b = f32(rb).reshape(-1, 4)
            rects = np.zeros((4, 4), np.int32)
            n = lib().orc_cutout_rects_st(st, C.c_int(H), C.c_int(W), C.c_int(b.shape[0]), _p(b), C.c_int(int(p)),
                                          C.c_float(0.4), C.c_float(0.1), _p(rects, c_i))
            views.append((img, False, rects[:n].copy(), rb))
        elif kind == "resize":
            views.append((resize_aug(img, p), False, None, (f32(rb) * np.float32(p)).astype(np.float32)))
        elif kind == "rotate":
            ri, rbx = rotate_aug(img, rb, p)
            views.append((ri, False, None, rbx))
    lib().orc_pyrandom_free(st)
    return views


def subsample_ref(out):
    """cald_train.py:110-113"""
    n = out["scores"].shape[0]
    if n > 40:
        inds = subsample_indices(n)
        return {k: v[inds] for k, v in out.items()}
    return out


def score_image(ref, aug_outs, aug_boxes, num_cls, bp):
    """cald_train.py:114-121, :187-228 given detector outputs.  ref is already sub-sampled."""
    cls_corrs = [cls_corr_view(ref["scores"], ref["labels"], num_cls).astype(np.float64)]
    if ref["boxes"].shape[0] == 0:
        return 0.0, np.mean(cls_corrs, axis=0)
    cons = []
    for out, ab in zip(aug_outs, aug_boxes):
        cls_corrs.append(cls_corr_view(out["scores"], out["labels"], num_cls).astype(np.float64))
        cons.append(consistency_view(ab, ref["scores_cls"], ref["prob_max"], out["boxes"], out["scores_cls"],
                                     out["prob_max"], bp))
    return float(np.mean(np.array(cons, np.float64))), np.mean(np.array(cls_corrs), axis=0)


def get_uncertainty(P, images, augs, num_cls, bp=1.3, min_size=600, max_size=1000, base_seed=0, positions=None):
    """Restatement of cald_train.py:91-231 over an in-memory pool of uint8 HWC images."""
    consistency_all, cls_all = [], []
    for pos, img in enumerate(images):
        gpos = pos if positions is None else positions[pos]
        ref = subsample_ref(detector_forward(P, img, min_size, max_size))
        if ref["boxes"].shape[0] == 0:
            c, cc = score_image(ref, [], [], num_cls, bp)
        else:
            views = build_views(img, augs, ref, image_seed(base_seed, gpos))
            outs = [detector_forward(P, v[0], min_size, max_size, v[1], v[2], noise=(v[4] if len(v) > 4 else None)) for v in views]
            c, cc = score_image(ref, outs, [v[3] for v in views], num_cls, bp)
        consistency_all.append(c); cls_all.append(cc)
    return consistency_all, cls_all


# ----------------------------------------------------------------------------- baseline sweeps
def lt_uncertainty(out):
    """lt_c_train.py:105-121 for one image's detections."""
    b = f32(out["boxes"]).reshape(-1, 4); p = f32(out["props"]).reshape(-1, 4); pm = f32(out["prob_max"])
    lib().orc_lt_uncertainty.restype = C.c_float
    return float(lib().orc_lt_uncertainty(C.c_int(b.shape[0]), _p(b), _p(p), _p(pm)))


def lt_get_uncertainty(P, images, min_size=600, max_size=1000):
    return [lt_uncertainty(detector_forward(P, img, min_size, max_size)) for img in images]


def gaussian_noise_seq(seed, H, W, stds):
    stds = f32(stds)
    out = np.empty((len(stds), 3 * H * W), np.float32)
    lib().orc_gaussian_noise_seq(C.c_uint64(seed), C.c_int(3 * H * W), C.c_int(len(stds)), _p(stds), _p(out))
    return out.reshape(len(stds), 3, H, W)


def topk_indices(values, k):
    """torch.topk(values, k)[1] with the contract's tie rule (value desc, index asc)."""
    v = f32(values)
    return np.array(sorted(range(v.size), key=lambda i: (-v[i], i))[:k], np.int64)


def ls_score_image(ref, aug_outs):
    """ls_c_train.py:118-153 given the detector outputs of the reference view and the six noisy views."""
    if ref["boxes"].shape[0] == 0:
        return 0.0
    rb, pm = f32(ref["boxes"]), f32(ref["prob_max"])
    if rb.shape[0] > 30:
        inds = topk_indices(pm, 30)
        rb, pm = rb[inds], pm[inds]
    U = float(np.max(np.float32(1) - pm))
    stab = [0.0] * rb.shape[0]
    for o in aug_outs:
        M = o["boxes"].shape[0]
        if M == 0:
            continue
        row = np.empty(rb.shape[0], np.float32)
        lib().orc_max_iou_rows(C.c_int(rb.shape[0]), _p(rb), C.c_int(M), _p(f32(o["boxes"]).reshape(-1, 4)), _p(row))
        for i in range(rb.shape[0]):
            stab[i] += float(row[i])
    st = np.array(stab) / 6.0
    return float(np.sum(pm * st) / np.sum(pm) - U)


def ls_get_uncertainty(P, images, min_size=600, max_size=1000, base_seed=0, positions=None):
    res = []
    for pos, img in enumerate(images):
        gpos = pos if positions is None else positions[pos]
        ref = detector_forward(P, img, min_size, max_size)
        if ref["boxes"].shape[0] == 0:
            res.append(0.0)
            continue
        H, W, _ = img.shape
        noise = gaussian_noise_seq(image_seed(base_seed, gpos), H, W, [8.0 * i for i in range(1, 7)])
        outs = [detector_forward(P, img, min_size, max_size, noise=noise[k]) for k in range(6)]
        res.append(ls_score_image(ref, outs))
    return res


# ---------------------------------------------------------------------------------------------
# JPEG decode (oracle/jpeg_oracle.c).  Returns uint8 [H][W][3] like
# PIL.Image.open(...).convert('RGB'); raises NotImplementedError for JPEG flavours outside the
# supported set (progressive, CMYK, 4:4:0, ...), ValueError for broken files.
# ---------------------------------------------------------------------------------------------
def jpeg_info(data):
    buf = np.frombuffer(bytes(data), np.uint8)
    H, W, nc = C.c_int(), C.c_int(), C.c_int()
    rc = lib().orc_jpeg_info(buf.ctypes.data_as(c_u8), C.c_size_t(buf.size), C.byref(H), C.byref(W), C.byref(nc))
    if rc == -2:
        raise NotImplementedError("unsupported JPEG flavour")
    if rc:
        raise ValueError("not a decodable JPEG")
    return H.value, W.value, nc.value


def jpeg_decode(data):
    H, W, _ = jpeg_info(data)
    buf = np.frombuffer(bytes(data), np.uint8)
    out = np.empty((H, W, 3), np.uint8)
    rc = lib().orc_jpeg_decode(buf.ctypes.data_as(c_u8), C.c_size_t(buf.size), out.ctypes.data_as(c_u8))
    if rc:
        raise ValueError("JPEG decode failed (%d)" % rc)
    return out
