"""ctypes binding of libcaldhip.so (include/cald_hip.h).

This is the thin FFI shim of the drop-in boundary: every symbol declared in include/cald_hip.h is
bound here with its exact C signature.  There is NO fallback: if the library is missing, or no
MI355X is visible when a compute call is made, the call raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcaldhip.so")
_lib = None

c_f = C.POINTER(C.c_float)
c_i = C.POINTER(C.c_int)
c_i64 = C.POINTER(C.c_int64)
c_d = C.POINTER(C.c_double)
c_u8 = C.POINTER(C.c_uint8)


class ModelCfg(C.Structure):
    _fields_ = [("arch", C.c_int), ("depth", C.c_int), ("num_classes", C.c_int), ("min_size", C.c_int),
                ("max_size", C.c_int), ("box_score_thresh", C.c_float), ("box_nms_thresh", C.c_float),
                ("detections_per_img", C.c_int), ("rpn_pre_nms_top_n", C.c_int), ("rpn_post_nms_top_n", C.c_int),
                ("rpn_nms_thresh", C.c_float), ("precision", C.c_int)]


PRECISION = {"fp32": 0, "f16x3": 1}


class View(C.Structure):
    _fields_ = [("image_dev", C.c_void_p), ("H", C.c_int), ("W", C.c_int), ("flip", C.c_int), ("nrect", C.c_int),
                ("rects", C.c_int * 16), ("noise_dev", C.c_void_p)]


class Dets(C.Structure):
    _fields_ = [("boxes_dev", C.c_void_p), ("scores_dev", C.c_void_p), ("labels_dev", C.c_void_p),
                ("props_dev", C.c_void_p), ("prob_max_dev", C.c_void_p), ("scores_cls_dev", C.c_void_p),
                ("count_dev", C.c_void_p), ("cap", C.c_int)]


MAX_AUGS = 32
N_MARGINS = 16          # CALD_N_MARGINS
MARGIN_NAMES = ['rpn_topk', 'rpn_iou', 'rpn_order', 'rpn_trunc', 'rpn_small', 'roi_level', 'roi_edge', 'post_thr', 'post_iou', 'post_order',
                'post_cap', 'ref_subsample', 'argmax', 'zero_row', 'cutout', 'reserved']
# Rounding noise of precision="f16x3" against the exact mode on each margin's scale, calibrated on 1 536 configs[1] images
# (tools/cascade_margins.py, profiles/r5_cascade_margins.txt): about the 90th percentile of |margin_exact - margin_f16x3|.
MARGIN_NOISE_F16X3 = [2e-5, 2e-6, 2e-5, 2e-5, 1e-5, 1e-6, 1e-4, 5e-6, 2e-6, 5e-6, 5e-6, 5e-6, 2e-6, 5e-6, 2e-6, 0.0]
AUG_FLIP, AUG_GAUSS, AUG_COLOR_ADJUST, AUG_COLOR_SWAP, AUG_SALT_PEPPER, AUG_CUTOUT, AUG_RESIZE, AUG_ROTATE = range(1, 9)


class AugSpec(C.Structure):
    _fields_ = [("kind", C.c_int), ("param", C.c_double)]


class PackJob(C.Structure):
    """cald_pack_job of include/cald_hip.h: the arguments of one cald_train_pack_conv call (device pointers)."""
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p),
                ("Cout", C.c_int), ("Cin", C.c_int), ("KH", C.c_int), ("KW", C.c_int), ("CinK", C.c_int), ("mode", C.c_int),
                ("packed", C.c_void_p)]


class SweepCfg(C.Structure):
    _fields_ = [("base_seed", C.c_uint64), ("bp", C.c_float), ("batch_images", C.c_int), ("n_augs", C.c_int),
                ("augs", AugSpec * MAX_AUGS)]


# name -> (restype, argtypes): must list every symbol of include/cald_hip.h
SIGNATURES = {
    "cald_last_error": (C.c_char_p, []),
    "cald_version": (C.c_int, []),
    "cald_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cald_ctx_destroy": (C.c_int, [C.c_void_p]),
    "cald_ctx_sync": (C.c_int, [C.c_void_p]),
    "cald_model_create": (C.c_int, [C.c_void_p, C.POINTER(ModelCfg), C.POINTER(C.c_void_p)]),
    "cald_model_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, c_f, c_i64, C.c_int]),
    "cald_model_finalize": (C.c_int, [C.c_void_p]),
    "cald_model_set_rpn_prune": (C.c_int, [C.c_void_p, C.c_int, c_i]),
    "cald_model_set_rpn_prune_capture": (C.c_int, [C.c_void_p, C.c_int]),
    "cald_model_rpn_prune_bound": (C.c_int, [C.c_void_p, c_f, c_f]),
    "cald_profile_prune_fallbacks": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "cald_model_destroy": (C.c_int, [C.c_void_p]),
    "cald_forward": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(View), C.POINTER(Dets)]),
    "cald_sweep": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_i, c_i, c_i64, C.POINTER(SweepCfg), c_d, c_d]),
    "cald_sweep_audit": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_i, c_i, c_i64, C.POINTER(SweepCfg), c_d, c_d, c_f]),
    "cald_sweep_ltc": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_i, c_i, C.c_int, c_d]),
    "cald_sweep_lsc": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_i, c_i, c_i64, C.c_uint64, C.c_int, c_d]),
    "cald_op_consistency": (C.c_int, [C.c_void_p, C.c_int, c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, C.c_int, C.c_float, c_f]),
    "cald_op_cls_corr": (C.c_int, [C.c_void_p, C.c_int, c_f, c_i64, C.c_int, c_f]),
    "cald_op_pil_resize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "cald_op_cutout_rects": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, c_f, C.c_int, c_i, c_i]),
    "cald_op_augment": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_int, c_f,
                                  C.c_void_p, c_f, c_i]),
    "cald_op_frcnn_postprocess": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                            C.c_int, c_f, c_f, c_i64, c_f, c_f, c_f, c_i]),
    "cald_op_roi_align": (C.c_int, [C.c_void_p, C.POINTER(c_f), c_i, C.c_int, C.c_int, c_f, c_f]),
    "cald_op_conv2d": (C.c_int, [C.c_void_p, c_f, C.c_int, C.c_int, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, c_f, c_f, c_f, c_f, C.c_int, c_f]),
    "cald_op_mfma_f16": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int64]),
    "cald_op_conv2d_f16x3": (C.c_int, [C.c_void_p, c_f, C.c_int, C.c_int, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, c_f, c_f, c_f, c_f, C.c_int, c_f]),
    "cald_op_conv_bench": (C.c_int, [C.c_void_p] + [C.c_int] * 12 + [c_d, c_d]),
    "cald_op_transform_size": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, c_i, c_i, c_i, c_i]),
    "cald_debug_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, c_f, C.c_int64, c_i64]),
    "cald_jpeg_info": (C.c_int, [C.c_void_p, C.c_size_t, c_i, c_i, c_i]),
    "cald_jpeg_decode_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]),
    "cald_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "cald_profile_read": (C.c_int, [C.c_void_p, c_d, c_d, c_i64, c_d]),
    "cald_profile_prune": (C.c_int, [C.c_void_p, c_d, c_d, c_d, c_d, c_d]),
    "cald_profile_roi_rows": (C.c_int, [C.c_void_p, c_d, c_i64]),
    "cald_profile_dump": (C.c_int, [C.c_void_p, C.c_char_p]),
    # training step (device pointers as c_void_p)
    "cald_train_packed_floats": (C.c_int, [C.c_int] * 6 + [c_i64]),
    "cald_train_pack_conv": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p]),
    "cald_train_pack_plan_scratch_floats": (C.c_int, [C.c_int, C.POINTER(PackJob), c_i64]),
    "cald_train_pack_plan_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(PackJob), C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "cald_train_pack_plan_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cald_train_pack_plan_destroy": (C.c_int, [C.c_void_p]),
    "cald_train_conv": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 8
                        + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "cald_train_conv_group": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_i, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)] + [C.c_int] * 8
                              + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int]),
    "cald_train_conv_wgrad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "cald_train_linear_wgrad": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_int]),
    "cald_train_relu_bwd": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cald_train_add": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cald_train_dilate": (C.c_int, [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p]),
    "cald_train_weave2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), c_i, c_i, C.c_void_p, C.c_void_p]),
    "cald_train_upsample_bwd": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]),
    "cald_train_rpn_proposals": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_i, C.POINTER(C.c_void_p), c_i, C.c_int, C.c_int, C.c_int,
                                           C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "cald_train_roi_sample_host": (C.c_int, [C.c_int, c_i, c_i, c_i, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i, c_i, c_i]),
    "cald_train_roi_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_void_p, C.c_void_p]),
    "cald_train_anchors": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_i, C.c_void_p]),
    "cald_train_match": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "cald_train_box_encode": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "cald_train_roi_align": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_i, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cald_train_roi_align_bwd": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_i, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cald_train_softmax_ce": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "cald_train_smooth_l1": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_float,
                                       C.c_void_p, C.c_void_p]),
    "cald_train_focal_loss": (C.c_int, [C.c_void_p, C.c_int, c_i, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "cald_train_bce_logits": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "cald_train_preprocess": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_i, C.c_int, C.c_int, C.c_void_p]),
    "cald_train_maxpool": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cald_train_subsample2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cald_train_sgd": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int]),
    "cald_train_seg_cache_size": (C.c_int, []),
    "cald_comm_unique_id": (C.c_int, [C.c_void_p]),
    "cald_comm_init_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cald_comm_adopt": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cald_comm_info": (C.c_int, [C.c_void_p, c_i, c_i]),
    "cald_comm_destroy": (C.c_int, [C.c_void_p]),
    "cald_allgather_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
}


def lib():
    """Loads libcaldhip.so.  torch is imported first so that both share one HIP runtime
    (torch's bundled libamdhip64.so.7 satisfies the library's NEEDED entry)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libcaldhip.so is not built (%s missing): run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` or `make -C cald_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        import torch  # noqa: F401  (loads the HIP runtime)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


ERR_UNSUPPORTED = -5


def check(rc):
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError("libcaldhip: %s (code %d)" % (lib().cald_last_error().decode(), rc))
    if rc != 0:
        raise RuntimeError("libcaldhip: %s (code %d)" % (lib().cald_last_error().decode(), rc))


def ptr(a, t=c_f):
    return a.ctypes.data_as(t) if a is not None else None
