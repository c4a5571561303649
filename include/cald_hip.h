/* cald_hip.h -- C ABI of libcaldhip.so: the MI355X (gfx950) implementation of CALD's unlabeled-pool
 * consistency sweep.  Plain pointers and sizes only; no torch types cross this boundary.
 *
 * The reference (we1pingyu/CALD) has no FFI: its boundary for this path is two in-process Python
 * call signatures.  Each entry point below names the reference interface it stands behind; the
 * Python shim that binds them (cald_amd/_ffi.py, ctypes) and the stub a maintainer would add to the
 * reference are shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, a negative code on failure; cald_last_error() gives the
 *     thread-local message (the Python shim raises RuntimeError with it).
 *   - pointers named *_dev are device (HBM) pointers on the context's GPU, everything else is host.
 *   - the caller owns and allocates all outputs; the library owns only ctx / model / workspace.
 *   - one context per device, one HIP stream per context, not thread-safe per context;
 *     one process per GPU for multi-GPU (the pool is sharded by the caller, see cald_amd/sweep.py).
 */
#ifndef CALD_HIP_H
#define CALD_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cald_ctx cald_ctx;
typedef struct cald_model cald_model;

#define CALD_OK 0
#define CALD_ERR_INVALID (-1)
#define CALD_ERR_HIP (-2)
#define CALD_ERR_STATE (-3)
#define CALD_ERR_MISSING_WEIGHT (-4)
#define CALD_ERR_UNSUPPORTED (-5)   /* input outside the supported set (e.g. progressive JPEG) */

/* arithmetic of the conv / linear GEMMs.
 *   FP32   exact: one k-ordered fp32 fma chain per output (v_mfma_f32_32x32x2_f32), bit-identical to the oracle.
 *   F16X3  the "fp16 MFMA path" of BASELINE.json configs[4]: operands split into fp16 hi + lo, three
 *          v_mfma_f32_32x32x16_f16 per product into fp32 accumulators (22-bit operands, ~1e-6 end to end): fp32-grade but
 *          NOT bit-identical -- ~1 % of images change through a flipped borderline detection, so the identical-top-k
 *          bar is met only by FP32; |activations| must stay below 4094. */
#define CALD_PRECISION_FP32 0
#define CALD_PRECISION_F16X3 1

#define CALD_ARCH_FRCNN 0      /* detection/frcnn_la.py FRCNN_Feature */
#define CALD_ARCH_RETINANET 1  /* detection/retinanet_cal.py RetinaNet */

const char* cald_last_error(void);
int cald_version(void);

/* stream: a hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream), or NULL for
 * a private stream.  Replaces torch.cuda.set_device(0) + default stream (cald_train.py:275). */
int cald_ctx_create(int device, void* stream, cald_ctx** out);
int cald_ctx_destroy(cald_ctx* ctx);
int cald_ctx_sync(cald_ctx* ctx);

/* Constructor arguments of fasterrcnn_resnet50_fpn_feature(num_classes, min_size, max_size)
 * (detection/frcnn_la.py:148-169, :278-289; call sites cald_train.py:340-347). */
typedef struct cald_model_cfg {
    int arch;               /* CALD_ARCH_* */
    int depth;              /* 50 | 101 */
    int num_classes;        /* incl. background for FRCNN */
    int min_size, max_size; /* 600/1000 VOC, 800/1333 COCO */
    float box_score_thresh; /* 0.05 */
    float box_nms_thresh;   /* 0.5 */
    int detections_per_img; /* 100 */
    int rpn_pre_nms_top_n;  /* 1000 */
    int rpn_post_nms_top_n; /* 1000 */
    float rpn_nms_thresh;   /* 0.7 */
    int precision;          /* CALD_PRECISION_* (0 = exact fp32, the default and the parity contract) */
} cald_model_cfg;

int cald_model_create(cald_ctx* ctx, const cald_model_cfg* cfg, cald_model** out);
/* model.load_state_dict() (cald_train.py:356): one call per tensor, torchvision key layout
 * ("backbone.body.layer1.0.conv1.weight", ...), float32 host data, row-major. */
int cald_model_load_tensor(cald_model* m, const char* key, const float* data, const int64_t* shape, int ndim);
/* folds FrozenBatchNorm into per-channel scale/shift, repacks weights K-major for the MFMA kernels */
int cald_model_finalize(cald_model* m);
/* Exact (CALD_PRECISION_FP32) Faster R-CNN sweeps evaluate the RPN head of P2 / P3 exactly only at the pixels that can hold one of a
 * level's rpn_pre_nms_top_n anchors, found by a split-fp16 look-ahead whose error bound is derived from a bit-exact statement of the matrix
 * instruction (oracle/mfma_f16_model.h, pinned to the hardware by the tests) and is checked at run time on every anchor evaluated both
 * ways (rpn_prune.hip): same detections bit for bit, ~1/8 less fp32 work.  A sweep whose data leave the look-ahead's
 * range (an activation of |x| >= 4094, a non-finite one) or exceed the bound is repeated with the dense head (cald_profile_prune_fallbacks
 * counts them).  On by default (environment CALD_RPN_PRUNE=0 turns it off for the process); this call switches it per model and returns
 * the previous state in *was (may be null).  cald_forward runs the dense head (unless capture mode is on, below). */
int cald_model_set_rpn_prune(cald_model* m, int on, int* was);
/* test hooks of the pruning: capture mode makes cald_forward take the pruned path too and keep the look-ahead's head maps as debug tensors
 * ("rpn_look0/1": [H][W][15], logits = channels 0..2; "rpn_pnorm0/1": the per-pixel |3 x 3 patch|_2; "rpn0/1": the maps the top-k reads, exact
 * at selected pixels and -FLT_MAX elsewhere); cald_model_rpn_prune_bound returns the constants of B_a(p) = c1[a] * pnorm(p) + c0[a], a < 3 */
int cald_model_set_rpn_prune_capture(cald_model* m, int on);
int cald_model_rpn_prune_bound(cald_model* m, float* c1, float* c0);
int cald_model_destroy(cald_model* m);

/* One detector input: task_model([tensor]) in cald_train.py:107 / :186.  The view is described by
 * its uint8 HWC source image in HBM plus the augmentation to apply on the fly. */
typedef struct cald_view {
    const uint8_t* image_dev; /* [H][W][3] uint8 */
    int H, W;
    int flip;                 /* cald_helper.HorizontalFlip */
    int nrect;                /* cald_helper.cutout rectangles (left, top, right, bottom), <= 4 */
    int rects[16];
    const float* noise_dev;   /* optional float32 [3][H][W] (CHW) added to image/255 before normalisation, or NULL.  With it a
                               * view carries ANY float input tensor exactly (the reference model accepts arbitrary floats,
                               * e.g. a caller-made GaussianNoise image): image = rounded uint8 grid, noise = x - image/255. */
} cald_view;

/* Result dict of the detector (detection/frcnn_la.py:131-141): device buffers, `cap` rows per view. */
typedef struct cald_dets {
    float* boxes_dev;      /* [n_views][cap][4] */
    float* scores_dev;     /* [n_views][cap] */
    int64_t* labels_dev;   /* [n_views][cap] */
    float* props_dev;      /* [n_views][cap][4] */
    float* prob_max_dev;   /* [n_views][cap] */
    float* scores_cls_dev; /* [n_views][cap][num_classes] */
    int32_t* count_dev;    /* [n_views] */
    int cap;
} cald_dets;

/* model(list_of_images) in eval mode, for up to 128 views at once (batch-1 semantics per view). */
int cald_forward(cald_model* m, int n_views, const cald_view* views, const cald_dets* out);

/* get_uncertainty(task_model, unlabeled_loader, augs, num_cls) (cald_train.py:91-231) over
 * n_images already resident in HBM.  consistency_out[n_images], cls_corr_out[n_images][num_classes-1].
 *
 * The augmented views of an image are given as an ordered list (the Python shim expands the reference's aug names
 * in the reference's order, cald_train.py:123-183).  Randomness: the reference draws GaussianNoise / SaltPepperNoise
 * from torch's global CPU generator and ColorSwap / cutout from Python's global `random`, in call order; here both
 * generators are re-seeded per image with base_seed * 1000003 + pool_pos[i] and consumed in list order, so results
 * do not depend on sharding or batching. */
#define CALD_AUG_FLIP 1          /* HorizontalFlip(image, boxes)            cald_train.py:123-126 */
#define CALD_AUG_GAUSS 2         /* GaussianNoise(image, std = param)       :127-135 ('ga' 16; 'multi_ga' 8..48) */
#define CALD_AUG_COLOR_ADJUST 3  /* ColorAdjust(image, factor = param)      :136-139 ('color_adjust' 1.5) */
#define CALD_AUG_COLOR_SWAP 4    /* ColorSwap(image)                        :140-143 */
#define CALD_AUG_SALT_PEPPER 5   /* SaltPepperNoise(image, prob = param)    :149-157 ('sp' 0.1; 'multi_sp' 0.05..0.3) */
#define CALD_AUG_CUTOUT 6        /* cutout(image, boxes, labels, cut_num = param) :158-166 ('cut_out' 2; 'multi_cut_out' 1..4) */
#define CALD_AUG_RESIZE 7        /* resize(image, boxes, ratio = param)     :167-179 ('smaller_resize' .8, 'larger_resize' 1.2, 'multi_resize' .7 .8 .9) */
#define CALD_AUG_ROTATE 8        /* rotate(image, boxes, angle = param)     :180-183 ('rotation' 5) */
#define CALD_MAX_AUGS 32
typedef struct cald_aug_spec {
    int kind;
    double param;          /* a Python float in the reference (e.g. 7 * 0.1 for multi_resize): kept in double */
} cald_aug_spec;
typedef struct cald_sweep_cfg {
    uint64_t base_seed;
    float bp;              /* args.bp, 1.3 */
    int batch_images;      /* images per batched launch sequence (0 = default 64; capped at CALD_MAX_VIEWS = 128) */
    int n_augs;
    cald_aug_spec augs[CALD_MAX_AUGS];
} cald_sweep_cfg;
int cald_sweep(cald_model* m, int n_images, const uint8_t* const* images_dev, const int* H, const int* W,
               const int64_t* pool_pos, const cald_sweep_cfg* cfg, double* consistency_out, double* cls_corr_out);

/* ---- cascade support: cald_sweep plus a record of how close every discrete decision of the path came to flipping ----
 * get_uncertainty's result is a continuous function of the detector's GEMM outputs except at its discrete decisions (RPN top-k /
 * NMS / post-NMS cut, RoI level, score threshold, class NMS, top-100, argmax over the IoU row, the linspace sub-sample, cutout's
 * accept test; cald_train.py:110-113, :158-166, :214, detection/frcnn_la.py:72-80, detection/frcnn_ll.py:284-321).  margins_out
 * [n_images][CALD_N_MARGINS] receives, per image and per kind of decision, the smallest distance of any decision THAT CAN REACH THE
 * OUTPUT to its flip point, over all views of the image (+inf where the kind did not occur).  An image whose margins all exceed the
 * rounding noise of a faster precision took the same decisions there as in CALD_PRECISION_FP32, and its scores differ by rounding
 * only; the others are re-scored exactly (cald_amd/sweep.py get_uncertainty_cascade, cald_cascade_plan).  Faster R-CNN only. */
#define CALD_N_MARGINS 16
#define CALD_MARGIN_RPN_TOPK 0       /* logit: k-th vs (k+1)-th objectness of a level's top-k cut */
#define CALD_MARGIN_RPN_IOU 1        /* IoU:   |max IoU with the kept boxes before it - rpn_nms_thresh| */
#define CALD_MARGIN_RPN_ORDER 2      /* logit: score gap of a (suppressor, suppressed) pair */
#define CALD_MARGIN_RPN_TRUNC 3      /* logit: post_nms_top_n-th vs next kept proposal */
#define CALD_MARGIN_RPN_SMALL 4      /* pixel: |w or h - 1e-3| of a clipped candidate */
#define CALD_MARGIN_ROI_LEVEL 5      /* log2:  distance of 4 + log2(sqrt(area) / 224) + 1e-6 to 3, 4 or 5 */
#define CALD_MARGIN_ROI_EDGE 6       /* feature pixel: distance of a RoIAlign sample to -1 or to the map size */
#define CALD_MARGIN_POST_THR 7       /* prob:  |class score - box_score_thresh| of a box NMS would keep */
#define CALD_MARGIN_POST_IOU 8       /* IoU:   |max IoU with kept same-class detections before it - box_nms_thresh| */
#define CALD_MARGIN_POST_ORDER 9     /* prob:  score gap of a (suppressor, suppressed) pair */
#define CALD_MARGIN_POST_CAP 10      /* prob:  detections_per_img-th detection vs the best candidate behind it */
#define CALD_MARGIN_REF_SUBSAMPLE 11 /* prob:  score gap of neighbours of which np.round(np.linspace(0, n - 1, 50)) picks one (n > 40) */
#define CALD_MARGIN_ARGMAX 12        /* IoU:   best vs best-of-another-proposal in a reference box's IoU row */
#define CALD_MARGIN_ZERO_ROW 13      /* prob:  score gap of an augmented view's first two detections when a row is all zero */
#define CALD_MARGIN_CUTOUT 14        /* ratio: |largest overlap ratio - 0.4 or 0.1| of a cutout trial */
int cald_sweep_audit(cald_model* m, int n_images, const uint8_t* const* images_dev, const int* H, const int* W,
                     const int64_t* pool_pos, const cald_sweep_cfg* cfg, double* consistency_out, double* cls_corr_out,
                     float* margins_out);

/* ---- the baseline sweeps of the same repo that share the detector forward ---- */
/* lt_c_train.py:105-121 get_uncertainty(task_model, unlabeled_loader) -> one float per image */
int cald_sweep_ltc(cald_model* m, int n_images, const uint8_t* const* images_dev, const int* H, const int* W,
                   int batch_images, double* uncertainty_out);
/* ls_c_train.py:108-155 get_uncertainty(task_model, unlabeled_loader) -> one float per image (six GaussianNoise views) */
int cald_sweep_lsc(cald_model* m, int n_images, const uint8_t* const* images_dev, const int* H, const int* W,
                   const int64_t* pool_pos, uint64_t base_seed, int batch_images, double* stability_out);

/* ---- operator-level entry points (used by the parity tests; same kernels as the paths above) ---- */
/* scoring of ONE (reference, augmentation) pair, cald_train.py:189-224 */
int cald_op_consistency(cald_ctx* ctx, int N, const float* aug_box, const float* ref_scores_cls, const float* ref_pm,
                        int M, const float* boxes, const float* scores_cls, const float* pm, int C, float bp,
                        float* consistency_out);
/* cald_train.py:114-117 */
int cald_op_cls_corr(cald_ctx* ctx, int n, const float* scores, const int64_t* labels, int C, float* out);
/* cald_helper.resize: PIL.Image.resize((ow, oh), BILINEAR) on uint8 RGB */
int cald_op_pil_resize(cald_ctx* ctx, const uint8_t* src_dev, int H, int W, uint8_t* dst_dev, int oh, int ow);
/* cald_helper.cutout rectangle selection (host side RNG = Python random seeded per image) */
int cald_op_cutout_rects(uint64_t seed, int H, int W, int N, const float* boxes, int cut_num, int* rects_out, int* n_out);
/* one augmented view outside the sweep (helper API of cald/cald_helper.py; same kernels as inside cald_sweep).  A fresh
 * generator is seeded with `seed` (torch's CPU generator for GAUSS / SALT_PEPPER, Python's `random` for COLOR_SWAP).
 *   CALD_AUG_GAUSS        GaussianNoise :72-75    dst_dev = float [3][H][W], the additive term randn * param / 255
 *   CALD_AUG_SALT_PEPPER  SaltPepperNoise :78-85  dst_dev = uint8 [H][W][3]
 *   CALD_AUG_COLOR_ADJUST ColorAdjust :65-69      dst_dev = uint8 [H][W][3]
 *   CALD_AUG_COLOR_SWAP   ColorSwap :56-62        aux_out[0] = index of the drawn channel permutation (no device work)
 *   CALD_AUG_ROTATE       rotate :135-223         dst_dev = uint8 [H][W][3], boxes_out[n_boxes][4] (host) */
int cald_op_augment(cald_ctx* ctx, int kind, double param, uint64_t seed, const uint8_t* src_dev, int H, int W,
                    int n_boxes, const float* boxes, void* dst_dev, float* boxes_out, int* aux_out);
/* RoIHeads.postprocess_detections + GeneralizedRCNNTransform.postprocess of one view (detection/frcnn_la.py:32-87, :292-315) on the
 * forward's own kernels: logits [R][C], deltas [R][4C] (class-major), proposals [R][4] in resized-image coordinates (host); outputs
 * (host, det_max rows each; scores_cls [det_max][C]) and the number of detections.  R <= 1000. */
int cald_op_frcnn_postprocess(cald_ctx* ctx, int R, int C, const float* logits, const float* deltas, const float* proposals,
                              int Hr, int Wr, int Ho, int Wo, float score_thr, float nms_thr, int det_max,
                              float* boxes_out, float* scores_out, int64_t* labels_out, float* props_out, float* prob_max_out,
                              float* scores_cls_out, int* n_out);
/* MultiScaleRoIAlign(output 7, sampling_ratio 2, aligned=False; detection/frcnn_la.py:205-209) of one view on the forward's own
 * kernels: feats[l] = [H_l][W_l][C] (host) for P2..P5, level_hw = {H0, W0, ..., H3, W3}, rois [R][4]; out [R][49][C] (host) */
int cald_op_roi_align(cald_ctx* ctx, const float* const* feats, const int* level_hw, int C, int R, const float* rois, float* out);
/* NHWC convolution on the MFMA kernel; weights in torch layout [Cout][Cin][KH][KW] (host) */
int cald_op_conv2d(cald_ctx* ctx, const float* in, int H, int W, int Cin, const float* weight, int Cout, int KH, int KW,
                   int stride, int pad, const float* bias, const float* bn_scale, const float* bn_shift,
                   const float* residual, int relu, float* out);
/* the same convolution in CALD_PRECISION_F16X3 (conv_h3.hip); falls back to the exact kernels for shapes it does not
 * cover (Cin % 16 != 0 or Cout not tiled by 128), exactly as inside a model */
int cald_op_conv2d_f16x3(cald_ctx* ctx, const float* in, int H, int W, int Cin, const float* weight, int Cout, int KH, int KW,
                         int stride, int pad, const float* bias, const float* bn_scale, const float* bn_shift,
                         const float* residual, int relu, float* out);
/* Parity hook of CALD_PRECISION_F16X3's arithmetic primitive: n independent dot products D[i] = C[i] + sum_{k<16} A[i][k] B[i][k], each
 * evaluated by the hardware as ONE output element of v_mfma_f32_32x32x16_f16 (the instruction conv_h3.hip / conv_h4.hip are built on;
 * there is no reference function -- the reference has no fp16 path).  A, B: fp16 bit patterns [n][16], C / D: fp32
 * bit patterns [n] (host).  tests/ hold oracle/mfma_f16_model.h, the CPU statement of that instruction, against it bit for bit. */
int cald_op_mfma_f16(cald_ctx* ctx, const uint16_t* A, const uint16_t* B, const uint32_t* C, uint32_t* D, int64_t n);
/* kernel-tuning aid (tools/bench_conv.py): average time of ONE conv layer shape (the model's own kernel selection) over a
 * ragged batch of n_views equal views filled with pseudo-random data; `group` > 1 issues that many independent copies as one
 * grouped launch (FPN / RPN style).  tflops_out counts algorithmic FLOPs (2 * M * Cout * KH*KW*Cin).  relu: bit 0 = ReLU in the
 * epilogue. */
int cald_op_conv_bench(cald_ctx* ctx, int n_views, int H, int W, int Cin, int Cout, int KH, int stride, int pad, int residual,
                       int relu, int iters, int group, double* ms_out, double* tflops_out);
/* detector-transform size (GeneralizedRCNNTransform): resized and padded sizes */
int cald_op_transform_size(int H, int W, int min_size, int max_size, int* Hr, int* Wr, int* Hp, int* Wp);
/* intermediate tensors of the LAST cald_forward (parity debugging): name in
 * {"input","conv1","pool1","C2".."C5","P2".."P6","rpn0".."rpn4","proposals","roi","fc6","fc7","pred"} */
int cald_debug_tensor(cald_model* m, const char* name, int view, float* host_out, int64_t capacity, int64_t* shape3);

/* ---- input side: PIL.Image.open(path).convert('RGB') of torchvision's VOCDetection /
 * CocoDetection __getitem__ (detection/voc_utils.py:47-58, detection/coco_utils.py; DataLoader at cald_train.py:434).
 * Bit-identical to Pillow / libjpeg-turbo defaults (ISLOW IDCT, fancy upsampling) for 8-bit baseline Huffman JPEGs:
 * grayscale or YCbCr 4:4:4 / 4:2:2 / 4:2:0, one interleaved scan, restart intervals allowed.  Other flavours
 * (progressive, CMYK, ...) return CALD_ERR_UNSUPPORTED -- nothing is decoded on the CPU. ---- */
/* host-only header parse: image size and component count */
int cald_jpeg_info(const uint8_t* data, size_t size, int* H, int* W, int* ncomp);
/* decodes n JPEG files (host bytes) into caller-allocated device images out_dev[i] = uint8 [H][W][3] (RGB) */
int cald_jpeg_decode_batch(cald_ctx* ctx, int n, const uint8_t* const* data, const size_t* sizes, uint8_t* const* out_dev);

/* ---- measurement: HIP-event timing of every conv/linear launch on the context stream ---- */
int cald_profile_enable(cald_ctx* ctx, int on);
/* gemm_flops = algorithmic FLOPs of the timed launches; the RoI-head layers (fc6 / fc7 / predictor) are counted on the
 * MEASURED proposal rows (device-side count after RPN NMS), not on the row capacity */
int cald_profile_read(cald_ctx* ctx, double* gemm_ms, double* gemm_flops, int64_t* gemm_launches, double* total_ms);
/* the exact sweep's certified RPN pruning: time and algorithmic FLOPs of its split-fp16 look-ahead launches
 * (NOT part of cald_profile_read's figures, which then count the exact kernels only, the gathered launches on their selected rows) and
 * the fraction of P2 / P3 pixels whose head was recomputed exactly; worst_bound_ratio = the largest |look-ahead - exact| / bound any sweep of
 * this context has observed on the anchors evaluated both ways (every sweep checks it and repeats itself with the dense head if it exceeds 1);
 * pruned_flops = the dense head's exact FLOPs that were not executed */
int cald_profile_prune(cald_ctx* ctx, double* lookahead_ms, double* lookahead_flops, double* selected_fraction_p2_p3, double* worst_bound_ratio,
                       double* pruned_flops);
/* sweeps of this context that were repeated with the dense head (bound exceeded, or an activation outside the look-ahead's range) */
int cald_profile_prune_fallbacks(cald_ctx* ctx, int64_t* n);
/* mean proposals per view (R of SURVEY 8d) over the Faster R-CNN forwards profiled since cald_profile_enable */
int cald_profile_roi_rows(cald_ctx* ctx, double* mean_rows_per_view, int64_t* views);
/* per-launch CSV (shape, algorithmic GFLOP, ms, TFLOP/s) of the launches recorded since cald_profile_enable */
int cald_profile_dump(cald_ctx* ctx, const char* path);

/* ---- training step: the device operators behind task_model(images, targets) / losses.backward() /
 * optimizer.step() of cald_train.py:40-74 (detection/engine.py:19-61), which the reference delegates to torchvision 0.8.2 +
 * cuDNN autograd.  cald_amd/train.py strings them into the Faster R-CNN training graph.  All pointers are DEVICE pointers,
 * activations are dense NHWC batches [N][H][W][C], every call is asynchronous on the context stream EXCEPT cald_train_anchors and
 * cald_train_rpn_proposals, which stage small host tables (base anchors, proposal counts) and synchronise the context stream before
 * they return (cald_train_preprocess sends its view descriptors through a pinned staging ring and does not wait).
 * Threading: the cald_train_* entry points share one process-wide cache of batch-geometry tables; they are NOT thread-safe, not even across
 * contexts -- call them from one thread per process (one process per GPU is the model everywhere in this library). ---- */
/* size in floats of the packed form of a torch-layout weight [Cout][Cin][KH][KW] (see cald_train_pack_conv) */
int cald_train_packed_floats(int Cout, int Cin, int KH, int KW, int CinK, int mode, int64_t* floats_out);
/* packs weight (+ optional bias / FrozenBatchNorm scale, shift: [Cout]) for the MFMA conv kernels, on the device.
 *   mode 0  forward over an input whose channel stride is CinK >= Cin (multiple of 4; extra channels must be zero or finite)
 *   mode 1  data gradient: the flipped, transposed filter applied to dY with channel stride CinK >= Cout; bn_scale (or null) is
 *           folded in (row co times bn_scale[co]): dY is then the gradient wrt the FrozenBatchNorm output
 *   mode 2  linear layer on rows laid out [tap][Cin] whose torch weight is [Cout][Cin * taps] (box_head.fc6 on RoIAlign rows)
 *   mode 3  data gradient of a mode-2 layer (dY rows with channel stride CinK >= Cout -> rows laid out [tap][Cin]) */
int cald_train_pack_conv(cald_ctx* ctx, const float* weight, const float* bias, const float* bn_scale, const float* bn_shift,
                         int Cout, int Cin, int KH, int KW, int CinK, int mode, float* packed);
/* Every trainable layer's packs in two launches: the optimizer step changes all weights at once (cald_train.py:62-64), and one
 * cald_train_pack_conv per layer and form is ~220 launch-bound launches at the start of every step.  A plan records the jobs (the
 * arguments of cald_train_pack_conv, device pointers that stay valid for the plan's lifetime); cald_train_pack_plan_run packs them
 * all on the context stream, bit for bit what the per-layer calls write.  scratch: device memory of
 * cald_train_pack_plan_scratch_floats floats owned by the plan until it is destroyed (zeroed by create on ctx's stream: run the
 * plan on that stream, or on one ordered after it). */
typedef struct cald_pack_job {
    const float *weight, *bias, *bn_scale, *bn_shift;
    int Cout, Cin, KH, KW, CinK, mode;
    float* packed;
} cald_pack_job;
typedef struct cald_pack_plan cald_pack_plan;
int cald_train_pack_plan_scratch_floats(int n, const cald_pack_job* jobs, int64_t* floats_out);
int cald_train_pack_plan_create(cald_ctx* ctx, int n, const cald_pack_job* jobs, float* scratch, int64_t scratch_floats, cald_pack_plan** plan_out);
int cald_train_pack_plan_run(cald_ctx* ctx, const cald_pack_plan* plan);
int cald_train_pack_plan_destroy(cald_pack_plan* plan);
/* out[N][Ho][Wo][out_ld] = epilogue(conv(in[N][H][W][CinK], packed)); flags: 1 bias, 2 scale/shift, 4 ReLU; residual (same
 * shape as out) and up ([N][Hup][Wup][Cout], nearest-upsampled) are added before the ReLU.  With mode 1 the call computes the
 * data gradient of a stride-1 conv (pad = K - 1 - forward pad); Cout / Cin are always those of the FORWARD weight.  mask (or
 * null; same shape as out): out = mask > 0 ? out : 0 at the very end -- the ReLU backward of the layer whose saved output it is. */
int cald_train_conv(cald_ctx* ctx, int N, int H, int W, const float* in, int CinK, const float* packed, int Cout, int Cin,
                    int KH, int KW, int stride, int pad, int mode, int flags, const float* residual, const float* up,
                    int Hup, int Wup, const float* mask, float* out, int out_ld);
/* cald_train_conv of ONE layer shape on n tensors in one launch (pyramid levels under shared-weight heads; packed[i] may differ per
 * problem as long as the shape is the same).  hw = {H_0, W_0, ...}; no residual / upsample inputs.  n <= 10. */
int cald_train_conv_group(cald_ctx* ctx, int n, int N, const int* hw, const float* const* ins, int CinK, const float* const* packed,
                          int Cout, int Cin, int KH, int KW, int stride, int pad, int mode, int flags, const float* const* masks,
                          float* const* outs, int out_ld);
/* dw[Cout][Cin][KH][KW] (=, or += when accumulate) sum over output pixels of g[q][co] * x[q @ tap][ci]; db[Cout] likewise (or
 * null).  x [N][H][W][ldx], g [N][Ho][Wo][ldg]; Cin, ldx, ldg multiples of 4.  Deterministic (fixed-order split reduction).
 * row_scale (or null): dw[co] is multiplied by row_scale[co] -- g is then the gradient wrt the FrozenBatchNorm OUTPUT of the layer. */
int cald_train_conv_wgrad(cald_ctx* ctx, int N, int H, int W, const float* x, int Cin, int ldx, const float* g, int Cout, int ldg,
                          int KH, int KW, int stride, int pad, const float* row_scale, float* dw, float* db, int accumulate);
/* linear layer: x [R][K], g [R][ldg] -> dw [Cout][K]; taps > 1: x rows are [tap][K / taps], dw is [Cout][K / taps][taps] */
int cald_train_linear_wgrad(cald_ctx* ctx, int R, const float* x, int K, const float* g, int Cout, int ldg, int taps,
                            float* dw, float* db, int accumulate);
/* g = (act > 0 ? g : 0) * scale[c]  (ReLU backward on the layer's output + FrozenBatchNorm scale); act / scale may be null */
int cald_train_relu_bwd(cald_ctx* ctx, long long rows, int C, float* g, const float* act, const float* scale);
/* dst = a + b (b null: copy); n floats, multiple of 4 */
int cald_train_add(cald_ctx* ctx, long long n, float* dst, const float* a, const float* b);
/* scatter g [N][Ho][Wo][C] onto the stride-1 grid out [N][Hd][Wd][C] (zeros elsewhere): first step of a strided data gradient */
int cald_train_dilate(cald_ctx* ctx, int N, int Ho, int Wo, int C, int s, int Hd, int Wd, const float* g, float* out);
/* last step of the stride-2 3x3 data gradient computed as four phase convolutions on the un-dilated dY (output pixels (2m + a, 2n + b)
 * use disjoint filter taps): weaves phases[2a + b] = [N][phase_hw[2k]][phase_hw[2k+1]][C] (pixel (m + phase_off[k], n + phase_off[k]) ->
 * (2m + a, 2n + b)) into out [N][H][W][C]; mask (or null, same shape): out = mask > 0 ? out : 0 */
int cald_train_weave2(cald_ctx* ctx, int N, int H, int W, int C, const float* const* phases, const int* phase_hw, const int* phase_off,
                      const float* mask, float* out);
/* FPN top-down backward: coarse += sum of the fine pixels that nearest-upsampling reads from each coarse pixel */
int cald_train_upsample_bwd(cald_ctx* ctx, int N, int Hf, int Wf, int Hc, int Wc, int C, const float* fine, float* coarse);
/* RegionProposalNetwork.filter_proposals at training sizes (pre / post_nms_top_n <= 2048): heads[l] = [N][Hl][Wl][head_ld] with
 * channel a = objectness of anchor a (A = 3), channel 3 + 4a + j = delta j; level_hw = {H0, W0, ..., H4, W4}; image_sizes = host
 * [N][2] resized (h, w).  proposals_out [N][post_n][4] and counts_out [N] are device buffers. */
int cald_train_rpn_proposals(cald_ctx* ctx, int N, int Hp, int Wp, const int* image_sizes, const float* const* heads,
                             const int* level_hw, int head_ld, int pre_n, int post_n, float nms_thr, float min_size,
                             float* proposals_out, int* counts_out);
/* roi_heads.select_training_samples on the HOST for a whole batch (labels from Matcher results, BalancedPositiveNegativeSampler,
 * the index lists of the loss kernels): torchvision 0.8.2's RoIHeads as frcnn_la.py:198-222 builds it.  All pointers are HOST
 * pointers.  Candidate table: per image slots[i] proposal rows (the first counts[i] used; counts null = all) then n_gt[i]
 * ground-truth rows, images back to back; matched = cald_train_match values per table row; gt_labels = the images' labels back to
 * back; keys = one iid uniform draw per table row (the k smallest of a class are kept: every k-subset equally likely).  Outputs
 * (capacity N * batch): table row, ground-truth row (sum n_gt = the extra zero box of images without boxes), label, image index per
 * sampled RoI in (image, table row) order; pos_rows / pred_idx (row * pred_ld + num_classes + 4 * label) of the foreground RoIs;
 * per_image_out[N] (may be null) = RoIs sampled per image. */
int cald_train_roi_sample_host(int N, const int* slots, const int* n_gt, const int* counts, const int32_t* matched,
                               const int64_t* gt_labels, const double* keys, int batch, double pos_fraction, int pred_ld, int num_classes,
                               int64_t* keep_rows, int64_t* gt_sel, int64_t* labels_out, float* img_col, int64_t* pos_rows,
                               int64_t* pred_idx, int* R_out, int* n_pos_out, int* per_image_out);
/* The device side of the RoI sampling in one launch: idx = cald_train_roi_sample_host's six output lists uploaded with stride cap
 * (table rows | ground-truth rows | labels | pred_idx | pos_rows | image index as float32); table = the candidate boxes [rows][4],
 * gts = all ground-truth boxes + one zero box.  rois_out [R][5] (image index, box) for cald_train_roi_align; box_tgt_out [n_pos][4] =
 * BoxCoder.encode of the foreground rows (weights wx..wh), the arithmetic of cald_train_box_encode. */
int cald_train_roi_gather(cald_ctx* ctx, const float* table, const float* gts, const int64_t* idx, int cap, int R, int n_pos,
                          float wx, float wy, float ww, float wh, float* rois_out, float* box_tgt_out);
/* AnchorGenerator: all anchors of one padded image over five levels, order (level, y, x, anchor): anchors_out [sum Hl*Wl*A][4].
 * kind 0 = Faster R-CNN (A = 3, frcnn_la.py:185-187), kind 1 = RetinaNet (A = 9, retinanet_cal.py:346-351) */
int cald_train_anchors(cald_ctx* ctx, int kind, int Hp, int Wp, const int* level_hw, float* anchors_out);
/* det_utils.Matcher: matched_out[i] = index of the best ground-truth box (torchvision.ops.box_iou), -1 if its IoU < lo, -2 if
 * lo <= IoU < hi; allow_low_quality restores every box that is some ground truth's best.  best_iou_out may be null. */
int cald_train_match(cald_ctx* ctx, int n_boxes, const float* boxes, int n_gt, const float* gt, float hi, float lo,
                     int allow_low_quality, int* matched_out, float* best_iou_out);
/* det_utils.BoxCoder.encode_single */
int cald_train_box_encode(cald_ctx* ctx, int n, const float* reference, const float* proposals, float wx, float wy, float ww,
                          float wh, float* out);
/* MultiScaleRoIAlign(7, sampling_ratio 2) over four dense levels feats[l] = [N][Hl][Wl][C]; rois [R][5] = (image, x1, y1, x2,
 * y2); out [R][49][C].  _bwd scatters gout into gfeats[l] = [N][Hl][Wl][C] (+=): deterministic -- the contributions are summed as
 * 64-bit fixed-point integers (scale from max|gout|), so the arrival order of the atomics does not matter. */
int cald_train_roi_align(cald_ctx* ctx, const float* const* feats, const int* level_hw, int C, int R, const float* rois, float* out);
int cald_train_roi_align_bwd(cald_ctx* ctx, int N, float* const* gfeats, const int* level_hw, int C, int R, const float* rois,
                             const float* gout);
/* F.cross_entropy over R rows of stride ld (mean); grad_out (same layout, may be null) = gscale * d loss / d logits */
int cald_train_softmax_ce(cald_ctx* ctx, int R, int C, int ld, const float* logits, const int64_t* labels, float gscale,
                          float* loss_out, float* grad_out);
/* det_utils.smooth_l1_loss(size_average=False) / denom over n 4-vectors starting at float offsets idx[i] of pred; beta = 0 is the L1
 * loss; weights (one per 4-vector, or null) scale each term; grad (zeroed by the caller, same offsets) may be null */
int cald_train_smooth_l1(cald_ctx* ctx, int n, const float* pred, const int64_t* idx, const float* target, float beta, float denom,
                         const float* weights, float gscale, float* loss_out, float* grad);
/* RetinaNet classification loss (retinanet_cal.py:100-133): sigmoid focal loss (gamma 2) summed over the anchors not between the
 * matcher thresholds, weighted per image by img_weight[n] = 1 / (max(1, #foreground) * N).  logits / grad: five level blocks
 * [N][level_pix[l]][ld] back to back, channel a * K + k; matched [N][sum level_pix * A] (cald_train_match values);
 * gt_labels[gt_off[n] + m] = class index (0-based logit column) of image n's ground-truth box m. */
int cald_train_focal_loss(cald_ctx* ctx, int N, const int* level_pix, int A, int K, int ld, const float* logits, const int* matched,
                          const int64_t* gt_labels, const int* gt_off, const float* img_weight, float alpha, float gscale,
                          float* loss_out, float* grad);
/* F.binary_cross_entropy_with_logits over n logits at float offsets idx[i] (mean) */
int cald_train_bce_logits(cald_ctx* ctx, int n, const float* logits, const int64_t* idx, const float* labels, float gscale,
                          float* loss_out, float* grad);
/* GeneralizedRCNNTransform of a training batch: images[i] = device uint8 [H_i][W_i][3]; hw = host {H_i, W_i, Hr_i, Wr_i} per image
 * (source and resized sizes); remainders[i] (or null) = device float [3][H_i][W_i] added to image / 255.  out [N][Hp][Wp][4]:
 * normalized, bilinearly resized, zero-padded; channel 3 is zero. */
int cald_train_preprocess(cald_ctx* ctx, int N, const uint8_t* const* images, const float* const* remainders, const int* hw,
                          int Hp, int Wp, float* out);
/* max_pool2d(3, 2, 1) and max_pool2d(1, 2, 0) on [N][H][W][C] */
int cald_train_maxpool(cald_ctx* ctx, int N, int H, int W, int C, const float* in, float* out);
int cald_train_subsample2(cald_ctx* ctx, int N, int H, int W, int C, const float* in, float* out);
/* torch.optim.SGD step over a flat buffer: d = grad + wd * p; buf = first_step ? d : momentum * buf + d; p -= lr * buf */
int cald_train_sgd(cald_ctx* ctx, long long n, float* param, const float* grad, float* momentum_buf, float lr, float momentum,
                   float weight_decay, int first_step);
/* ---- multi-GPU: the sweep's one collective.  Replaces detection/utils.py:75-115 (`all_gather`: pickle, pad to the longest
 * rank, gather) and :302-324 (`init_distributed_mode`, NCCL process group).  One process per GPU; RCCL is bound at first use
 * (dlopen librccl.so.1), the single-GPU path never touches it.
 *   rank 0:      cald_comm_unique_id(id)  -> ship the 128 bytes to every rank by any means (file, socket, the host's own launcher)
 *   every rank:  cald_comm_init_rank(ctx, id, world, rank, &comm)       (or cald_comm_adopt() around an ncclComm_t the host owns)
 *   every rank:  cald_allgather_scores(comm, send_dev, recv_dev, rows_per_rank, row_len)   -- rows of float64, device buffers;
 *                recv_dev [world * rows_per_rank][row_len], rank r's rows at r * rows_per_rank.  Asynchronous on the context's stream.
 * With the strided shard (rank r scores pool positions p % world == r, padded to ceil(pool / world) rows) row j of rank r IS pool
 * position r + j * world: no index column and no padding protocol travel. */
typedef struct cald_comm cald_comm;
int cald_comm_unique_id(void* id128_out);
int cald_comm_init_rank(cald_ctx* ctx, const void* id128, int world_size, int rank, cald_comm** out);
int cald_comm_adopt(cald_ctx* ctx, void* rccl_comm, cald_comm** out);
int cald_comm_info(const cald_comm* comm, int* world_size, int* rank);
int cald_comm_destroy(cald_comm* comm);
int cald_allgather_scores(cald_comm* comm, const double* send_dev, double* recv_dev, int64_t rows_per_rank, int row_len);

/* Number of dense-batch geometry tables the training operators keep cached on the device (all contexts).  The cache is bounded
 * (1024 tables, CALD_SEG_CACHE_CAP overrides); it is emptied between operator calls, never inside one.  Diagnostic. */
int cald_train_seg_cache_size(void);

#ifdef __cplusplus
}
#endif
#endif
