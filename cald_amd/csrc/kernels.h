// kernels.h -- launcher prototypes and device-visible descriptor structs (product code).
#pragma once
#include "common.h"

struct ViewDesc {
    const uint8_t* src;   // device uint8 HWC (original image, or the PIL-resized one)
    int H, W;             // source size
    int flip;             // HorizontalFlip view
    int nrect;            // cutout rectangles (left, top, right, bottom)
    int rects[4 * CALD_MAX_CUT];
    int Hr, Wr;           // detector-transform resized size (image_sizes)
    int Ho, Wo;           // size the detections are scaled back to (= H, W of this view's source)
    const float* noise;   // optional additive noise, CHW float32 (GaussianNoise view), else null
    int swap;             // ColorSwap view: index into cald_helper.ColorSwap's permutation table (0 = identity)
};

// detections of one view, fixed capacity det_cap rows (frcnn_la.py:131-141 result dict)
struct DetBuffers {
    float* boxes;       // [V][cap][4]
    float* scores;      // [V][cap]
    long long* labels;  // [V][cap]
    float* props;       // [V][cap][4]
    float* prob_max;    // [V][cap]
    float* scores_cls;  // [V][cap][C]
    int* count;         // [V]
    int cap;
    int C;
};

// elementwise.hip
void launch_preprocess(const ViewDesc* views, const LevelSeg* seg0, float* out, int V, int max_pix, hipStream_t st);
void launch_pil_horizontal(const uint8_t* src, int H, int W, uint8_t* dst, int ow, const int* bounds, const int* kk, int ksize, hipStream_t st);
void launch_pil_vertical(const uint8_t* src, int H, int W, uint8_t* dst, int oh, const int* bounds, const int* kk, int ksize, hipStream_t st);
void launch_maxpool(const float* in, float* out, const LevelSeg* sin, const LevelSeg* sout, int C, int V, int max_out_pix, hipStream_t st,
                    bool out16 = false);   // out16: store the split form of conv_h3.hip (same buffer, one word per element)
void launch_subsample2(const float* in, float* out, const LevelSeg* sin, const LevelSeg* sout, int C, int V, int max_out_pix, hipStream_t st);

void launch_affine_nearest(const uint8_t* src, int H, int W, uint8_t* dst, int oh, int ow, const int* a, hipStream_t st);
// all torch-generator views of one image, in draw order: kind 0 = randn * p0 / 255 -> float dst[3*H*W] (CHW),
// kind 1 = rand -> salt (u < p0) / pepper (u > p1) on src -> uint8 dst[H*W*3] (HWC)
#define CALD_MAX_NOISE_SEG 16
struct NoiseSeg { int kind; float p0, p1; void* dst; };
struct NoiseJob { unsigned long long seed; const uint8_t* src; int H, W, nseg; NoiseSeg seg[CALD_MAX_NOISE_SEG]; };
void launch_noise_stream(const NoiseJob* jobs, int n, hipStream_t st);
// ColorAdjust (PIL ImageEnhance Brightness -> Contrast -> Color); tmp: H*W*3 bytes, lsum: one u64
void launch_color_adjust(const uint8_t* src, int H, int W, float factor, uint8_t* tmp, unsigned long long* lsum, uint8_t* dst,
                         hipStream_t st);

// rpn.hip
struct RpnArgs {
    const float* head[5];        // per level: [sum pix][head_ld], ch a = logit, ch A+4a+j = delta
    const LevelSeg* seg[5];      // level geometry (levels 2..6 of the plan)
    const LevelSeg* seg0;        // padded input geometry (for anchor strides)
    const ViewDesc* views;       // Hr, Wr for clipping
    const float* base_anchors;   // [5][A][4]
    int head_ld, A, V;
    int pre_n, post_n;
    float nms_thr, min_size;
    unsigned long long* cand_key;  // [V][5*pre_n]
    float* cand_box;               // [V][5*pre_n][4]
    float* sorted_box;             // [V][5*pre_n][4]  (score order, level offset applied)
    float* sorted_raw;             // [V][5*pre_n][4]  (score order, no offset)
    int* sorted_count;             // [V]
    float* proposals;              // [V][prop_stride][4]
    int prop_stride;               // rows per view in `proposals` (>= post_n)
    int* prop_count;               // [V]
    // decision-margin audit (audit.hip), both null in a plain forward:
    unsigned long long* next_key = nullptr;  // [V][5][2]: per level the k-th selected key and the best key left out by the top-k (0 = level has <= k anchors)
    unsigned long long* trunc_key = nullptr; // [V][2]: the post_n-th and (post_n + 1)-th key of the merged post-NMS order (0 = fewer kept)
};
void launch_rpn(const RpnArgs& a, hipStream_t st);

// roi.hip
struct RoiArgs {
    const float* feat[4];
    const LevelSeg* seg[4];
    int C, V;
    const float* proposals;  // [V][ROI_CAP][4]
    const int* prop_count;   // [V]
    float* out;              // [V][ROI_CAP][49][C]
    int* order;              // [V][1024] scratch: the view's RoIs sorted by (pyramid level, row band, column) -- processing order only
    int out16;               // CALD_PRECISION_F16X3: store the split form conv_h3.hip consumes (one word per element) instead of fp32
};
void launch_roi_align(const RoiArgs& a, hipStream_t st);

// MultiScaleRoIAlign pieces shared by the inference kernel (roi.hip) and the training kernels (train.hip)
__device__ inline int roi_level(const float4 b) {
    const float area = (b.z - b.x) * (b.w - b.y);
    const float s = sqrtf(area);
    float k = floorf((4.0f + det_log2f(s / 224.0f)) + 1e-6f);
    if (!(k >= 2.0f)) k = 2.0f;
    if (k > 5.0f) k = 5.0f;
    return (int)k - 2;
}

struct RoiSample { int lo, hi; float l, h; int valid; };
__device__ inline RoiSample roi_sample(float start, float bin, int p, int i, int size) {
    RoiSample s;
    const float t = start + (float)p * bin + ((float)i + 0.5f) * bin / 2.0f;
    s.valid = !(t < -1.0f || t > (float)size);
    float tt = t <= 0.0f ? 0.0f : t;
    int lo = (int)tt, hi;
    if (lo >= size - 1) { hi = lo = size - 1; tt = (float)lo; } else hi = lo + 1;
    s.lo = lo; s.hi = hi;
    s.l = tt - (float)lo; s.h = 1.0f - s.l;
    if (!s.valid) { s.lo = s.hi = 0; }
    return s;
}

struct PostArgs {
    const float* pred;        // [V][ROI_CAP][pred_ld]: logits C, then deltas 4C
    int pred_ld, C, V;
    const float* proposals;   // [V][ROI_CAP][4]
    const int* prop_count;
    const ViewDesc* views;
    float score_thr, nms_thr;
    float* prob;              // [V][ROI_CAP][C] scratch (softmax)
    float* pmax;              // [V][ROI_CAP]
    unsigned long long* keys; // [V][key_cap] scratch
    float* cbox;              // [V][2*key_cap][4] scratch: decoded+clipped candidate boxes (raw, then class-offset)
    int* key_count;           // [V] scratch counter
    int key_cap;              // power of two >= ROI_CAP * min(C-1, 19)
    DetBuffers det;
    // decision-margin audit (audit.hip), both null in a plain forward:
    unsigned long long* kept_key = nullptr;   // [V][det.cap]: candidate key of every detection kept, in output order
    float* post_maxc = nullptr;               // [V]: largest clipped coordinate over the view's candidates (the class offset's scale)
};
void launch_frcnn_postprocess(const PostArgs& a, hipStream_t st);

// score.hip
struct ScoreArgs {
    DetBuffers det;            // detections of ALL views of the batch
    const int* ref_view;       // [P] view index of the reference view of pair p
    const int* aug_view;       // [P] view index of the augmented view of pair p
    const int* aug_kind;       // [P] 0 = boxes unchanged, 1 = flip, 2 = scale, 3 = rotate
    const float* aug_param;    // [P][12] flip: {W}; scale: {ratio}; rotate: {a00,a01,a02,a10,a11,a12,sx,sy,W,H}
    const int* ref_sel;        // [nimg][50] sub-sample indices into the reference detections
    const int* ref_n;          // [nimg] number of (sub-sampled) reference boxes
    const int* pair_img;       // [P] image slot of the pair
    int P;
    float bp;
    float* cons;               // [P] consistency_img per pair
};
void launch_consistency(const ScoreArgs& a, hipStream_t st);
void launch_cls_corr(const DetBuffers& det, const int* ref_sel, const int* ref_n, const int* view_img, const int* view_is_ref,
                     int V, float* out /*[V][C-1]*/, hipStream_t st);

// rpn_prune.hip -- certified pruning of the RPN head on P2 / P3 in the exact sweep (the file's header has the argument)
struct RpnPruneArgs {
    const float* feat[2];        // P2, P3: fp32 [pixel][256]
    const LevelSeg* seg[2];
    float* energy[2];            // [pixel][energy_parts] scratch: sum of squares over the 256 channels (4 partial sums of 64 channels each when the FPN
                                 // output conv's epilogue wrote them, ConvArgs::energy4; 1 when prune_energy_kernel did)
    unsigned* split[2];          // optional [pixel][256] words: the split-fp16 form of P2 / P3 (h16.h) written by the energy kernel for the look-ahead conv
    float* pnorm[2];             // [pixel] scratch: |3 x 3 patch|_2 (select kernel -> scatter kernel)
    const float* head[2];        // approximate head maps [pixel][head_ld] (logits = channels 0..2)
    float* head_out[2];          // the same buffers: unselected pixels get logit -FLT_MAX, selected ones their exact rows
    // Two selection stages (rpn_prune.hip): stage 0 = the pixels holding an anchor whose LOWER bound reaches tau (at least k anchors: their exact
    // logits give a sharper threshold), stage 1 = the remaining pixels with an upper bound at or above that threshold.  Single-stage mode
    // (stages == 1, round 5's rule) uses the stage-0 arrays only.
    const float* head_rows[2][2];   // [stage][level] exact head rows of the selected pixels, compact [n_selected][head_ld] per view (at the view's pixel offset)
    int* row_map[2][2];          // [stage][level]: [pixel] -> selected pixel index, compact per view
    int* nsel[2];                // [stage]: [2][V] selected pixels per (level, view): the dyn_rows of the gathered launches
    unsigned* tau_key;           // [2][V] orderable key of tau (written by stage 0, read by stage 1)
    unsigned long long* stat;    // optional [4]: selected / total pixels of P2 and P3 accumulated over the calls (profiling), or null
    unsigned long long* log[2];  // optional, per stage [4]: the same counts of THIS forward's stage only (cald_profile_dump books the gathered launches with them), or null
    float* check;                // [2]: running max of |look-ahead - exact| / bound over the selected anchors (must stay <= 1); 1.0f once an activation left the split's range
    float c1[3], c0[3];          // bound per anchor: c1 * |patch|_2 + c0
    int head_ld, pre_n, V, energy_parts, stages;
};
void launch_rpn_prune_energy(const RpnPruneArgs& a, hipStream_t st);       // before the look-ahead conv (writes its split-form input)
void launch_rpn_prune_select(const RpnPruneArgs& a, int max_pix, int stage, hipStream_t st);   // stage 0 (or the only one), then -- after stage 0's exact rows exist -- stage 1
void launch_rpn_prune_scatter(const RpnPruneArgs& a, hipStream_t st);

// audit.hip -- decision margins of one Faster R-CNN forward (cascade mode: which images may differ from the exact mode by more than
// continuous rounding).  Every discrete decision of the forward (top-k cut, NMS IoU test, score threshold, RoI level, sort
// order where the order matters) leaves its distance to the flip point; a view's record keeps the minimum per kind.
#define CALD_VM 16
enum { VM_RPN_TOPK = 0, VM_RPN_IOU, VM_RPN_ORDER, VM_RPN_TRUNC, VM_RPN_SMALL, VM_ROI_LEVEL, VM_ROI_EDGE, VM_POST_THR, VM_POST_IOU,
       VM_POST_ORDER, VM_POST_CAP, VM_POST_SUBORDER, VM_POST_TOP2 };
struct AuditArgs {
    // RPN stage (buffers of launch_rpn)
    const unsigned long long* cand_key; const float* cand_box; const float* sorted_box; const unsigned char* flags;
    const unsigned long long* next_key; const unsigned long long* trunc_key;
    int pre_n, post_n; float rpn_nms_thr, min_size;
    // RoI stage
    const float* proposals; const int* prop_count; const LevelSeg* seg[4];
    // box-head post-processing
    const float* prob; const float* pred; int pred_ld, C; const ViewDesc* views;
    const unsigned long long* keys; const int* key_count; int key_cap;
    const unsigned long long* kept_key; const float* post_maxc; const int* det_count; int cap;
    float score_thr, post_nms_thr;
    int V;
    float delta;      // relevance slack on scores (>> the fast mode's score error): decisions about boxes more than delta below a cut are ignored
    float* out;       // [V][CALD_VM]
};
void launch_audit(const AuditArgs& a, hipStream_t st);
// per (reference, augmentation) pair: out[2 p] = min over reference boxes of (best IoU - best IoU among detections of ANOTHER proposal),
// out[2 p + 1] = 1 if some reference box overlaps no detection at all (argmax falls on detection 0), else 0
void launch_pair_audit(const ScoreArgs& a, float* out, hipStream_t st);

// retina.hip
struct RetinaArgs {
    const float* cls[5];          // per level [sum pix][cls_ld], channel a*K + k
    const float* reg[5];          // per level [sum pix][reg_ld], channel a*4 + j
    const LevelSeg* seg[5];       // P3..P7 geometry
    const LevelSeg* seg0;
    const ViewDesc* views;
    const float* base_anchors;    // [5][A][4]
    int cls_ld, reg_ld, A, K, V;
    float score_thr, nms_thr, min_box;
    int per_class;                // detections_per_img (300), applied per class
    int cand_cap;                 // power of two >= anchors per view
    int* cand_count;              // [V][K]
    unsigned long long* cand_key; // [V][K][cand_cap]
    float* cand_box;              // [V][K][cand_cap][4]
    unsigned char* cand_skip;     // [V][K][cand_cap] 1 = removed by remove_small_boxes
    int* kept_anchor;             // [V][K][per_class]
    float* kept_box;              // [V][K][per_class][4]
    int* kept_count;              // [V][K]
    DetBuffers det;               // cap >= K * per_class
};
void launch_retina_postprocess(const RetinaArgs& a, int max_anchors, hipStream_t st);

// baseline sweeps
void launch_lt_uncertainty(const DetBuffers& det, int V, float* out, hipStream_t st);
void launch_max_iou(const ScoreArgs& a, float* out /*[P][50]*/, hipStream_t st);
