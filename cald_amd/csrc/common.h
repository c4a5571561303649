// cald_amd/csrc/common.h -- shared declarations of the MI355X (gfx950) CALD hot-path library.
// Product code: never includes or links anything under oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define CALD_MAX_VIEWS 128     // views (image x augmentation) per batched launch (hard cap; the sweep issues forwards of <= fwd_views)
#define CALD_MAX_LEVELS 9      // spatial levels: 0 input, 1 /2, 2 /4 (P2) ... 6 /64 (pool), 7 roi rows per view, 8 all roi rows as ONE segment
#define CALD_ROI_CAP 1000      // rpn_post_nms_top_n: fixed row capacity per view in the roi GEMMs
#define CALD_MAX_CUT 4

// Per-view geometry of one spatial level inside a ragged batch.  Activations of level L with C
// channels live in one buffer; view v starts at float offset pix_off * C and is [H][W][C] (NHWC).
struct LevelSeg {
    long long pix_off;
    int H, W;
    int tile_start;   // first 128-row M tile of this view at this level
    int pad_;
};

// The view an M tile belongs to: the largest v in [0, V) with seg[v].tile_start <= mt (tile_start is the running sum of the views' tile
// counts: non-decreasing, seg[0].tile_start == 0).  A bisection -- seven dependent scalar loads at 92 views where the linear scan it
// replaced made one per view before it, i.e. ~45 on average in front of every workgroup's first operand load.
#if defined(__HIPCC__)
__device__ __forceinline__ int seg_find_view(const LevelSeg* __restrict__ seg, const int V, const int mt) {
    int lo = 0, hi = V - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg[mid].tile_start <= mt) lo = mid; else hi = mid - 1;
    }
    return lo;
}
#endif

// One batch plan = geometry of every level for every view (+ sentinel entry V for tile_start).
struct BatchPlan {
    LevelSeg seg[CALD_MAX_LEVELS][CALD_MAX_VIEWS + 1];
};

struct ConvArgs {
    const float* in;
    float* out;
    const float* w;        // [Kpad][CoutPad], rows in chain order (api.hip conv_k_index: (16-channel chunk, kh, kw, channel) or (kh, kw, cin))
    const float* w4;       // same weights packed [Kpad/16][2][CoutPad][2][4] for conv_p4.hip (k = 16 kt + 8 kq + 2 j + h), or null
    const void* w16;       // fp16 hi/lo split of the same weights * 2^S, [Kpad/16][2 (hi, lo)][CoutPad][16] (conv_h3.hip), or null
    float w16_unscale;     // 2^-(S + 4): undoes the weight scale 2^S and conv_h3's activation scale 2^4 (exact powers of two)
    const float* bias;     // [CoutPad] or null
    const float* scale;    // FrozenBN scale/shift or null
    const float* shift;
    const float* residual; // same geometry as out, or null
    const float* up;       // coarser level to nearest-upsample-add (FPN top-down), or null
    const LevelSeg* seg_in;
    const LevelSeg* seg_out;
    const LevelSeg* seg_up;
    const int* dyn_rows;   // optional per-view dynamic row count (roi GEMMs), else null
    const int* row_map;    // optional (conv_p4.hip, EPI 0): output row m of a view is computed at input-grid pixel row_map[view's pix_off + m] and stored
                           // at row m -- the gathered launches of the certified RPN pruning (rpn_prune.hip); needs dyn_rows; else null
    int V;
    int Cin, Cout, CoutPad, Kpad;
    int KH, KW, stride, pad;
    int relu;
    int total_mtiles;
    int out_ld;            // output row stride in floats (== Cout normally)
    const float* zeros;    // >= 16 bytes of zeros, 16-byte aligned (source of out-of-image taps)
    int in_relu;           // apply ReLU to the input while gathering (LastLevelP6P7: p7(relu(p6)))
    int exp_flags;         // kernel-tuning experiments only (0 in the product path): bit 0 = skip the output stores
    // training backward only (train.hip; honoured by conv_p4.hip): after everything else, out = mask > 0 ? out : 0 -- the ReLU backward
    // of the layer whose saved post-ReLU output `mask` (same geometry and row stride as out) this data gradient flows into; else null
    const float* mask;
    // conv_stem.hip: the 7 x 7 / 2 stem's weights packed [20 quads of k-pairs][2 h][64 n][4] over the 154-slot chain (7 filter rows x
    // (21 (kw, c) taps + 1 zero slot)); set only when every view's output is an exact grid of 8 x 16 blocks, else null
    const float* wstem;
    // CALD_PRECISION_F16X3 only (conv_h3.hip / conv_h4.hip): the split form of an activation tensor (h16.h: per 16-channel chunk
    // [16 fp16 hi | 16 fp16 lo] of 16 x, the same 4 bytes per element as fp32), beside or instead of the fp32 one.  A producer that
    // writes it (out16; `out` may then be null) saves every consumer (in16) the split arithmetic in its k-loop, and lets conv_h4.hip
    // move operands HBM -> LDS with buffer_load ... lds.  Null = fp32 only.
    const unsigned* in16;
    unsigned* out16;
    // conv_p4.hip, exact mode, the FPN output convs of P2 / P3 under the certified RPN pruning (rpn_prune.hip): per pixel four partial sums of
    // squares over the 256 output channels ([pixel][4]: n-tile x wave column, 64 channels each) -- the energy the pruning's bound needs -- and,
    // through out16, the split-fp16 copy its look-ahead conv reads.  Both were a separate pass over P2 / P3 (prune_energy_kernel) in round 5.
    float* energy4;
    int ex16;              // `residual` / `up` point at a split-form tensor (same element count) instead of an fp32 one
    // tuning only (cald_op_conv_bench under CALD_CONV_TRACE; null in the product path): per workgroup eight 64-bit words -- s_memtime at
    // entry, after the prologue's first barrier, after the k-loop, after the epilogue's last store was issued, after the stores drained;
    // HW_ID; XCC_ID; blockIdx -- the timeline of a launch on the chip (tools/conv_trace.py)
    unsigned long long* trace;
};

// Several independent conv problems in ONE launch (the five FPN levels under the shared-weight RPN / RetinaNet heads, the
// FPN output convs): workgroups [blk0[i], blk0[i + 1]) belong to problem i.  Small levels then fill the tail of the large
// ones instead of running as under-filled launches of their own.
#define CALD_MAX_GROUP 10
struct ConvGroup {
    int n;
    int blk0[CALD_MAX_GROUP + 1];
    ConvArgs p[CALD_MAX_GROUP];
};

// Position of input element (tap = kh*KW + kw, channel ci) in the k-ordered fma chain (DESIGN.md arithmetic contract).
//   Cin % 16 == 0:  (channel chunk of 16, kh, kw, channel inside the chunk) -- all taps of a 16-channel chunk are consecutive
//                   k-tiles, so the nine passes of a 3x3 filter re-touch the same 64-byte pixel segments while they are still
//                   in L2 (with cin innermost over the whole channel vector every tap re-streamed the tensor from HBM);
//   otherwise (or more than 32 taps: the tap-validity mask is 32 bits):  (kh, kw, cin), e.g. the 4-channel stem.
// For 1x1 layers and linear layers both orders are the plain channel order.
__host__ __device__ inline int conv_k_index(int tap, int ci, int taps, int cinp) {
    return (cinp % 16 == 0 && taps <= 32) ? ((ci >> 4) * taps + tap) * 16 + (ci & 15) : tap * cinp + ci;
}
__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ inline int cout_pad(int cout) { return cout >= 128 ? round_up(cout, 128) : (cout >= 64 ? round_up(cout, 64) : round_up(cout, 32)); }

// ---------------------------------------------------------------------------------------------
// Deterministic float32 elementary functions (DESIGN.md "arithmetic contract"): fixed fmaf
// polynomials, built with -ffp-contract=off so that every fused multiply-add is explicit.
// They stand where the reference calls torch.exp / softmax / sigmoid / numpy.log
// (frcnn_la.py:40, retinanet_cal.py:411, cald_train.py:214).
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline float det_bits2f(uint32_t u) {
    union { uint32_t u; float f; } c; c.u = u; return c.f;
}
__host__ __device__ inline uint32_t det_f2bits(float f) {
    union { uint32_t u; float f; } c; c.f = f; return c.u;
}

__host__ __device__ inline float det_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return INFINITY;
    if (x < -87.0f) return 0.0f;
    float n = rintf(x * 1.44269504f);
    float r = fmaf(n, -0.693145752f, x);
    r = fmaf(n, -1.42860677e-6f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    int ni = (int)n;
    int n1 = ni / 2, n2 = ni - n1;
    y = y * det_bits2f((uint32_t)(n1 + 127) << 23);
    y = y * det_bits2f((uint32_t)(n2 + 127) << 23);
    return y;
}

__host__ __device__ inline float det_logf(float x) {
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    uint32_t u = det_f2bits(x);
    if (u < 0x00800000u) { x = x * 8388608.0f; u = det_f2bits(x); e = -23; }
    e += (int)(u >> 23) - 126;
    float m = det_bits2f((u & 0x007fffffu) | 0x3f000000u);
    float f;
    if (m < 0.70710678f) { e -= 1; f = (m + m) - 1.0f; } else { f = m - 1.0f; }
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float y = (p * f) * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float r = f + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}
// sin and cos for x in [0, 2*pi] (Box-Muller angle of torch.randn): quadrant reduction + minimax polynomials
__host__ __device__ inline void det_sincosf(float x, float* s, float* c) {
    float q = rintf(x * 0.636619772f);
    float r = fmaf(q, -1.5703125f, x);
    r = fmaf(q, -4.837512969970703125e-4f, r);
    r = fmaf(q, -7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = fmaf(ps, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    float sn = fmaf(ps * z, r, r);
    float pc = 2.443315711809948e-5f;
    pc = fmaf(pc, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    float cs = fmaf(pc * z, z, fmaf(z, -0.5f, 1.0f));
    int qi = ((int)q) & 3;
    float ss = (qi & 1) ? cs : sn, cc = (qi & 1) ? sn : cs;
    if (qi == 2 || qi == 3) ss = -ss;
    if (qi == 1 || qi == 2) cc = -cc;
    *s = ss; *c = cc;
}
__host__ __device__ inline float det_log2f(float x) { return det_logf(x) * 1.44269504f; }
__host__ __device__ inline float det_sigmoidf(float x) { return 1.0f / (1.0f + det_expf(-x)); }

#ifdef __HIPCC__
// The split of one activation: 16 x = hi + lo (+ 2^-22 relative); the word a producer stores for its consumers (ConvArgs::out16).
// The same two conversions the loader performs on an fp32 input, so a pre-split tensor gives bit-identical products.
__device__ __forceinline__ unsigned split16_word(const float x) {
    const float s = x * 16.0f;
    const _Float16 hi = (_Float16)s;
    const _Float16 lo = (_Float16)(s - (float)hi);
    return (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
}
__device__ __forceinline__ uint4 split16_word4(const float4 v) {
    return make_uint4(split16_word(v.x), split16_word(v.y), split16_word(v.z), split16_word(v.w));
}
#endif

#define BBOX_XFORM_CLIP_F 4.135166556742356f  /* (float)math.log(1000/16) */

// BoxCoder.decode_single (torchvision 0.8.2 _utils.py; weights as arguments)
__host__ __device__ inline void det_box_decode(const float* box, const float* d, float wx, float wy, float ww,
                                               float wh, float* o) {
    float width = box[2] - box[0], height = box[3] - box[1];
    float cx = box[0] + 0.5f * width, cy = box[1] + 0.5f * height;
    float dx = d[0] / wx, dy = d[1] / wy, dw = d[2] / ww, dh = d[3] / wh;
    if (dw > BBOX_XFORM_CLIP_F) dw = BBOX_XFORM_CLIP_F;
    if (dh > BBOX_XFORM_CLIP_F) dh = BBOX_XFORM_CLIP_F;
    float pcx = dx * width + cx, pcy = dy * height + cy;
    float pw = det_expf(dw) * width, ph = det_expf(dh) * height;
    o[0] = pcx - 0.5f * pw; o[1] = pcy - 0.5f * ph; o[2] = pcx + 0.5f * pw; o[3] = pcy + 0.5f * ph;
}
__host__ __device__ inline float det_clamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Orderable 32-bit key of a float: larger float -> larger unsigned (NaN sorts above +inf).
__host__ __device__ inline uint32_t det_orderable(float f) {
    uint32_t u = det_f2bits(f);
    if ((u << 1) == 0u) u = 0u;   // -0.0 and +0.0 compare equal
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE function attribute: raise it once for every device a
// kernel is launched on (one context per device may live in the same process), thread-safely.
#include <atomic>
struct PerDeviceOnce {
    std::atomic<bool> done[64];
    PerDeviceOnce() { for (auto& d : done) d.store(false); }
    bool first() {   // true exactly once per current device
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        return !done[dev].exchange(true);
    }
};
template <typename K> static inline void allow_big_lds(PerDeviceOnce& once, K kernel) {
    if (once.first())
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
}

// ---------------------------------------------------------------------------------------------
// kernel launchers (implemented in the .hip files; all asynchronous on `stream`)
// ---------------------------------------------------------------------------------------------
void launch_conv(const ConvArgs& a, hipStream_t stream);
// all problems in one launch when they qualify for the same tiled kernel variant, else one launch each
int launch_conv_group(const ConvArgs* probs, int n, hipStream_t stream);   // returns the number of kernel launches issued
