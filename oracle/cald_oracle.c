/* =====================================================================================
 * cald_oracle.c  --  TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C) of the CALD
 * consistency sweep hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product (cald_amd/) never does.
 *
 * Parity status
 *   scoring half  (rows A1-A11, A23): PINNED against golden vectors captured from the
 *       imported reference (oracle/ref_harness.py + oracle/make_golden.py -> tests/golden/).
 *   detector half (rows A13-A20):     "parity unpinned" -- its arithmetic lives in
 *       torchvision 0.8.2 (reference README.md:10), which is neither vendored in
 *       /root/reference nor installable here.  The restatement follows the in-repo pins
 *       detection/frcnn_la.py:32-87 (postprocess), :292-315 (box rescale),
 *       detection/frcnn_ll.py:207-238,284-321,323-374 (RPN copy) and the published
 *       torchvision 0.8.2 algorithms; convolutions / linear layers are
 *       additionally cross-checked against torch.nn.functional on CPU in tests/.
 *
 * Arithmetic contract (what makes GPU<->oracle comparison bit-exact):
 *   all float32, compiled with -ffp-contract=off; dot products are ONE k-ordered fmaf chain
 *   per output starting from +0 (k order for convs: (16-channel chunk, kh, kw, channel) when Cin % 16 == 0, else (kh, kw, cin); natural for linear);
 *   exp/log are the fixed polynomials of orc_math.h; sorts are (key desc, index asc).
 * ===================================================================================== */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "orc_math.h"

#define ORC_API __attribute__((visibility("default")))
/* OpenMP team size of the oracle loops (a 256-thread host makes tiny test convolutions crawl) */
static int orc_threads = 16;
ORC_API void orc_set_threads(int n) { orc_threads = n < 1 ? 1 : n; }
static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }

ORC_API float orc_exp(float x) { return orc_expf(x); }
ORC_API float orc_log(float x) { return orc_logf(x); }
ORC_API void orc_exp_array(const float* x, float* y, int n) { for (int i = 0; i < n; i++) y[i] = orc_expf(x[i]); }
ORC_API void orc_log_array(const float* x, float* y, int n) { for (int i = 0; i < n; i++) y[i] = orc_logf(x[i]); }

/* -------------------------------------------------------------------------------------
 * Wave-order sum: the scoring kernel reduces over classes with a 64-lane butterfly
 * (lane i adds lane i^32, i^16, ... i^1); classes beyond 64 are pre-added per lane.
 * The oracle performs the same additions in the same order (all lanes end equal).
 * ------------------------------------------------------------------------------------- */
static float wave_sum(const float* v, int n) {
    float lane[64];
    for (int l = 0; l < 64; l++) {
        float s = 0.0f;
        for (int k = l; k < n; k += 64) s = s + v[k];
        lane[l] = s;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        float t[64];
        for (int l = 0; l < 64; l++) t[l] = lane[l] + lane[l ^ off];
        memcpy(lane, t, sizeof(t));
    }
    return lane[0];
}

/* scipy.stats.entropy(pk, qk) element: scipy.special.rel_entr */
static inline float rel_entr(float x, float y) {
    if (x != x || y != y) return NAN;
    if (x > 0.0f && y > 0.0f) return x * orc_logf(x / y);
    if (x == 0.0f && y >= 0.0f) return 0.0f;
    return INFINITY;
}

/* cald_train.py:211-216  JS divergence of two class vectors through scipy.stats.entropy,
 * which renormalises p, q and m=(p+q)/2 to sum 1 and stays in float32. */
ORC_API float orc_js_divergence(const float* p, const float* q, int C) {
    float m[256], t1[256], t2[256];
    for (int k = 0; k < C; k++) m[k] = (p[k] + q[k]) / 2.0f;
    float sp = wave_sum(p, C), sq = wave_sum(q, C), sm = wave_sum(m, C);
    for (int k = 0; k < C; k++) {
        float pk = p[k] / sp, qk = q[k] / sq, mk = m[k] / sm;
        t1[k] = rel_entr(pk, mk);
        t2[k] = rel_entr(qk, mk);
    }
    float js = 0.5f * wave_sum(t1, C) + 0.5f * wave_sum(t2, C);
    if (js < 0.0f) js = 0.0f;
    return js;
}

/* cald_train.py:203-210  IoU of one transformed reference box against one detection */
static inline float cald_iou(const float* ab, const float* B) {
    float w = fminf_(ab[2], B[2]) - fmaxf_(ab[0], B[0]);
    float h = fminf_(ab[3], B[3]) - fmaxf_(ab[1], B[1]);
    float Aarea = (ab[2] - ab[0]) * (ab[3] - ab[1]);
    float Barea = (B[2] - B[0]) * (B[3] - B[1]);
    float inter = w * h;
    float iou = inter / ((Aarea + Barea) - inter);
    if (w < 0.0f) iou = 0.0f;
    if (h < 0.0f) iou = 0.0f;
    return iou;
}

/* cald_train.py:189-224 for ONE augmented view.  Returns consistency_img (min over
 * reference boxes, initial 1.0) and, through the optional detail arrays, the per-reference
 * -box (max_iou, argmax, js, score).  M == 0 -> 0.0 (cald_train.py:198-201). */
ORC_API float orc_consistency_view(int N, const float* aug_box, const float* ref_scores_cls,
                                   const float* ref_pm, int M, const float* boxes,
                                   const float* scores_cls, const float* pm, int C, float bp,
                                   float* d_maxiou, int* d_argmax, float* d_js, float* d_score) {
    if (M == 0) return 0.0f;
    float cons = 1.0f;
    for (int i = 0; i < N; i++) {
        const float* ab = aug_box + 4 * i;
        int j = 0;
        float best = cald_iou(ab, boxes);
        for (int k = 1; k < M; k++) {           /* torch.argmax: first maximum, NaN is maximal */
            float v = cald_iou(ab, boxes + 4 * k);
            if (best != best) break;
            if (v > best || v != v) { best = v; j = k; }
        }
        float js = orc_js_divergence(ref_scores_cls + (size_t)i * C, scores_cls + (size_t)j * C, C);
        float t = 0.5f * (1.0f - js);
        float u = ref_pm[i] + pm[j];
        float s = fabsf((best + t * u) - bp);
        if (s < cons) cons = s;
        if (d_maxiou) { d_maxiou[i] = best; d_argmax[i] = j; d_js[i] = js; d_score[i] = s; }
    }
    return cons;
}

/* cald_train.py:114-117 / :194-197  per-view class-max vector; python negative indexing for
 * label 0 (RetinaNet quirk).  out has C-1 slots, zero-initialised here. */
ORC_API void orc_cls_corr_view(int n, const float* scores, const int64_t* labels, int C, float* out) {
    for (int k = 0; k < C - 1; k++) out[k] = 0.0f;
    for (int d = 0; d < n; d++) {
        long l = (long)labels[d] - 1;
        if (l < 0) l += C - 1;
        if (l < 0 || l >= C - 1) continue;
        if (scores[d] > out[l]) out[l] = scores[d];
    }
}

/* cald_train.py:110-113: np.round(np.linspace(0, n-1, 50)).astype(int) */
ORC_API int orc_subsample_indices(int n, int* inds) {
    if (n <= 40) { for (int i = 0; i < n; i++) inds[i] = i; return n; }
    double step = (double)(n - 1) / 49.0;
    for (int i = 0; i < 50; i++) {
        double v = (i == 49) ? (double)(n - 1) : (double)i * step + 0.0;
        inds[i] = (int)nearbyint(v);
    }
    return 50;
}

/* -------------------------------------------------------------------------------------
 * Python `random` (MT19937, random.seed(int) + random.uniform) -- cald_helper.py:108-114
 * draws from the global Python RNG; the sweep re-seeds it per pool position so that the
 * result is independent of sharding (SURVEY section 7 "RNG-dependent augmentations").
 * ------------------------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } orc_mt;
static void mt_init_genrand(orc_mt* s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}
static void mt_init_by_array(orc_mt* s, const uint32_t* key, int klen) {
    mt_init_genrand(s, 19650218u);
    int i = 1, j = 0, k = (624 > klen ? 624 : klen);
    for (; k; k--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
        if (j >= klen) j = 0;
    }
    for (k = 623; k; k--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
    }
    s->mt[0] = 0x80000000u;
}
static uint32_t mt_next(orc_mt* s) {
    if (s->idx >= 624) {
        uint32_t* mt = s->mt;
        for (int kk = 0; kk < 624; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[(kk + 1) % 624] & 0x7fffffffu);
            mt[kk] = mt[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
}
static void mt_seed_py(orc_mt* s, uint64_t seed) {
    uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
    mt_init_by_array(s, key, key[1] ? 2 : 1);
}
static double mt_random(orc_mt* s) {
    uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
    return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}
static double mt_uniform(orc_mt* s, double a, double b) { return a + (b - a) * mt_random(s); }
ORC_API void orc_py_random(uint64_t seed, int n, double* out) {
    orc_mt s; mt_seed_py(&s, seed);
    for (int i = 0; i < n; i++) out[i] = mt_random(&s);
}

/* cald_helper.py:88-132 cutout: selects up to cut_num rectangles (left, top, right, bottom
 * ints).  boxes are the (sub-sampled) reference detections in ORIGINAL image coordinates.
 * Returns the number of rectangles accepted. */
static int cutout_impl(orc_mt* sp, int H, int W, int N, const float* boxes, int cut_num,
                       float remove_thres, float min_thres, int* rects) {
    int count = 0;
    for (int t = 0; t < 50; t++) {
        double sh = mt_uniform(sp, 0.05 * H, 0.2 * H);
        double sw = mt_uniform(sp, 0.05 * W, 0.2 * W);
        double left = mt_uniform(sp, 0.0, (double)W - sw);
        double right = left + sw;
        double top = mt_uniform(sp, 0.0, (double)H - sh);
        double bottom = top + sh;
        int il = (int)left, it = (int)top, ir = (int)right, ib = (int)bottom;
        float c[4] = {(float)il, (float)it, (float)ir, (float)ib};
        float rmax = 0.0f; int isnan_ = 0;
        for (int i = 0; i < N; i++) {           /* intersect(), cald_helper.py:226-243 */
            const float* b = boxes + 4 * i;
            float iw = fminf_(c[2], b[2]) - fmaxf_(c[0], b[0]); if (iw < 0.0f) iw = 0.0f;
            float ih = fminf_(c[3], b[3]) - fmaxf_(c[1], b[1]); if (ih < 0.0f) ih = 0.0f;
            float area = (b[2] - b[0]) * (b[3] - b[1]);
            float ratio = (iw * ih) / area;
            if (ratio != ratio) isnan_ = 1;
            if (i == 0 || ratio > rmax) rmax = ratio;
        }
        if (!isnan_ && (rmax > remove_thres || rmax < min_thres)) continue;
        rects[4 * count + 0] = il; rects[4 * count + 1] = it; rects[4 * count + 2] = ir; rects[4 * count + 3] = ib;
        count++;
        if (count >= cut_num) break;
    }
    return count;
}
ORC_API int orc_cutout_rects(uint64_t seed, int H, int W, int N, const float* boxes, int cut_num,
                             float remove_thres, float min_thres, int* rects) {
    orc_mt s; mt_seed_py(&s, seed);
    return cutout_impl(&s, H, W, N, boxes, cut_num, remove_thres, min_thres, rects);
}
/* One Python `random` generator kept across several draws of one image (get_uncertainty calls ColorSwap's
 * random.randint and cutout's random.uniform on the same global generator, cald_train.py:140-166). */
ORC_API void* orc_pyrandom_new(uint64_t seed) {
    orc_mt* s = (orc_mt*)malloc(sizeof(orc_mt));
    mt_seed_py(s, seed);
    return s;
}
ORC_API void orc_pyrandom_free(void* st) { free(st); }
/* random.randint(0, n - 1) == randrange(n): _randbelow_with_getrandbits (k = n.bit_length(), rejection) */
ORC_API int orc_pyrandom_randbelow(void* st, int n) {
    int k = 0;
    for (int t = n; t; t >>= 1) k++;
    for (;;) {
        const uint32_t r = mt_next((orc_mt*)st) >> (32 - k);
        if ((int)r < n) return (int)r;
    }
}
ORC_API int orc_cutout_rects_st(void* st, int H, int W, int N, const float* boxes, int cut_num,
                                float remove_thres, float min_thres, int* rects) {
    return cutout_impl((orc_mt*)st, H, W, N, boxes, cut_num, remove_thres, min_thres, rects);
}

/* -------------------------------------------------------------------------------------
 * PIL Image.resize(size, BILINEAR) on 8-bit RGB (cald_helper.py:47-53): Pillow's two-pass
 * antialiased fixed-point resampler (published algorithm, src/libImaging/Resample.c).
 * ------------------------------------------------------------------------------------- */
#define PIL_PRECISION_BITS (32 - 8 - 2)
static double pil_filter(int fid, double x) {
    if (x < 0.0) x = -x;
    if (fid == 0) return x < 1.0 ? 1.0 - x : 0.0;
    /* bicubic, a = -0.5 (Pillow Resample.c bicubic_filter) */
    if (x < 1.0) return ((-0.5 + 2.0) * x - (-0.5 + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * -0.5;
    return 0.0;
}
static int pil_coeffs_f(int inSize, int outSize, int fid, int** bounds_out, int32_t** kk_out);
static int pil_coeffs(int inSize, int outSize, int** bounds_out, int32_t** kk_out) { return pil_coeffs_f(inSize, outSize, 0, bounds_out, kk_out); }
static int pil_coeffs_f(int inSize, int outSize, int fid, int** bounds_out, int32_t** kk_out) {
    double scale = (double)inSize / (double)outSize, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    double support = (fid == 0 ? 1.0 : 2.0) * filterscale;
    int ksize = (int)ceil(support) * 2 + 1;
    double* pre = (double*)malloc(sizeof(double) * (size_t)outSize * ksize);
    int* bounds = (int*)malloc(sizeof(int) * 2 * (size_t)outSize);
    int32_t* kk = (int32_t*)malloc(sizeof(int32_t) * (size_t)outSize * ksize);
    for (int xx = 0; xx < outSize; xx++) {
        double center = 0.0 + (xx + 0.5) * scale, ww = 0.0, ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5); if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5); if (xmax > inSize) xmax = inSize;
        xmax -= xmin;
        double* k = pre + (size_t)xx * ksize;
        int x;
        for (x = 0; x < xmax; x++) {
            double w = pil_filter(fid, (x + xmin - center + 0.5) * ss);
            k[x] = w; ww += w;
        }
        for (x = 0; x < xmax; x++) if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; x++) k[x] = 0;
        bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    }
    for (size_t i = 0; i < (size_t)outSize * ksize; i++)
        kk[i] = pre[i] < 0 ? (int32_t)(-0.5 + pre[i] * (1 << PIL_PRECISION_BITS))
                           : (int32_t)(0.5 + pre[i] * (1 << PIL_PRECISION_BITS));
    free(pre);
    *bounds_out = bounds; *kk_out = kk;
    return ksize;
}
static inline uint8_t pil_clip8(int32_t v) {
    v >>= PIL_PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
ORC_API void orc_pil_coeffs(int inSize, int outSize, int* ksize_out, int* bounds, int32_t* kk) {
    int *b; int32_t* k;
    int ks = pil_coeffs(inSize, outSize, &b, &k);
    *ksize_out = ks;
    if (bounds) memcpy(bounds, b, sizeof(int) * 2 * (size_t)outSize);
    if (kk) memcpy(kk, k, sizeof(int32_t) * (size_t)outSize * ks);
    free(b); free(k);
}
static void pil_resize_f(const uint8_t* src, int H, int W, uint8_t* dst, int oh, int ow, int fid);
ORC_API void orc_pil_resize_bilinear(const uint8_t* src, int H, int W, uint8_t* dst, int oh, int ow) { pil_resize_f(src, H, W, dst, oh, ow, 0); }
ORC_API void orc_pil_resize_bicubic(const uint8_t* src, int H, int W, uint8_t* dst, int oh, int ow) { pil_resize_f(src, H, W, dst, oh, ow, 1); }
static void pil_resize_f(const uint8_t* src, int H, int W, uint8_t* dst, int oh, int ow, int fid) {
    int *bh, *bv; int32_t *kh, *kv;
    const uint8_t* cur = src; int curW = W; uint8_t* tmp = NULL;
    if (ow != W) {
        int ks = pil_coeffs_f(W, ow, fid, &bh, &kh);
        tmp = (uint8_t*)malloc((size_t)H * ow * 3);
        for (int y = 0; y < H; y++)
            for (int xx = 0; xx < ow; xx++) {
                int xmin = bh[2 * xx], xmax = bh[2 * xx + 1];
                const int32_t* k = kh + (size_t)xx * ks;
                for (int c = 0; c < 3; c++) {
                    int32_t ss = 1 << (PIL_PRECISION_BITS - 1);
                    for (int x = 0; x < xmax; x++) ss += (int32_t)src[((size_t)y * W + x + xmin) * 3 + c] * k[x];
                    tmp[((size_t)y * ow + xx) * 3 + c] = pil_clip8(ss);
                }
            }
        free(bh); free(kh);
        cur = tmp; curW = ow;
    }
    if (oh != H) {
        int ks = pil_coeffs_f(H, oh, fid, &bv, &kv);
        for (int yy = 0; yy < oh; yy++) {
            int ymin = bv[2 * yy], ymax = bv[2 * yy + 1];
            const int32_t* k = kv + (size_t)yy * ks;
            for (int x = 0; x < curW; x++)
                for (int c = 0; c < 3; c++) {
                    int32_t ss = 1 << (PIL_PRECISION_BITS - 1);
                    for (int y = 0; y < ymax; y++) ss += (int32_t)cur[((size_t)(y + ymin) * curW + x) * 3 + c] * k[y];
                    dst[((size_t)yy * curW + x) * 3 + c] = pil_clip8(ss);
                }
        }
        free(bv); free(kv);
    } else {
        memcpy(dst, cur, (size_t)oh * curW * 3);
    }
    free(tmp);
}

/* -------------------------------------------------------------------------------------
 * Detector transform (torchvision GeneralizedRCNNTransform)
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_transform_size(int H, int W, int min_size, int max_size, int* Hr, int* Wr, int* Hp, int* Wp) {
    double mn = (double)(H < W ? H : W), mx = (double)(H > W ? H : W);
    double scale = (double)min_size / mn;
    if (mx * scale > (double)max_size) scale = (double)max_size / mx;
    *Hr = (int)floor((double)H * scale);
    *Wr = (int)floor((double)W * scale);
    *Hp = ((*Hr + 31) / 32) * 32;
    *Wp = ((*Wr + 31) / 32) * 32;
}

static const float ORC_MEAN[3] = {0.485f, 0.456f, 0.406f};
static const float ORC_STD[3] = {0.229f, 0.224f, 0.225f};

/* One view: uint8 HWC source -> (flip | cutout) -> to_tensor (/255) -> normalise -> bilinear
 * resize (align_corners=False, scale = in/out) -> zero pad.  Output NHWC with 4 channels
 * (4th = 0), [Hp][Wp][4].  rects are (left, top, right, bottom) in view coordinates. */
ORC_API void orc_preprocess_view(const uint8_t* src, int H, int W, int flip, int nrect, const int* rects,
                                 int Hr, int Wr, int Hp, int Wp, float* out, const float* noise /* CHW or NULL */) {
    memset(out, 0, sizeof(float) * (size_t)Hp * Wp * 4);
    float sh = (float)H / (float)Hr, sw = (float)W / (float)Wr;
#pragma omp parallel for schedule(static) num_threads(orc_threads)
    for (int y = 0; y < Hr; y++) {
        float fy = sh * ((float)y + 0.5f) - 0.5f; if (fy < 0.0f) fy = 0.0f;
        int y0 = (int)fy; int y1 = y0 + (y0 < H - 1 ? 1 : 0);
        float ly = fy - (float)y0, hy = 1.0f - ly;
        for (int x = 0; x < Wr; x++) {
            float fx = sw * ((float)x + 0.5f) - 0.5f; if (fx < 0.0f) fx = 0.0f;
            int x0 = (int)fx; int x1 = x0 + (x0 < W - 1 ? 1 : 0);
            float lx = fx - (float)x0, hx = 1.0f - lx;
            int ys[2] = {y0, y1}, xs[2] = {x0, x1};
            float v[2][2][3];
            for (int a = 0; a < 2; a++)
                for (int b = 0; b < 2; b++) {
                    int yy = ys[a], xx = xs[b], cut = 0;
                    for (int r = 0; r < nrect; r++)
                        if (xx >= rects[4 * r] && xx < rects[4 * r + 2] && yy >= rects[4 * r + 1] && yy < rects[4 * r + 3]) cut = 1;
                    int sx = flip ? (W - 1 - xx) : xx;
                    for (int c = 0; c < 3; c++) {
                        float u = cut ? 0.0f : (float)src[((size_t)yy * W + sx) * 3 + c] / 255.0f;
                        if (noise) u = u + noise[((size_t)c * H + yy) * W + sx];
                        v[a][b][c] = (u - ORC_MEAN[c]) / ORC_STD[c];
                    }
                }
            float* o = out + ((size_t)y * Wp + x) * 4;
            for (int c = 0; c < 3; c++)
                o[c] = hy * (hx * v[0][0][c] + lx * v[0][1][c]) + ly * (hx * v[1][0][c] + lx * v[1][1][c]);
        }
    }
}

/* -------------------------------------------------------------------------------------
 * Convolution, NHWC, weights given K-major [KH][KW][Cin][Cout].  One fmaf chain per output from +0, in
 * (16-channel chunk, kh, kw, channel) order when Cin % 16 == 0 (<= 32 taps), else (kh, kw, cin); then (+bias) -> (*bn_scale, +bn_shift as two roundings,
 * FrozenBatchNorm2d: x*scale+bias) -> (+residual) -> (+nearest-upsampled `up`) -> ReLU.
 * ------------------------------------------------------------------------------------- */
#define CT_P 6
#define CT_V 32
__attribute__((target_clones("arch=skylake-avx512", "default")))
ORC_API void orc_conv2d_nhwc(const float* in, int H, int W, int Cin, const float* wk, int Cout, int KH, int KW,
                             int stride, int pad, const float* bias, const float* bn_scale, const float* bn_shift,
                             const float* residual, const float* up, int upH, int upW, int relu,
                             float* out, int Ho, int Wo) {
    float* zero = (float*)calloc((size_t)Cin, sizeof(float));
    long npix = (long)Ho * Wo;
    float uph_scale = up ? (float)upH / (float)Ho : 0.0f, upw_scale = up ? (float)upW / (float)Wo : 0.0f;
#pragma omp parallel for schedule(dynamic, 2) num_threads(orc_threads)
    for (long pb0 = 0; pb0 < npix; pb0 += CT_P * 16) {
      /* weights chunk [K][CT_V] is reused across the 16 pixel tiles of this block */
      for (int co0 = 0; co0 < Cout; co0 += CT_V) {
       for (long p0 = pb0; p0 < pb0 + CT_P * 16 && p0 < npix; p0 += CT_P) {
        int np_ = (int)(npix - p0 < CT_P ? npix - p0 : CT_P);
        {
            int nv = Cout - co0 < CT_V ? Cout - co0 : CT_V;
            float acc[CT_P][CT_V];
            for (int p = 0; p < CT_P; p++) for (int v = 0; v < CT_V; v++) acc[p][v] = 0.0f;
            /* chain order (DESIGN.md contract): Cin % 16 == 0 and <= 32 taps: (16-channel chunk, kh, kw, channel in chunk);
             * otherwise (kh, kw, cin).  1x1 / linear layers: plain channel order either way. */
            const int chunked = (Cin % 16 == 0) && (KH * KW <= 32);
            const int cstep = chunked ? 16 : Cin;
            for (int c0 = 0; c0 < Cin; c0 += cstep)
            for (int kh = 0; kh < KH; kh++)
                for (int kw = 0; kw < KW; kw++) {
                    const float* rows[CT_P];
                    for (int p = 0; p < CT_P; p++) {
                        rows[p] = zero;
                        if (p < np_) {
                            long pp = p0 + p; int oy = (int)(pp / Wo), ox = (int)(pp % Wo);
                            int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
                            if (iy >= 0 && iy < H && ix >= 0 && ix < W) rows[p] = in + ((size_t)iy * W + ix) * Cin;
                        }
                    }
                    const float* wbase = wk + ((size_t)(kh * KW + kw) * Cin) * Cout + co0;
                    if (nv == CT_V) {
                        for (int ci = c0; ci < c0 + cstep; ci++) {
                            const float* wr = wbase + (size_t)ci * Cout;
#pragma GCC unroll 8
                            for (int p = 0; p < CT_P; p++) {
                                float a = rows[p][ci];
#pragma omp simd
                                for (int v = 0; v < CT_V; v++) acc[p][v] = __builtin_fmaf(a, wr[v], acc[p][v]);
                            }
                        }
                    } else {
                        for (int ci = c0; ci < c0 + cstep; ci++) {
                            const float* wr = wbase + (size_t)ci * Cout;
                            for (int p = 0; p < CT_P; p++) {
                                float a = rows[p][ci];
                                for (int v = 0; v < nv; v++) acc[p][v] = __builtin_fmaf(a, wr[v], acc[p][v]);
                            }
                        }
                    }
                }
            for (int p = 0; p < np_; p++) {
                long pp = p0 + p;
                int oy = (int)(pp / Wo), ox = (int)(pp % Wo);
                const float* upr = NULL;
                if (up) {
                    int sy = (int)floorf((float)oy * uph_scale); if (sy > upH - 1) sy = upH - 1;
                    int sx = (int)floorf((float)ox * upw_scale); if (sx > upW - 1) sx = upW - 1;
                    upr = up + ((size_t)sy * upW + sx) * Cout;
                }
                for (int v = 0; v < nv; v++) {
                    int co = co0 + v;
                    float r = acc[p][v];
                    if (bias) r = r + bias[co];
                    if (bn_scale) { r = r * bn_scale[co]; r = r + bn_shift[co]; }
                    if (residual) r = r + residual[(size_t)pp * Cout + co];
                    if (upr) r = r + upr[co];
                    if (relu) r = r > 0.0f ? r : 0.0f;
                    out[(size_t)pp * Cout + co] = r;
                }
            }
        }
       }
      }
    }
    free(zero);
}

/* Linear: out[m][n] = relu?(chain_k(in[m][k]*w[n][k]) + bias[n]);  wk given K-major [K][N]. */
ORC_API void orc_linear(const float* in, int M, int K, const float* wk, int N, const float* bias, int relu, float* out) {
    orc_conv2d_nhwc(in, 1, M, K, wk, N, 1, 1, 1, 0, bias, NULL, NULL, NULL, NULL, 0, 0, relu, out, 1, M);
}

/* max_pool2d(k=3, s=2, p=1), NHWC */
ORC_API void orc_maxpool3x3s2(const float* in, int H, int W, int C, float* out, int Ho, int Wo) {
#pragma omp parallel for schedule(static) num_threads(orc_threads)
    for (int oy = 0; oy < Ho; oy++)
        for (int ox = 0; ox < Wo; ox++)
            for (int c = 0; c < C; c++) {
                float m = -INFINITY;
                for (int kh = 0; kh < 3; kh++)
                    for (int kw = 0; kw < 3; kw++) {
                        int iy = oy * 2 - 1 + kh, ix = ox * 2 - 1 + kw;
                        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                        float v = in[((size_t)iy * W + ix) * C + c];
                        if (v > m || v != v) m = v;
                    }
                out[((size_t)oy * Wo + ox) * C + c] = m;
            }
}

/* LastLevelMaxPool: max_pool2d(x, 1, 2, 0) = every second pixel */
ORC_API void orc_subsample2(const float* in, int H, int W, int C, float* out, int Ho, int Wo) {
    for (int oy = 0; oy < Ho; oy++)
        for (int ox = 0; ox < Wo; ox++)
            memcpy(out + ((size_t)oy * Wo + ox) * C, in + ((size_t)(oy * 2) * W + ox * 2) * C, sizeof(float) * C);
}

/* -------------------------------------------------------------------------------------
 * Anchors (torchvision AnchorGenerator, Appendix A): base = round(stack(-ws,-hs,ws,hs)/2)
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_base_anchors(const float* sizes, int ns, const float* ratios, int nr, float* base /*[nr*ns][4]*/) {
    for (int r = 0; r < nr; r++) {
        float hr = sqrtf(ratios[r]);
        float wr = 1.0f / hr;
        for (int s = 0; s < ns; s++) {
            float ws = wr * sizes[s], hs = hr * sizes[s];
            float* b = base + 4 * (r * ns + s);
            b[0] = rintf(-ws / 2.0f); b[1] = rintf(-hs / 2.0f); b[2] = rintf(ws / 2.0f); b[3] = rintf(hs / 2.0f);
        }
    }
}

/* BoxCoder.decode_single for one (anchor/proposal, delta) pair. */
static inline void box_decode(const float* box, const float* d, float wx, float wy, float ww, float wh, float clipv, float* o) {
    float width = box[2] - box[0], height = box[3] - box[1];
    float cx = box[0] + 0.5f * width, cy = box[1] + 0.5f * height;
    float dx = d[0] / wx, dy = d[1] / wy, dw = d[2] / ww, dh = d[3] / wh;
    if (dw > clipv) dw = clipv;
    if (dh > clipv) dh = clipv;
    float pcx = dx * width + cx, pcy = dy * height + cy;
    float pw = orc_expf(dw) * width, ph = orc_expf(dh) * height;
    o[0] = pcx - 0.5f * pw; o[1] = pcy - 0.5f * ph; o[2] = pcx + 0.5f * pw; o[3] = pcy + 0.5f * ph;
}
#define BBOX_XFORM_CLIP 4.135166556742356 /* math.log(1000/16) */

static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* sort helper: (key desc, index asc) */
typedef struct { float key; int idx; } orc_kv;
static int kv_cmp(const void* a, const void* b) {
    const orc_kv *x = (const orc_kv*)a, *y = (const orc_kv*)b;
    if (x->key > y->key) return -1;
    if (x->key < y->key) return 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* torchvision nms on already score-sorted boxes: keep i, suppress later j with IoU > thr */
static int nms_sorted(const float* b /*[n][4] sorted*/, int n, float thr, int max_keep, int* keep) {
    uint8_t* dead = (uint8_t*)calloc((size_t)n + 1, 1);
    int nk = 0;
    for (int i = 0; i < n; i++) {
        if (dead[i]) continue;
        keep[nk++] = i;
        if (nk >= max_keep) break;
        const float* bi = b + 4 * i;
        float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
        for (int j = i + 1; j < n; j++) {
            if (dead[j]) continue;
            const float* bj = b + 4 * j;
            float xx1 = fmaxf_(bi[0], bj[0]), yy1 = fmaxf_(bi[1], bj[1]);
            float xx2 = fminf_(bi[2], bj[2]), yy2 = fminf_(bi[3], bj[3]);
            float w = fmaxf_(0.0f, xx2 - xx1), h = fmaxf_(0.0f, yy2 - yy1);
            float inter = w * h;
            float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
            float ovr = inter / ((ai + aj) - inter);
            if (ovr > thr) dead[j] = 1;
        }
    }
    free(dead);
    return nk;
}

/* batched_nms: boxes + idx*(max_coord+1) in fp32, one nms, keep in score order (first max_keep).
 * keep receives indices into the INPUT arrays. */
ORC_API int orc_batched_nms(const float* boxes, const float* scores, const int* groups, int n, float thr,
                            int max_keep, int* keep) {
    if (n == 0) return 0;
    float maxc = boxes[0];
    for (int i = 0; i < 4 * n; i++) if (boxes[i] > maxc) maxc = boxes[i];
    orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * n);
    for (int i = 0; i < n; i++) { kv[i].key = scores[i]; kv[i].idx = i; }
    qsort(kv, n, sizeof(orc_kv), kv_cmp);
    float* sb = (float*)malloc(sizeof(float) * 4 * n);
    for (int i = 0; i < n; i++) {
        float off = (float)groups[kv[i].idx] * (maxc + 1.0f);
        for (int c = 0; c < 4; c++) sb[4 * i + c] = boxes[4 * kv[i].idx + c] + off;
    }
    int* k2 = (int*)malloc(sizeof(int) * n);
    int nk = nms_sorted(sb, n, thr, max_keep, k2);
    for (int i = 0; i < nk; i++) keep[i] = kv[k2[i]].idx;
    free(kv); free(sb); free(k2);
    return nk;
}

/* -------------------------------------------------------------------------------------
 * RPN proposals for one image (frcnn_ll.py:284-321 filter_proposals, :323-374 forward).
 * head[l]: NHWC [h][w][hc] with channel a = objectness logit of anchor a, channel A+4a+j = delta j.
 * Output: proposals [<=post_n][4]; returns count.
 * ------------------------------------------------------------------------------------- */
ORC_API int orc_rpn_proposals(int L, const float* const* head, const int* fh, const int* fw, int hc, int A,
                              const float* base_anchors /*[L][A][4]*/, int Hp, int Wp, int Hr, int Wr,
                              int pre_n, int post_n, float nms_thr, float min_size,
                              float* props, float* prop_scores) {
    int total = 0;
    for (int l = 0; l < L; l++) { int n = fh[l] * fw[l] * A; total += n < pre_n ? n : pre_n; }
    float* cb = (float*)malloc(sizeof(float) * 4 * total);
    float* cs = (float*)malloc(sizeof(float) * total);
    int* cl = (int*)malloc(sizeof(int) * total);
    int nc = 0;
    for (int l = 0; l < L; l++) {
        int n = fh[l] * fw[l] * A, k = n < pre_n ? n : pre_n;
        int sth = Hp / fh[l], stw = Wp / fw[l];
        orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * n);
        for (int i = 0; i < n; i++) { kv[i].key = head[l][(size_t)(i / A) * hc + (i % A)]; kv[i].idx = i; }
        qsort(kv, n, sizeof(orc_kv), kv_cmp);
        for (int t = 0; t < k; t++) {
            int i = kv[t].idx, a = i % A, pix = i / A, y = pix / fw[l], x = pix % fw[l];
            const float* ba = base_anchors + ((size_t)l * A + a) * 4;
            float anchor[4] = {(float)(x * stw) + ba[0], (float)(y * sth) + ba[1], (float)(x * stw) + ba[2], (float)(y * sth) + ba[3]};
            float box[4];
            box_decode(anchor, head[l] + (size_t)pix * hc + A + 4 * a, 1.0f, 1.0f, 1.0f, 1.0f, (float)BBOX_XFORM_CLIP, box);
            box[0] = clampf(box[0], 0.0f, (float)Wr); box[2] = clampf(box[2], 0.0f, (float)Wr);
            box[1] = clampf(box[1], 0.0f, (float)Hr); box[3] = clampf(box[3], 0.0f, (float)Hr);
            if ((box[2] - box[0]) >= min_size && (box[3] - box[1]) >= min_size) {
                memcpy(cb + 4 * nc, box, sizeof(box)); cs[nc] = kv[t].key; cl[nc] = l; nc++;
            }
        }
        free(kv);
    }
    int* keep = (int*)malloc(sizeof(int) * (nc + 1));
    int nk = orc_batched_nms(cb, cs, cl, nc, nms_thr, post_n, keep);
    for (int i = 0; i < nk; i++) { memcpy(props + 4 * i, cb + 4 * keep[i], 4 * sizeof(float)); if (prop_scores) prop_scores[i] = cs[keep[i]]; }
    free(cb); free(cs); free(cl); free(keep);
    return nk;
}

/* -------------------------------------------------------------------------------------
 * MultiScaleRoIAlign (7x7, sampling 2, aligned=False) on NHWC levels P2..P5.
 * Output layout [R][49][C]  (bin-major, channel-minor): this is the K order of the fc6 chain.
 * ------------------------------------------------------------------------------------- */
ORC_API int orc_roi_level(const float* box) {
    float area = (box[2] - box[0]) * (box[3] - box[1]);
    float s = sqrtf(area);
    float k = floorf((4.0f + orc_log2f(s / 224.0f)) + 1e-6f);
    if (!(k >= 2.0f)) k = 2.0f;           /* clamp(min=2); NaN -> 2 */
    if (k > 5.0f) k = 5.0f;
    return (int)k - 2;
}
ORC_API void orc_roi_align(int L, const float* const* feat, const int* fh, const int* fw, int C,
                           const float* rois, int R, float* out) {
    const int PH = 7, PW = 7, SR = 2;
#pragma omp parallel for schedule(dynamic, 4) num_threads(orc_threads)
    for (int r = 0; r < R; r++) {
        const float* box = rois + 4 * r;
        int l = orc_roi_level(box); if (l > L - 1) l = L - 1;
        float scale = 1.0f / (float)(4 << l);
        int Hf = fh[l], Wf = fw[l];
        const float* f = feat[l];
        float x1 = box[0] * scale, y1 = box[1] * scale, x2 = box[2] * scale, y2 = box[3] * scale;
        float rw = x2 - x1; if (!(rw >= 1.0f)) rw = 1.0f;
        float rh = y2 - y1; if (!(rh >= 1.0f)) rh = 1.0f;
        float bw = rw / (float)PW, bh = rh / (float)PH;
        for (int ph = 0; ph < PH; ph++)
            for (int pw = 0; pw < PW; pw++) {
                float* o = out + ((size_t)r * 49 + ph * 7 + pw) * C;
                for (int c = 0; c < C; c++) o[c] = 0.0f;
                for (int iy = 0; iy < SR; iy++) {
                    float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)SR;
                    for (int ix = 0; ix < SR; ix++) {
                        float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)SR;
                        float w1, w2, w3, w4; int yl, xl, yh, xh;
                        if (y < -1.0f || y > (float)Hf || x < -1.0f || x > (float)Wf) {
                            w1 = w2 = w3 = w4 = 0.0f; yl = xl = yh = xh = 0;
                        } else {
                            float yy = y <= 0.0f ? 0.0f : y, xx = x <= 0.0f ? 0.0f : x;
                            yl = (int)yy; xl = (int)xx;
                            if (yl >= Hf - 1) { yh = yl = Hf - 1; yy = (float)yl; } else yh = yl + 1;
                            if (xl >= Wf - 1) { xh = xl = Wf - 1; xx = (float)xl; } else xh = xl + 1;
                            float ly = yy - (float)yl, lx = xx - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
                            w1 = hy * hx; w2 = hy * lx; w3 = ly * hx; w4 = ly * lx;
                        }
                        const float* p1 = f + ((size_t)yl * Wf + xl) * C;
                        const float* p2 = f + ((size_t)yl * Wf + xh) * C;
                        const float* p3 = f + ((size_t)yh * Wf + xl) * C;
                        const float* p4 = f + ((size_t)yh * Wf + xh) * C;
                        for (int c = 0; c < C; c++)
                            o[c] = o[c] + (((w1 * p1[c] + w2 * p2[c]) + w3 * p3[c]) + w4 * p4[c]);
                    }
                }
                for (int c = 0; c < C; c++) o[c] = o[c] / 4.0f;
            }
    }
}

/* -------------------------------------------------------------------------------------
 * RoIHeads.postprocess_detections (frcnn_la.py:32-87) + transform.postprocess (:292-315).
 * logits [R][C], deltas [R][4C] (class-major), proposals [R][4] in resized-image coordinates.
 * Outputs (<= det_max rows): boxes/props scaled back to the original image, scores, labels,
 * prob_max, scores_cls [n][C].  Returns n.
 * ------------------------------------------------------------------------------------- */
/* frcnn_la.py:307-315 resize_boxes (called by GeneralizedRCNNTransform.postprocess :292-304 on `boxes` and `props`): the ratios are
 * python floats (float64 quotients of the integer sizes); multiplying a float32 tensor by a python float rounds the scalar to float32
 * first, then multiplies in float32.  Pinned to the executed reference function by tests/golden/resize_boxes.npz. */
ORC_API void orc_resize_boxes(const float* boxes, int n, int Hr, int Wr, int Ho, int Wo, float* out) {
    const float rh = (float)((double)Ho / (double)Hr), rw = (float)((double)Wo / (double)Wr);
    for (int i = 0; i < n; i++) {
        out[4 * i + 0] = boxes[4 * i + 0] * rw; out[4 * i + 1] = boxes[4 * i + 1] * rh;
        out[4 * i + 2] = boxes[4 * i + 2] * rw; out[4 * i + 3] = boxes[4 * i + 3] * rh;
    }
}

ORC_API int orc_frcnn_postprocess(int R, int C, const float* logits, const float* deltas, const float* proposals,
                                  int Hr, int Wr, int Ho, int Wo, float score_thr, float nms_thr, int det_max,
                                  float* o_boxes, float* o_scores, int64_t* o_labels, float* o_props,
                                  float* o_pm, float* o_scls) {
    float* prob = (float*)malloc(sizeof(float) * (size_t)R * C);
    float* pmax = (float*)malloc(sizeof(float) * (size_t)R);
    for (int r = 0; r < R; r++) {
        const float* lg = logits + (size_t)r * C;
        float m = lg[0];
        for (int c = 1; c < C; c++) if (lg[c] > m) m = lg[c];
        float s = 0.0f;
        for (int c = 0; c < C; c++) { float e = orc_expf(lg[c] - m); prob[(size_t)r * C + c] = e; s = s + e; }
        for (int c = 0; c < C; c++) prob[(size_t)r * C + c] = prob[(size_t)r * C + c] / s;
        float pm = prob[(size_t)r * C + 1];
        for (int c = 2; c < C; c++) if (prob[(size_t)r * C + c] > pm) pm = prob[(size_t)r * C + c];
        pmax[r] = pm;
    }
    int cap = R * (C - 1);
    float* cb = (float*)malloc(sizeof(float) * 4 * (size_t)cap);
    float* cs = (float*)malloc(sizeof(float) * (size_t)cap);
    int* cg = (int*)malloc(sizeof(int) * (size_t)cap);
    int* cr = (int*)malloc(sizeof(int) * (size_t)cap);
    int nc = 0;
    for (int r = 0; r < R; r++)
        for (int c = 1; c < C; c++) {
            float sc = prob[(size_t)r * C + c];
            if (!(sc > score_thr)) continue;
            float box[4];
            box_decode(proposals + 4 * r, deltas + (size_t)r * 4 * C + 4 * c, 10.0f, 10.0f, 5.0f, 5.0f, (float)BBOX_XFORM_CLIP, box);
            box[0] = clampf(box[0], 0.0f, (float)Wr); box[2] = clampf(box[2], 0.0f, (float)Wr);
            box[1] = clampf(box[1], 0.0f, (float)Hr); box[3] = clampf(box[3], 0.0f, (float)Hr);
            memcpy(cb + 4 * (size_t)nc, box, sizeof(box)); cs[nc] = sc; cg[nc] = c; cr[nc] = r; nc++;
        }
    int* keep = (int*)malloc(sizeof(int) * ((size_t)nc + 1));
    int nk = orc_batched_nms(cb, cs, cg, nc, nms_thr, det_max, keep);
    for (int i = 0; i < nk; i++) {
        int k = keep[i], r = cr[k];
        orc_resize_boxes(cb + 4 * k, 1, Hr, Wr, Ho, Wo, o_boxes + 4 * i);
        orc_resize_boxes(proposals + 4 * r, 1, Hr, Wr, Ho, Wo, o_props + 4 * i);
        o_scores[i] = cs[k]; o_labels[i] = cg[k]; o_pm[i] = pmax[r];
        memcpy(o_scls + (size_t)i * C, prob + (size_t)r * C, sizeof(float) * C);
    }
    free(prob); free(pmax); free(cb); free(cs); free(cg); free(cr); free(keep);
    return nk;
}

/* -------------------------------------------------------------------------------------
 * RetinaNet.postprocess_detections (detection/retinanet_cal.py:402-490) + the stock
 * GeneralizedRCNNTransform.postprocess (boxes only, float32 tensor ratios).
 * cls[l]: NHWC [h][w][A*K] logits (channel a*K+k), reg[l]: [h][w][A*4] (channel a*4+j).
 * Outputs hold up to K*per_class rows, grouped by class in class order (NOT globally sorted).
 * ------------------------------------------------------------------------------------- */
ORC_API int orc_retina_postprocess(int L, const float* const* cls, const float* const* reg, const int* fh, const int* fw,
                                   int A, int K, const float* base_anchors /*[L][A][4]*/, int Hp, int Wp, int Hr, int Wr,
                                   int Ho, int Wo, float score_thr, float nms_thr, int per_class, float min_box,
                                   float* o_boxes, float* o_scores, int64_t* o_labels, float* o_pm, float* o_scls) {
    long total = 0;
    for (int l = 0; l < L; l++) total += (long)fh[l] * fw[l] * A;
    float* sc = (float*)malloc(sizeof(float) * (size_t)total * K);
    float* bx = (float*)malloc(sizeof(float) * (size_t)total * 4);
    float* pm = (float*)malloc(sizeof(float) * (size_t)total);
    long base = 0;
    for (int l = 0; l < L; l++) {
        int sth = Hp / fh[l], stw = Wp / fw[l];
        long n = (long)fh[l] * fw[l] * A;
#pragma omp parallel for schedule(static) num_threads(orc_threads)
        for (long i = 0; i < n; i++) {
            int a = (int)(i % A); long pix = i / A; int y = (int)(pix / fw[l]), x = (int)(pix % fw[l]);
            const float* lg = cls[l] + (size_t)pix * A * K + (size_t)a * K;
            float m = 0.0f;
            for (int k = 0; k < K; k++) { float s = orc_sigmoidf(lg[k]); sc[(size_t)(base + i) * K + k] = s; if (k == 0 || s > m) m = s; }
            pm[base + i] = m;
            const float* ba = base_anchors + ((size_t)l * A + a) * 4;
            float anchor[4] = {(float)(x * stw) + ba[0], (float)(y * sth) + ba[1], (float)(x * stw) + ba[2], (float)(y * sth) + ba[3]};
            float b[4];
            box_decode(anchor, reg[l] + (size_t)pix * A * 4 + (size_t)a * 4, 1.0f, 1.0f, 1.0f, 1.0f, (float)BBOX_XFORM_CLIP, b);
            b[0] = clampf(b[0], 0.0f, (float)Wr); b[2] = clampf(b[2], 0.0f, (float)Wr);
            b[1] = clampf(b[1], 0.0f, (float)Hr); b[3] = clampf(b[3], 0.0f, (float)Hr);
            memcpy(bx + 4 * (size_t)(base + i), b, sizeof(b));
        }
        base += n;
    }
    float rh = (float)Ho / (float)Hr, rw = (float)Wo / (float)Wr;   /* stock resize_boxes: float32 tensor division */
    orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * (size_t)total);
    float* sb = (float*)malloc(sizeof(float) * 4 * (size_t)total);
    int* keep = (int*)malloc(sizeof(int) * ((size_t)total + 1));
    int nout = 0;
    for (int k = 0; k < K; k++) {
        int nc = 0;
        for (long i = 0; i < total; i++) {
            float s = sc[(size_t)i * K + k];
            if (!(s > score_thr)) continue;
            const float* b = bx + 4 * (size_t)i;
            if (!((b[2] - b[0]) >= min_box && (b[3] - b[1]) >= min_box)) continue;
            kv[nc].key = s; kv[nc].idx = (int)i; nc++;
        }
        qsort(kv, nc, sizeof(orc_kv), kv_cmp);
        for (int i = 0; i < nc; i++) memcpy(sb + 4 * (size_t)i, bx + 4 * (size_t)kv[i].idx, 4 * sizeof(float));
        int nk = nms_sorted(sb, nc, nms_thr, per_class, keep);
        for (int i = 0; i < nk; i++) {
            int ai = kv[keep[i]].idx;
            const float* b = bx + 4 * (size_t)ai;
            o_boxes[4 * (size_t)nout + 0] = b[0] * rw; o_boxes[4 * (size_t)nout + 1] = b[1] * rh;
            o_boxes[4 * (size_t)nout + 2] = b[2] * rw; o_boxes[4 * (size_t)nout + 3] = b[3] * rh;
            o_scores[nout] = sc[(size_t)ai * K + k]; o_labels[nout] = k; o_pm[nout] = pm[ai];
            memcpy(o_scls + (size_t)nout * K, sc + (size_t)ai * K, sizeof(float) * K);
            nout++;
        }
    }
    free(sc); free(bx); free(pm); free(kv); free(sb); free(keep);
    return nout;
}


/* -------------------------------------------------------------------------------------
 * PIL Image.rotate(angle, expand=True) with the default NEAREST resampling
 * (cald_helper.py:153): Image.rotate's matrix arithmetic (python floats, round(.,15)) followed
 * by Pillow's 16.16 fixed-point nearest-neighbour affine loop (Geometry.c affine_fixed).
 * ------------------------------------------------------------------------------------- */
#include <stdio.h>
static double py_round15(double v) { char buf[64]; snprintf(buf, sizeof(buf), "%.15f", v); return strtod(buf, NULL); }
static int pil_floor_(double v) { return v < 0.0 ? (int)floor(v) : (int)v; }
static int pil_fix(double v) { return pil_floor_(v * 65536.0 + 0.5); }
ORC_API void orc_pil_rotate_matrix(int H, int W, double angle_deg, double* m /*6*/, int* nH, int* nW) {
    double angle = fmod(angle_deg, 360.0); if (angle < 0) angle += 360.0;
    double w = (double)W, h = (double)H, cx = w / 2.0, cy = h / 2.0;
    double ang = -(angle * (3.141592653589793 / 180.0));
    m[0] = py_round15(cos(ang)); m[1] = py_round15(sin(ang)); m[2] = 0.0;
    m[3] = py_round15(-sin(ang)); m[4] = py_round15(cos(ang)); m[5] = 0.0;
    double tx = -cx, ty = -cy;
    double m2 = m[0] * tx + m[1] * ty + m[2], m5 = m[3] * tx + m[4] * ty + m[5];
    m[2] = m2 + cx; m[5] = m5 + cy;
    double xs[4] = {0, w, w, 0}, ys[4] = {0, 0, h, h}, xmin = 0, xmax = 0, ymin = 0, ymax = 0;
    for (int i = 0; i < 4; i++) {
        double X = m[0] * xs[i] + m[1] * ys[i] + m[2], Y = m[3] * xs[i] + m[4] * ys[i] + m[5];
        if (i == 0 || X < xmin) xmin = X; if (i == 0 || X > xmax) xmax = X;
        if (i == 0 || Y < ymin) ymin = Y; if (i == 0 || Y > ymax) ymax = Y;
    }
    int nw = (int)ceil(xmax) - (int)floor(xmin), nh = (int)ceil(ymax) - (int)floor(ymin);
    double px = -(nw - W) / 2.0, py = -(nh - H) / 2.0;
    m2 = m[0] * px + m[1] * py + m[2]; m5 = m[3] * px + m[4] * py + m[5];
    m[2] = m2; m[5] = m5;
    *nH = nh; *nW = nw;
}
ORC_API void orc_pil_affine_nearest(const uint8_t* src, int H, int W, const double* a, uint8_t* dst, int oh, int ow) {
    int a0 = pil_fix(a[0]), a1 = pil_fix(a[1]), a3 = pil_fix(a[3]), a4 = pil_fix(a[4]);
    int a2 = pil_fix(a[2] + a[0] * 0.5 + a[1] * 0.5), a5 = pil_fix(a[5] + a[3] * 0.5 + a[4] * 0.5);
    memset(dst, 0, (size_t)oh * ow * 3);
    for (int y = 0; y < oh; y++) {
        int xx = a2, yy = a5;
        for (int x = 0; x < ow; x++) {
            int xin = xx >> 16;
            if (xin >= 0 && xin < W) {
                int yin = yy >> 16;
                if (yin >= 0 && yin < H) memcpy(dst + ((size_t)y * ow + x) * 3, src + ((size_t)yin * W + xin) * 3, 3);
            }
            xx += a0; yy += a3;
        }
        a2 += a1; a5 += a4;
    }
}


/* -------------------------------------------------------------------------------------
 * torch.rand on the CPU generator after torch.manual_seed(seed): MT19937 seeded with
 * init_genrand(seed & 0xffffffff), one 32-bit draw per element, value = (r & 0xffffff) * 2^-24.
 * SaltPepperNoise (cald_helper.py:78-85) on a uint8 image: noise has the CHW layout of
 * to_tensor(image); salt / pepper are the image's global max / min.
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_torch_rand(uint64_t seed, int n, float* out) {
    orc_mt s; mt_init_genrand(&s, (uint32_t)(seed & 0xffffffffu));
    for (int i = 0; i < n; i++) out[i] = (float)((double)(mt_next(&s) & 0xffffffu) * (1.0 / 16777216.0));
}
ORC_API void orc_salt_pepper(const uint8_t* src, int H, int W, float prob, uint64_t seed, uint8_t* dst) {
    orc_mt s; mt_init_genrand(&s, (uint32_t)(seed & 0xffffffffu));
    uint8_t mx = src[0], mn = src[0];
    for (size_t i = 0; i < (size_t)H * W * 3; i++) { if (src[i] > mx) mx = src[i]; if (src[i] < mn) mn = src[i]; }
    const float lo = (float)((double)prob / 2.0), hi = (float)(1.0 - (double)prob / 2.0);
    memcpy(dst, src, (size_t)H * W * 3);
    for (int c = 0; c < 3; c++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                float u = (float)((double)(mt_next(&s) & 0xffffffu) * (1.0 / 16777216.0));
                size_t o = ((size_t)y * W + x) * 3 + c;
                if (u < lo) dst[o] = mx;
                if (u > hi) dst[o] = mn;
            }
}

/* cald_helper.py:135-223 rotate(): box corners through the affine matrix (float32 after
 * .float()), axis-aligned hull, rescale by (rotated size / original size), clamp.
 * nW2/nH2 are the PIL-expanded image sizes (new_image.width/height). */
ORC_API void orc_rotate_boxes(const float* boxes, int N, int H, int W, double angle_deg, int pilW, int pilH, float* out) {
    double ang = angle_deg * (3.141592653589793 / 180.0);      /* np.radians */
    double alpha = cos(ang), beta = sin(ang), cx = W / 2.0, cy = H / 2.0;
    double m02 = (1 - alpha) * cx - beta * cy, m12 = beta * cx + (1 - alpha) * cy;
    double c_ = fabs(alpha), s_ = fabs(beta);
    int nW = (int)((H * s_) + (W * c_)), nH = (int)((H * c_) + (W * s_));
    m02 += (nW / 2.0) - cx; m12 += (nH / 2.0) - cy;
    float a00 = (float)alpha, a01 = (float)beta, a02 = (float)m02, a10 = (float)(-beta), a11 = (float)alpha, a12 = (float)m12;
    float sx = (float)((double)pilW / (double)W), sy = (float)((double)pilH / (double)H);
    for (int i = 0; i < N; i++) {
        const float* b = boxes + 4 * i;
        float bw = b[2] - b[0], bh = b[3] - b[1];
        float xs[4] = {b[0], b[0] + bw, b[0], b[2]}, ys[4] = {b[1], b[1], b[1] + bh, b[3]};
        float xmin = 0, xmax = 0, ymin = 0, ymax = 0;
        for (int k = 0; k < 4; k++) {
            float X = (a00 * xs[k] + a01 * ys[k]) + a02 * 1.0f;
            float Y = (a10 * xs[k] + a11 * ys[k]) + a12 * 1.0f;
            if (k == 0 || X < xmin) xmin = X; if (k == 0 || X > xmax) xmax = X;
            if (k == 0 || Y < ymin) ymin = Y; if (k == 0 || Y > ymax) ymax = Y;
        }
        float r[4] = {xmin / sx, ymin / sy, xmax / sx, ymax / sy};
        out[4 * i + 0] = clampf(r[0], 0.0f, (float)W); out[4 * i + 1] = clampf(r[1], 0.0f, (float)H);
        out[4 * i + 2] = clampf(r[2], 0.0f, (float)W); out[4 * i + 3] = clampf(r[3], 0.0f, (float)H);
    }
}


/* -------------------------------------------------------------------------------------
 * GaussianNoise (cald_helper.py:72-75): image + torch.randn(size) * std / 255.0.
 * torch.randn on the CPU generator (>= 16 elements): the tensor is first filled with uniforms from
 * the MT19937 stream, then every 16-chunk is transformed by Box-Muller (elements j and j+8 pair
 * up); if the size is not a multiple of 16 the last 16 entries are refilled from 16 NEW uniforms and
 * transformed.  log / sin / cos are the oracle's deterministic float32 versions (torch uses its
 * vector math library; agreement ~1e-6).  out[i] = (randn[i] * std) / 255.
 * ------------------------------------------------------------------------------------- */
static void normal_fill_16(const float* u, float* o, float std) {
    for (int j = 0; j < 8; j++) {
        float u1 = 1.0f - u[j], u2 = u[j + 8];
        float radius = sqrtf(-2.0f * orc_logf(u1));
        float theta = 6.283185307179586f * u2, sn, cs;
        orc_sincosf(theta, &sn, &cs);
        o[j] = ((radius * cs) * std) / 255.0f;
        o[j + 8] = ((radius * sn) * std) / 255.0f;
    }
}
ORC_API void orc_gaussian_noise(uint64_t seed, int n, float std, float* out) {
    orc_mt s; mt_init_genrand(&s, (uint32_t)(seed & 0xffffffffu));
    float* u = (float*)malloc(sizeof(float) * ((size_t)n + 16));
    for (int i = 0; i < n; i++) u[i] = (float)((double)(mt_next(&s) & 0xffffffu) * (1.0 / 16777216.0));
    for (int i = 0; i + 15 < n; i += 16) normal_fill_16(u + i, out + i, std);
    if (n % 16 != 0 && n >= 16) {
        float v[16];
        for (int i = 0; i < 16; i++) v[i] = (float)((double)(mt_next(&s) & 0xffffffu) * (1.0 / 16777216.0));
        normal_fill_16(v, out + n - 16, std);
    }
    free(u);
}


/* -------------------------------------------------------------------------------------
 * the baseline sweeps that share the detector forward.
 * lt_c_train.py:92-121  uncertainty = min(1, |calcu_iou(box, prop) + prob_max - 1|) over detections
 * (calcu_iou's +1 convention incl. its asymmetric area, lt_c_train.py:92-103).
 * ------------------------------------------------------------------------------------- */
ORC_API float orc_lt_uncertainty(int n, const float* boxes, const float* props, const float* pm) {
    float unc = 1.0f;
    for (int i = 0; i < n; i++) {
        const float *A = boxes + 4 * i, *B = props + 4 * i;
        float width = (fminf_(A[2], B[2]) - fmaxf_(A[0], B[0])) + 1.0f;
        float height = (fminf_(A[3], B[3]) - fmaxf_(A[1], B[1])) + 1.0f;
        float iou = 0.0f;
        if (!(width <= 0.0f || height <= 0.0f)) {
            float Aarea = (A[2] - A[0]) * ((A[3] - A[1]) + 1.0f);
            float Barea = (B[2] - B[0]) * ((B[3] - B[1]) + 1.0f);
            float iner = width * height;
            iou = iner / ((Aarea + Barea) - iner);
        }
        float u = fabsf((iou + pm[i]) - 1.0f);
        if (u < unc) unc = u;
    }
    return unc;
}

/* ls_c_train.py:136-150: max IoU of each reference box against one view's detections (0 when the view is empty) */
ORC_API void orc_max_iou_rows(int N, const float* ref_boxes, int M, const float* boxes, float* out) {
    for (int i = 0; i < N; i++) {
        float best = 0.0f;
        for (int k = 0; k < M; k++) {
            float v = cald_iou(ref_boxes + 4 * i, boxes + 4 * k);
            if (k == 0 || v > best || v != v) best = v;
            if (best != best) break;
        }
        out[i] = best;
    }
}

/* torch.randn called nseg times in a row on the same generator (ls_c_train.py:129-131: GaussianNoise(image, i*8),
 * i = 1..6): the MT19937 stream simply continues from one call to the next. */
ORC_API void orc_gaussian_noise_seq(uint64_t seed, int n, int nseg, const float* stds, float* out /*[nseg][n]*/) {
    orc_mt s; mt_init_genrand(&s, (uint32_t)(seed & 0xffffffffu));
    float* u = (float*)malloc(sizeof(float) * ((size_t)n + 16));
    for (int g = 0; g < nseg; g++) {
        float* o = out + (size_t)g * n;
        for (int i = 0; i < n; i++) u[i] = (float)((double)(mt_next(&s) & 0xffffffu) * (1.0 / 16777216.0));
        for (int i = 0; i + 15 < n; i += 16) normal_fill_16(u + i, o + i, stds[g]);
        if (n % 16 != 0 && n >= 16) {
            float v[16];
            for (int i = 0; i < 16; i++) v[i] = (float)((double)(mt_next(&s) & 0xffffffu) * (1.0 / 16777216.0));
            normal_fill_16(v, o + n - 16, stds[g]);
        }
    }
    free(u);
}


/* One torch CPU generator consumed by a sequence of torch.randn(n) (kind 0, scaled: * std / 255) and
 * torch.rand(n) (kind 1) calls: get_uncertainty draws GaussianNoise and SaltPepperNoise views of one image from
 * the same global generator, in call order (cald_train.py:127-157). */
ORC_API void orc_torch_stream(uint64_t seed, int nops, const int* kinds, int n, const float* stds, float* out /*[nops][n]*/) {
    orc_mt s; mt_init_genrand(&s, (uint32_t)(seed & 0xffffffffu));
    float* u = (float*)malloc(sizeof(float) * ((size_t)n + 16));
    for (int g = 0; g < nops; g++) {
        float* o = out + (size_t)g * n;
        if (kinds[g] == 1) {
            for (int i = 0; i < n; i++) o[i] = (float)((double)(mt_next(&s) & 0xffffffu) * (1.0 / 16777216.0));
            continue;
        }
        for (int i = 0; i < n; i++) u[i] = (float)((double)(mt_next(&s) & 0xffffffu) * (1.0 / 16777216.0));
        for (int i = 0; i + 15 < n; i += 16) normal_fill_16(u + i, o + i, stds[g]);
        if (n % 16 != 0 && n >= 16) {
            float v[16];
            for (int i = 0; i < 16; i++) v[i] = (float)((double)(mt_next(&s) & 0xffffffu) * (1.0 / 16777216.0));
            normal_fill_16(v, o + n - 16, stds[g]);
        }
    }
    free(u);
}

/* cald_helper.ColorAdjust (cald_helper.py:65-69): torchvision F.adjust_brightness / _contrast / _saturation on a
 * PIL image = PIL.ImageEnhance.{Brightness, Contrast, Color}(img).enhance(factor) = Image.blend(degenerate, img,
 * factor).  Pillow's ImagingBlend for alpha outside [0, 1]: temp = (float)(in1 + alpha * (in2 - in1)) in C float
 * arithmetic, clipped to [0, 255] and truncated; inside [0, 1] the same expression truncated without clipping.
 * Degenerates: black; the rounded mean of convert('L') (L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16);
 * convert('L') of the image itself. */
static uint8_t pil_blend1(int in1, int in2, float alpha) {
    volatile float prod = alpha * (float)(in2 - in1);
    float temp = (float)in1 + prod;
    if (alpha >= 0.0f && alpha <= 1.0f) return (uint8_t)temp;
    if (temp <= 0.0f) return 0;
    if (temp >= 255.0f) return 255;
    return (uint8_t)temp;
}
static int pil_l(const uint8_t* p) { return (p[0] * 19595 + p[1] * 38470 + p[2] * 7471 + 0x8000) >> 16; }
ORC_API void orc_color_adjust(const uint8_t* src, int H, int W, float factor, uint8_t* dst) {
    const size_t npx = (size_t)H * W;
    uint8_t* a = (uint8_t*)malloc(npx * 3);
    uint8_t* b = (uint8_t*)malloc(npx * 3);
    /* brightness */
    if (factor == 1.0f) memcpy(a, src, npx * 3);
    else for (size_t i = 0; i < npx * 3; i++) a[i] = factor == 0.0f ? 0 : pil_blend1(0, src[i], factor);
    /* contrast */
    double sum = 0.0;
    for (size_t i = 0; i < npx; i++) sum += (double)pil_l(a + 3 * i);
    const int mean = (int)(sum / (double)npx + 0.5);
    if (factor == 1.0f) memcpy(b, a, npx * 3);
    else for (size_t i = 0; i < npx * 3; i++) b[i] = factor == 0.0f ? (uint8_t)mean : pil_blend1(mean, a[i], factor);
    /* saturation */
    for (size_t i = 0; i < npx; i++) {
        const int l = pil_l(b + 3 * i);
        for (int c = 0; c < 3; c++)
            dst[3 * i + c] = factor == 1.0f ? b[3 * i + c] : (factor == 0.0f ? (uint8_t)l : pil_blend1(l, b[3 * i + c], factor));
    }
    free(a); free(b);
}
