// host_logic.h -- host-side restatements of the reference's non-GPU logic on the sweep path (no device work):
//   * detector-transform size rule (torchvision GeneralizedRCNNTransform)
//   * Python's `random` (MT19937: random.seed(int), random.random(), random.uniform) and cald_helper.cutout's rectangle
//     selection (cald/cald_helper.py:88-132)
//   * np.round(np.linspace(0, n-1, 50)) sub-sampling (cald_train.py:110-113) and numpy's pairwise float64 / float32 sums
//   * Pillow's resampling coefficients (Resample.c precompute_coeffs / normalize_coeffs_8bpc; BILINEAR and BICUBIC)
//   * Pillow's Image.rotate(expand=True) matrix + 16.16 fixed-point coefficients, and cald_helper.rotate's box constants
// Product code; the oracle holds its own independent restatement of each of these.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static inline void transform_size(int H, int W, int min_size, int max_size, int* Hr, int* Wr, int* Hp, int* Wp) {
    double mn = (double)(H < W ? H : W), mx = (double)(H > W ? H : W);
    double scale = (double)min_size / mn;
    if (mx * scale > (double)max_size) scale = (double)max_size / mx;
    *Hr = (int)std::floor((double)H * scale);
    *Wr = (int)std::floor((double)W * scale);
    *Hp = ((*Hr + 31) / 32) * 32;
    *Wp = ((*Wr + 31) / 32) * 32;
}

// ---- Python's random module: MT19937 + random.seed(int) + random.random() ----
struct PyRandom {
    uint32_t mt[624]; int idx;
    void init_genrand(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    void seed(uint64_t a) {
        uint32_t key[2] = {(uint32_t)(a & 0xffffffffu), (uint32_t)(a >> 32)};
        int klen = key[1] ? 2 : 1;
        init_genrand(19650218u);
        int i = 1, j = 0;
        for (int k = 624; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            i++; j++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
            if (j >= klen) j = 0;
        }
        for (int k = 623; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            i++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int k = 0; k < 624; k++) {
                uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
        return y;
    }
    double random() { uint32_t a = next() >> 5, b = next() >> 6; return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0); }
    double uniform(double a, double b) { return a + (b - a) * random(); }
    // random.randint(0, n - 1) == randrange(n): _randbelow_with_getrandbits (k = n.bit_length(), rejection sampling)
    int randbelow(int n) {
        int k = 0;
        for (int t = n; t; t >>= 1) k++;
        for (;;) { const uint32_t r = next() >> (32 - k); if ((int)r < n) return (int)r; }
    }
};

// cald_helper.cutout (cald/cald_helper.py:88-132): rectangle selection only; the fill happens in
// the preprocess kernel.  boxes: sub-sampled reference detections, original image coordinates.
// The generator is the caller's: get_uncertainty draws ColorSwap's randint and every cutout call of one image from the
// same global Python generator, in call order (cald_train.py:140-166).
// margin (optional): the smallest distance of any trial's largest overlap ratio to the two accept thresholds (0.4, 0.1) -- the only
// place where the reference boxes decide something discrete here (audit.hip).
static inline int cutout_rects(PyRandom& rng, int H, int W, int N, const float* boxes, int cut_num, int* rects, float* margin = nullptr) {
    int count = 0;
    if (margin) *margin = INFINITY;
    for (int t = 0; t < 50; t++) {
        double sh = rng.uniform(0.05 * H, 0.2 * H);
        double sw = rng.uniform(0.05 * W, 0.2 * W);
        double left = rng.uniform(0.0, (double)W - sw), right = left + sw;
        double top = rng.uniform(0.0, (double)H - sh), bottom = top + sh;
        int il = (int)left, it = (int)top, ir = (int)right, ib = (int)bottom;
        float c[4] = {(float)il, (float)it, (float)ir, (float)ib};
        float rmax = 0.0f; bool any_nan = false;
        for (int i = 0; i < N; i++) {
            const float* b = boxes + 4 * i;
            float iw = std::fmin(c[2], b[2]) - std::fmax(c[0], b[0]); if (iw < 0.0f) iw = 0.0f;
            float ih = std::fmin(c[3], b[3]) - std::fmax(c[1], b[1]); if (ih < 0.0f) ih = 0.0f;
            float area = (b[2] - b[0]) * (b[3] - b[1]);
            float ratio = (iw * ih) / area;
            if (ratio != ratio) any_nan = true;
            if (i == 0 || ratio > rmax) rmax = ratio;
        }
        if (margin && !any_nan) { const float d = std::fmin(std::fabs(rmax - 0.4f), std::fabs(rmax - 0.1f)); if (d < *margin) *margin = d; }
        if (!any_nan && (rmax > 0.4f || rmax < 0.1f)) continue;
        rects[4 * count] = il; rects[4 * count + 1] = it; rects[4 * count + 2] = ir; rects[4 * count + 3] = ib;
        if (++count >= cut_num) break;
    }
    return count;
}
static inline int cutout_rects(uint64_t seed, int H, int W, int N, const float* boxes, int cut_num, int* rects) {
    PyRandom rng; rng.seed(seed);
    return cutout_rects(rng, H, W, N, boxes, cut_num, rects);
}

// np.round(np.linspace(0, n-1, 50)).astype(int)  (cald_train.py:110-113)
static inline int subsample_indices(int n, int* inds) {
    if (n <= 40) { for (int i = 0; i < n; i++) inds[i] = i; return n; }
    double step = (double)(n - 1) / 49.0;
    for (int i = 0; i < 50; i++) {
        double v = (i == 49) ? (double)(n - 1) : (double)i * step;
        inds[i] = (int)std::nearbyint(v);
    }
    return 50;
}

// numpy pairwise summation of float64 (np.mean over a 1-D array)
static inline double np_sum(const double* a, int n) {
    if (n < 8) { double r = 0.0; for (int i = 0; i < n; i++) r += a[i]; return r; }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; k++) r[k] = a[k];
        int i;
        for (i = 8; i < n - (n % 8); i += 8) for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2; n2 -= n2 % 8;
    return np_sum(a, n2) + np_sum(a + n2, n - n2);
}

// Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc; fid 0 = BILINEAR (support 1), 1 = BICUBIC (support 2, a = -0.5)
static inline double pil_filter(int fid, double x) {
    if (x < 0.0) x = -x;
    if (fid == 0) return x < 1.0 ? 1.0 - x : 0.0;
    if (x < 1.0) return ((-0.5 + 2.0) * x - (-0.5 + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * -0.5;
    return 0.0;
}
static inline int pil_coeffs(int inSize, int outSize, int fid, std::vector<int>& bounds, std::vector<int>& kk) {
    double scale = (double)inSize / (double)outSize, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    double support = (fid == 0 ? 1.0 : 2.0) * filterscale;
    int ksize = (int)std::ceil(support) * 2 + 1;
    std::vector<double> pre((size_t)outSize * ksize);
    bounds.assign((size_t)outSize * 2, 0); kk.assign((size_t)outSize * ksize, 0);
    for (int xx = 0; xx < outSize; xx++) {
        double center = (xx + 0.5) * scale, ww = 0.0, ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5); if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5); if (xmax > inSize) xmax = inSize;
        xmax -= xmin;
        double* k = &pre[(size_t)xx * ksize];
        int x;
        for (x = 0; x < xmax; x++) {
            double w = pil_filter(fid, (x + xmin - center + 0.5) * ss);
            k[x] = w; ww += w;
        }
        for (x = 0; x < xmax; x++) if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; x++) k[x] = 0;
        bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    }
    for (size_t i = 0; i < pre.size(); i++)
        kk[i] = pre[i] < 0 ? (int)(-0.5 + pre[i] * (double)(1 << 22)) : (int)(0.5 + pre[i] * (double)(1 << 22));
    return ksize;
}

// PIL Image.rotate(angle, expand=True): matrix arithmetic of Image.rotate (python floats, round(., 15)),
// then the FIX()ed 16.16 coefficients of Geometry.c affine_fixed.  (cald_helper.py:153)
static inline double py_round15(double v) { char buf[64]; snprintf(buf, sizeof(buf), "%.15f", v); return strtod(buf, nullptr); }
static inline int pil_fix(double v) { double t = v * 65536.0 + 0.5; return t < 0.0 ? (int)std::floor(t) : (int)t; }
static inline void pil_rotate_setup(int H, int W, double angle_deg, int fix[6], int* nH, int* nW) {
    double angle = std::fmod(angle_deg, 360.0); if (angle < 0) angle += 360.0;
    const double w = (double)W, h = (double)H, cx = w / 2.0, cy = h / 2.0;
    const double ang = -(angle * (3.141592653589793 / 180.0));
    double m[6] = {py_round15(std::cos(ang)), py_round15(std::sin(ang)), 0.0, py_round15(-std::sin(ang)), py_round15(std::cos(ang)), 0.0};
    double m2 = m[0] * -cx + m[1] * -cy + m[2], m5 = m[3] * -cx + m[4] * -cy + m[5];
    m[2] = m2 + cx; m[5] = m5 + cy;
    const double xs[4] = {0, w, w, 0}, ys[4] = {0, 0, h, h};
    double xmin = 0, xmax = 0, ymin = 0, ymax = 0;
    for (int i = 0; i < 4; i++) {
        const double X = m[0] * xs[i] + m[1] * ys[i] + m[2], Y = m[3] * xs[i] + m[4] * ys[i] + m[5];
        if (i == 0 || X < xmin) xmin = X; if (i == 0 || X > xmax) xmax = X;
        if (i == 0 || Y < ymin) ymin = Y; if (i == 0 || Y > ymax) ymax = Y;
    }
    const int nw = (int)std::ceil(xmax) - (int)std::floor(xmin), nh = (int)std::ceil(ymax) - (int)std::floor(ymin);
    const double px = -(nw - W) / 2.0, py = -(nh - H) / 2.0;
    m2 = m[0] * px + m[1] * py + m[2]; m5 = m[3] * px + m[4] * py + m[5];
    m[2] = m2; m[5] = m5;
    fix[0] = pil_fix(m[0]); fix[1] = pil_fix(m[1]); fix[3] = pil_fix(m[3]); fix[4] = pil_fix(m[4]);
    fix[2] = pil_fix(m[2] + m[0] * 0.5 + m[1] * 0.5); fix[5] = pil_fix(m[5] + m[3] * 0.5 + m[4] * 0.5);
    *nH = nh; *nW = nw;
}
// cald_helper.rotate box transform constants (cald_helper.py:135-222): float32 affine matrix, scale, clamp bounds
static inline void rotate_box_params(int H, int W, double angle_deg, int pilW, int pilH, float* p /*12*/) {
    const double ang = angle_deg * (3.141592653589793 / 180.0);
    const double alpha = std::cos(ang), beta = std::sin(ang), cx = W / 2.0, cy = H / 2.0;
    double m02 = (1 - alpha) * cx - beta * cy, m12 = beta * cx + (1 - alpha) * cy;
    const double c_ = std::fabs(alpha), s_ = std::fabs(beta);
    const int nW = (int)((H * s_) + (W * c_)), nH = (int)((H * c_) + (W * s_));
    m02 += (nW / 2.0) - cx; m12 += (nH / 2.0) - cy;
    p[0] = (float)alpha; p[1] = (float)beta; p[2] = (float)m02; p[3] = (float)(-beta); p[4] = (float)alpha; p[5] = (float)m12;
    p[6] = (float)((double)pilW / (double)W); p[7] = (float)((double)pilH / (double)H); p[8] = (float)W; p[9] = (float)H;
    p[10] = p[11] = 0.0f;
}

// cald_helper.rotate box transform (cald_helper.py:160-222) on the host, same float32 arithmetic as score.hip kind 3
static inline void rotate_boxes_host(const float* p /*rotate_box_params*/, const float* boxes, int N, float* out) {
    for (int i = 0; i < N; i++) {
        const float* b = boxes + 4 * i;
        const float bw = b[2] - b[0], bh = b[3] - b[1];
        const float xs[4] = {b[0], b[0] + bw, b[0], b[2]}, ys[4] = {b[1], b[1], b[1] + bh, b[3]};
        float xmin = 0.f, xmax = 0.f, ymin = 0.f, ymax = 0.f;
        for (int q = 0; q < 4; q++) {
            const float X = (p[0] * xs[q] + p[1] * ys[q]) + p[2] * 1.0f;
            const float Y = (p[3] * xs[q] + p[4] * ys[q]) + p[5] * 1.0f;
            if (q == 0 || X < xmin) xmin = X; if (q == 0 || X > xmax) xmax = X;
            if (q == 0 || Y < ymin) ymin = Y; if (q == 0 || Y > ymax) ymax = Y;
        }
        auto cl = [](float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); };
        out[4 * i] = cl(xmin / p[6], 0.0f, p[8]); out[4 * i + 1] = cl(ymin / p[7], 0.0f, p[9]);
        out[4 * i + 2] = cl(xmax / p[6], 0.0f, p[8]); out[4 * i + 3] = cl(ymax / p[7], 0.0f, p[9]);
    }
}

// numpy pairwise summation of float32 (np.sum over a 1-D float32 array)
static inline float np_sum_f32(const float* a, int n) {
    if (n < 8) { float r = 0.0f; for (int i = 0; i < n; i++) r += a[i]; return r; }
    if (n <= 128) {
        float r[8];
        for (int k = 0; k < 8; k++) r[k] = a[k];
        int i;
        for (i = 8; i < n - (n % 8); i += 8) for (int k = 0; k < 8; k++) r[k] += a[i + k];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2; n2 -= n2 % 8;
    return np_sum_f32(a, n2) + np_sum_f32(a + n2, n - n2);
}

