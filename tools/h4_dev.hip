// tools/h4_dev.hip -- development harness of conv_h4.hip / conv_h3.hip (the f16x3 kernels): small shapes are checked against a
// float64 host convolution of the same operands (bound: 2^-19 * sum |a| |w| per output, the split's precision), large shapes are timed.
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/h4_dev.hip cald_amd/csrc/conv_h4.o cald_amd/csrc/conv_h3.o -o tools/h4_dev.bin
//   run:    tools/h4_dev.bin verify | tools/h4_dev.bin time V H W Cin Cout K stride pad epi(0/1/2) [iters] [kernel 4|3]
#include "../cald_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>

bool launch_conv_h4(const ConvArgs& a, hipStream_t stream);
bool launch_conv_h3(const ConvArgs& a, hipStream_t stream);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static unsigned rng = 12345u;
static float frand() { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) / 8388608.0f - 1.0f; }

static void split_host(float x, uint16_t* hi, uint16_t* lo) {
    const float s = x * 16.0f;
    const _Float16 h = (_Float16)s; const _Float16 l = (_Float16)(s - (float)h);
    memcpy(hi, &h, 2); memcpy(lo, &l, 2);
}
static float join_host(uint16_t hi, uint16_t lo) { _Float16 h, l; memcpy(&h, &hi, 2); memcpy(&l, &lo, 2); return ((float)h + (float)l) * 0.0625f; }
// fp32 [P][C] -> split form (h16.h)
static std::vector<unsigned char> to_split(const std::vector<float>& x, long long P, int C) {
    std::vector<unsigned char> o((size_t)P * C * 4);
    for (long long p = 0; p < P; p++)
        for (int c = 0; c < C; c++) {
            uint16_t hi, lo; split_host(x[(size_t)p * C + c], &hi, &lo);
            unsigned char* b = o.data() + ((size_t)p * C + (c & ~15)) * 4 + (c & 15) * 2;
            memcpy(b, &hi, 2); memcpy(b + 32, &lo, 2);
        }
    return o;
}
static float split_get(const unsigned char* t, long long p, int C, int c) {
    const unsigned char* b = t + ((size_t)p * C + (c & ~15)) * 4 + (c & 15) * 2;
    uint16_t hi, lo; memcpy(&hi, b, 2); memcpy(&lo, b + 32, 2);
    return join_host(hi, lo);
}
// same as api.hip pack_w16
static std::vector<uint16_t> pack_w16(const std::vector<float>& w, int Kpad, int CoutPad, float* unscale) {
    std::vector<uint16_t> o(w.size() * 2);
    float mx = 0.0f;
    for (float x : w) { const float ax = std::fabs(x); if (ax > mx) mx = ax; }
    int S = 0;
    if (mx > 0.0f) { int e; std::frexp(mx, &e); S = 14 - e; }
    *unscale = std::ldexp(1.0f, -(S + 4));
    for (int k = 0; k < Kpad; k++) {
        const int kt = k >> 4, kk = k & 15;
        for (int n = 0; n < CoutPad; n++) {
            const float x = std::ldexp(w[(size_t)k * CoutPad + n], S);
            const _Float16 hi = (_Float16)x; const _Float16 lo = (_Float16)(x - (float)hi);
            uint16_t hb, lb; memcpy(&hb, &hi, 2); memcpy(&lb, &lo, 2);
            o[(((size_t)kt * 2 + 0) * CoutPad + n) * 16 + kk] = hb;
            o[(((size_t)kt * 2 + 1) * CoutPad + n) * 16 + kk] = lb;
        }
    }
    return o;
}

struct Case { int V, H, W, Cin, Cout, K, stride, pad, epi, out_split, ex_split; };

template <typename T> static T* dev(const std::vector<T>& h) { T* d; CK(hipMalloc((void**)&d, h.size() * sizeof(T) + 64)); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

static int run_case(const Case& c, bool verify, int iters, int kernel) {
    const int KH = c.K, KW = c.K, Ho = (c.H + 2 * c.pad - KH) / c.stride + 1, Wo = (c.W + 2 * c.pad - KW) / c.stride + 1;
    const int CoutPad = cout_pad(c.Cout), K = KH * KW * c.Cin, Kpad = round_up(K, 16);
    const long long Pin = (long long)c.V * c.H * c.W, Pout = (long long)c.V * Ho * Wo;
    const int upH = (Ho + 1) / 2, upW = (Wo + 1) / 2;
    const long long Pup = (long long)c.V * upH * upW;
    std::vector<float> x((size_t)Pin * c.Cin), w((size_t)Kpad * CoutPad, 0.0f), wt((size_t)K * c.Cout), b(CoutPad, 0.0f), sc(CoutPad, 0.0f), sh(CoutPad, 0.0f);
    const bool zero = getenv("H4_ZERO") != nullptr;      // zero-filled operands: how much of the rate is the chip's power budget (DVFS)
    if (verify) for (auto& f : x) f = frand() * 3.0f; else for (size_t i = 0; i < x.size(); i++) x[i] = zero ? 0.0f : (float)((i * 2654435761u) >> 8 & 0xffff) / 32768.0f - 1.0f;
    for (int tap = 0; tap < KH * KW; tap++)
        for (int ci = 0; ci < c.Cin; ci++)
            for (int n = 0; n < c.Cout; n++) {
                const float f = (getenv("H4_ZERO") && !verify) ? 0.0f : frand() * 0.05f;
                wt[((size_t)tap * c.Cin + ci) * c.Cout + n] = f;
                w[(size_t)conv_k_index(tap, ci, KH * KW, c.Cin) * CoutPad + n] = f;
            }
    for (int n = 0; n < c.Cout; n++) { b[n] = frand() * 0.1f; sc[n] = 1.0f + frand() * 0.2f; sh[n] = frand() * 0.05f; }
    float unscale; std::vector<uint16_t> w16 = pack_w16(w, Kpad, CoutPad, &unscale);
    std::vector<unsigned char> xs = to_split(x, Pin, c.Cin);
    // extra operand: residual [Pout][Cout] or up [Pup][Cout]
    const long long Pex = c.epi == 1 ? Pout : (c.epi == 2 ? Pup : 0);
    std::vector<float> ex((size_t)Pex * c.Cout);
    for (auto& f : ex) f = frand();
    std::vector<unsigned char> exs = to_split(ex, Pex, c.Cout);
    BatchPlan* P = new BatchPlan; memset(P, 0, sizeof(BatchPlan));
    for (int v = 0; v <= c.V; v++) {
        P->seg[0][v].pix_off = (long long)v * c.H * c.W; P->seg[0][v].tile_start = v * ((c.H * c.W + 127) / 128); P->seg[0][v].H = c.H; P->seg[0][v].W = c.W;
        P->seg[1][v].pix_off = (long long)v * Ho * Wo; P->seg[1][v].tile_start = v * ((Ho * Wo + 127) / 128); P->seg[1][v].H = Ho; P->seg[1][v].W = Wo;
        P->seg[2][v].pix_off = (long long)v * upH * upW; P->seg[2][v].tile_start = v * ((upH * upW + 127) / 128); P->seg[2][v].H = upH; P->seg[2][v].W = upW;
    }
    BatchPlan* d_p; CK(hipMalloc((void**)&d_p, sizeof(BatchPlan))); CK(hipMemcpy(d_p, P, sizeof(BatchPlan), hipMemcpyHostToDevice));
    unsigned char* d_x = dev(xs); uint16_t* d_w16 = dev(w16); float *d_b = dev(b), *d_sc = dev(sc), *d_sh = dev(sh);
    unsigned char* d_exs = Pex ? dev(exs) : nullptr; float* d_ex = Pex ? dev(ex) : nullptr;
    float* d_out; unsigned* d_out16;
    CK(hipMalloc((void**)&d_out, (size_t)Pout * c.Cout * 4 + 64)); CK(hipMalloc((void**)&d_out16, (size_t)Pout * c.Cout * 4 + 64));
    CK(hipMemset(d_out, 0xff, (size_t)Pout * c.Cout * 4)); CK(hipMemset(d_out16, 0xff, (size_t)Pout * c.Cout * 4));
    ConvArgs a; memset(&a, 0, sizeof(a));
    a.in = nullptr; a.in16 = reinterpret_cast<const unsigned*>(d_x); a.out = c.out_split == 1 ? nullptr : d_out; a.out16 = c.out_split ? d_out16 : nullptr;
    a.w16 = d_w16; a.w16_unscale = unscale; a.bias = d_b; a.scale = d_sc; a.shift = d_sh;
    if (c.epi == 1) { a.residual = c.ex_split ? reinterpret_cast<const float*>(d_exs) : d_ex; a.ex16 = c.ex_split; }
    if (c.epi == 2) { a.up = c.ex_split ? reinterpret_cast<const float*>(d_exs) : d_ex; a.ex16 = c.ex_split; }
    a.seg_in = d_p->seg[0]; a.seg_out = d_p->seg[1]; a.seg_up = d_p->seg[2]; a.V = c.V; a.Cin = c.Cin; a.Cout = c.Cout; a.CoutPad = CoutPad; a.Kpad = Kpad;
    a.KH = KH; a.KW = KW; a.stride = c.stride; a.pad = c.pad; a.relu = 1; a.total_mtiles = c.V * ((Ho * Wo + 127) / 128); a.out_ld = c.Cout;
    auto launch = [&]() -> bool { return kernel == 3 ? launch_conv_h3(a, 0) : launch_conv_h4(a, 0); };
    if (!launch()) { printf("  kernel %d refused the shape\n", kernel); return 1; }
    CK(hipDeviceSynchronize());
    int bad = 0;
    if (verify) {
        std::vector<float> o((size_t)Pout * c.Cout); std::vector<unsigned char> o16((size_t)Pout * c.Cout * 4);
        CK(hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o16.data(), d_out16, o16.size(), hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (int v = 0; v < c.V; v++)
            for (int oy = 0; oy < Ho; oy++)
                for (int ox = 0; ox < Wo; ox++) {
                    const long long po = ((long long)v * Ho + oy) * Wo + ox;
                    for (int n = 0; n < c.Cout; n++) {
                        double s = 0.0, sa = 0.0;
                        for (int th = 0; th < KH; th++)
                            for (int tw = 0; tw < KW; tw++) {
                                const int iy = oy * c.stride - c.pad + th, ix = ox * c.stride - c.pad + tw;
                                if (iy < 0 || iy >= c.H || ix < 0 || ix >= c.W) continue;
                                const long long pi = ((long long)v * c.H + iy) * c.W + ix;
                                for (int ci = 0; ci < c.Cin; ci++) {
                                    const double av = split_get(xs.data(), pi, c.Cin, ci), wv = wt[((size_t)(th * KW + tw) * c.Cin + ci) * c.Cout + n];
                                    s += av * wv; sa += std::fabs(av * wv);
                                }
                            }
                        double val = s + b[n]; val = val * sc[n] + sh[n];
                        if (c.epi == 1) val += c.ex_split ? split_get(exs.data(), po, c.Cout, n) : ex[(size_t)po * c.Cout + n];
                        if (c.epi == 2) {
                            int sy = (int)std::floor((float)oy * ((float)upH / (float)Ho)); if (sy > upH - 1) sy = upH - 1;
                            int sx = (int)std::floor((float)ox * ((float)upW / (float)Wo)); if (sx > upW - 1) sx = upW - 1;
                            const long long pu = ((long long)v * upH + sy) * upW + sx;
                            val += c.ex_split ? split_get(exs.data(), pu, c.Cout, n) : ex[(size_t)pu * c.Cout + n];
                        }
                        if (val < 0.0) val = 0.0;
                        const double tol = std::ldexp(sa + std::fabs(val) + 1.0, -19);
                        if (c.out_split != 1) {
                            const double d = std::fabs((double)o[(size_t)po * c.Cout + n] - val);
                            if (!(d <= tol)) { if (bad < 5) printf("  fp32 out mismatch v%d (%d,%d) n%d: got %g want %g\n", v, oy, ox, n, o[(size_t)po * c.Cout + n], val); bad++; }
                            if (d / tol > worst) worst = d / tol;
                        }
                        if (c.out_split) {
                            const double g = split_get(o16.data(), po, c.Cout, n), d = std::fabs(g - val);
                            if (!(d <= 2 * tol)) { if (bad < 5) printf("  split out mismatch v%d (%d,%d) n%d: got %g want %g\n", v, oy, ox, n, g, val); bad++; }
                        }
                    }
                }
        printf("  %s  (worst |err| / bound = %.3f)\n", bad ? "FAIL" : "ok", worst);
    } else {
        for (int i = 0; i < 2; i++) launch();
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        const double fl = 2.0 * (double)Pout * c.Cout * K;
        printf("  kernel h%d: %.3f ms  %.1f TFLOP/s-equivalent (%.1f issued fp16)\n", kernel, ms, fl / ms * 1e-9, 3 * fl / ms * 1e-9);
    }
    hipFree(d_x); hipFree(d_w16); hipFree(d_b); hipFree(d_sc); hipFree(d_sh); hipFree(d_out); hipFree(d_out16); hipFree(d_p);
    if (d_exs) hipFree(d_exs); if (d_ex) hipFree(d_ex);
    delete P;
    return bad ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && !strcmp(argv[1], "verify")) {
        const int kernel = argc > 2 ? atoi(argv[2]) : 4;
        const Case cases[] = {
            // V  H   W  Cin Cout K  s  p epi out_split ex_split
            {2, 21, 27, 32, 256, 3, 1, 1, 0, 2, 0},      // 3 x 3, borders, ragged M tiles, both outputs
            {3, 16, 20, 16, 256, 1, 1, 0, 0, 0, 0},      // KT = 1
            {1, 30, 40, 48, 256, 1, 1, 0, 1, 1, 1},      // KT = 3, split residual, split-only output
            {2, 33, 29, 64, 512, 3, 2, 1, 0, 2, 0},      // stride 2, two N tiles
            {2, 24, 36, 80, 256, 1, 1, 0, 2, 2, 1},      // FPN top-down from a split tensor (KT = 5)
            {2, 24, 36, 64, 256, 1, 1, 0, 2, 0, 0},      // FPN top-down from an fp32 tensor
            {1, 19, 25, 128, 256, 3, 1, 1, 1, 2, 0},     // residual from fp32
            {5, 12, 11, 32, 256, 3, 1, 1, 0, 1, 0},      // five views, odd number of M tiles
        };
        int fails = 0;
        for (const Case& c : cases) {
            printf("case V=%d %dx%d Cin=%d Cout=%d k=%d s=%d p=%d epi=%d out_split=%d ex_split=%d\n", c.V, c.H, c.W, c.Cin, c.Cout, c.K, c.stride, c.pad, c.epi, c.out_split, c.ex_split);
            fails += run_case(c, true, 1, kernel);
        }
        printf(fails ? "VERIFY FAILED (%d)\n" : "VERIFY OK\n", fails);
        return fails ? 1 : 0;
    }
    if (argc >= 11 && !strcmp(argv[1], "time")) {
        Case c{atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]), atoi(argv[9]), atoi(argv[10]), 1, 1};
        const int iters = argc > 11 ? atoi(argv[11]) : 5, kernel = argc > 12 ? atoi(argv[12]) : 4;
        printf("time V=%d %dx%d Cin=%d Cout=%d k=%d s=%d p=%d epi=%d\n", c.V, c.H, c.W, c.Cin, c.Cout, c.K, c.stride, c.pad, c.epi);
        return run_case(c, false, iters, kernel);
    }
    fprintf(stderr, "usage: %s verify [kernel] | time V H W Cin Cout K stride pad epi [iters] [kernel]\n", argv[0]);
    return 2;
}
