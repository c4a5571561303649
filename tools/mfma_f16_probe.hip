// Probe: how does v_mfma_f32_32x32x16_f16 round?  For random fp16 A (row 0), B (col 0) and fp32 C it prints the hardware
// result next to (a) a sequential fp32 fma chain over k, (b) the exactly rounded value of c + sum_k a_k b_k (RNE, computed
// with long double / __int128-free double-double since magnitudes are kept moderate), (c) fp32 adds of exact products in
// k order without fma.  Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_probe.hip -o /tmp/probe/p && /tmp/probe/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D, int n) {
    // trial t: A[t][16], B[t][16], C[t]; every lane row/col gets the same vectors so D[0][0] is the dot product
    const int lane = threadIdx.x;
    for (int t = 0; t < n; t++) {
        h8 a, b;
        for (int i = 0; i < 8; i++) { a[i] = A[t * 16 + 8 * (lane >> 5) + i]; b[i] = B[t * 16 + 8 * (lane >> 5) + i]; }
        f32x16 c;
        for (int i = 0; i < 16; i++) c[i] = C[t];
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        if (lane == 0) D[t] = c[0];
    }
}
int main(int argc, char** argv) {
    const int n = 200000;
    int mode = argc > 1 ? atoi(argv[1]) : 0;
    _Float16* hA = (_Float16*)malloc(n * 16 * 2); _Float16* hB = (_Float16*)malloc(n * 16 * 2);
    float* hC = (float*)malloc(n * 4); float* hD = (float*)malloc(n * 4);
    srand(1);
    auto rnd = [&]() { return (double)rand() / RAND_MAX * 2.0 - 1.0; };
    for (int t = 0; t < n; t++) {
        for (int i = 0; i < 16; i++) {
            double sa = mode == 0 ? 1.0 : std::ldexp(1.0, (rand() % 13) - 6);     // mode 1: wide exponent spread
            double sb = mode == 0 ? 1.0 : std::ldexp(1.0, (rand() % 13) - 6);
            hA[t * 16 + i] = (_Float16)(rnd() * sa); hB[t * 16 + i] = (_Float16)(rnd() * sb);
        }
        hC[t] = (float)(rnd() * (mode == 2 ? 1e-3 : 4.0));
    }
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, n * 32); hipMalloc(&dB, n * 32); hipMalloc(&dC, n * 4); hipMalloc(&dD, n * 4);
    hipMemcpy(dA, hA, n * 32, hipMemcpyHostToDevice); hipMemcpy(dB, hB, n * 32, hipMemcpyHostToDevice); hipMemcpy(dC, hC, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, n);
    hipMemcpy(hD, dD, n * 4, hipMemcpyDeviceToHost);
    long eq_seq = 0, eq_exact = 0, eq_add = 0, eq_rev = 0, eq_g4 = 0, eq_g8 = 0, eq_exact_then = 0;
    for (int t = 0; t < n; t++) {
        float seq = hC[t], addc = hC[t], rev = hC[t];
        long double ex = (long double)hC[t];
        long double dots = 0.0L;
        for (int i = 0; i < 16; i++) {
            float a = (float)hA[t * 16 + i], b = (float)hB[t * 16 + i];
            seq = fmaf(a, b, seq);
            addc = addc + a * b;                 // product exact in fp32
            ex += (long double)a * (long double)b;
            dots += (long double)a * (long double)b;
        }
        for (int i = 15; i >= 0; i--) rev = fmaf((float)hA[t * 16 + i], (float)hB[t * 16 + i], rev);
        // groups of 4 / 8: exact partial dot rounded-added to acc
        float g4 = hC[t], g8 = hC[t];
        for (int g = 0; g < 4; g++) { long double s = 0; for (int i = 0; i < 4; i++) s += (long double)(float)hA[t * 16 + 4 * g + i] * (float)hB[t * 16 + 4 * g + i]; g4 = (float)((long double)g4 + s); }
        for (int g = 0; g < 2; g++) { long double s = 0; for (int i = 0; i < 8; i++) s += (long double)(float)hA[t * 16 + 8 * g + i] * (float)hB[t * 16 + 8 * g + i]; g8 = (float)((long double)g8 + s); }
        float exact_then = hC[t] + (float)dots;   // dot rounded to fp32 first, then added
        eq_seq += seq == hD[t]; eq_exact += (float)ex == hD[t]; eq_add += addc == hD[t]; eq_rev += rev == hD[t];
        eq_g4 += g4 == hD[t]; eq_g8 += g8 == hD[t]; eq_exact_then += exact_then == hD[t];
        if (t < 4) printf("t%d hw %.9g seq %.9g exact %.9g g4 %.9g g8 %.9g\n", t, hD[t], seq, (float)ex, g4, g8);
    }
    printf("mode %d n %d: == sequential-fma %ld, == exact-one-rounding %ld, == fp32 adds %ld, == reverse fma %ld, == groups-of-4 %ld, == groups-of-8 %ld, == round(dot)+c %ld\n",
           mode, n, eq_seq, eq_exact, eq_add, eq_rev, eq_g4, eq_g8, eq_exact_then);
    return 0;
}
