// Probe: can exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) and VALU fp32 FMA run concurrently on a SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV>   // NV = VALU pk_fma per MFMA
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    f32x2 v[16];
    for (int i = 0; i < 16; i++) { v[i][0] = (float)threadIdx.x; v[i][1] = 1.0f; }
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    f32x2 pa = {a, a * 0.5f}, pb = {b, b * 0.25f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; q++) {
                const int idx = (m * NV + q) & 15;
                v[idx] = __builtin_elementwise_fma(pa, pb, v[idx]);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    for (int i = 0; i < 16; i++) s += v[i][0] + v[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV> void run(const char* name, int bpc = 4, int iters = 4000) {
    float* d; hipMalloc(&d, 1024 * 256 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * bpc;   // bpc blocks x 4 waves per CU -> bpc waves per SIMD
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<NV>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 1.0001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = blocks * 4.0;
    double mfma_fl = waves * iters * 4.0 * 4096.0, valu_fl = waves * iters * 4.0 * NV * 256.0;
    printf("%s bpc=%d iters=%d NV=%d: %.3f ms  MFMA %.1f TF  VALU %.1f TF  total %.1f TF\n", name, bpc, iters, NV, ms, mfma_fl / ms / 1e9, valu_fl / ms / 1e9, (mfma_fl + valu_fl) / ms / 1e9);
    hipFree(d);
}
int main() {
    for (int bpc = 1; bpc <= 4; bpc++) run<0>("mfma only", bpc, 4000);
    run<0>("mfma only", 1, 40000); run<0>("mfma only", 2, 40000); run<0>("mfma only", 4, 40000);
    run<2>("hybrid", 1, 40000); run<4>("hybrid", 1, 40000); run<8>("hybrid", 1, 40000); run<16>("hybrid", 1, 40000);
    run<4>("hybrid", 2, 40000); run<8>("hybrid", 2, 40000);
    return 0;
}
