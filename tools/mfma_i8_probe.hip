// Probe: v_mfma_i32_32x32x32_i8 operand convention + issue rate (tools only, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_i8_probe.hip -o tools/mfma_i8_probe.bin && tools/mfma_i8_probe.bin
// Convention checked: lane l holds row (A) / column (B) l % 32 and the 16 k-bytes 16 * (l / 32) + 0..15; C/D: col = l & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).  Integer accumulation is exact, so only the A/B k pairing matters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const signed char* A, const signed char* B, int* D) {   // A[32][32] row-major (m, k); B[32][32] (k, n)
    const int l = threadIdx.x, half = l >> 5, rc = l & 31;
    i32x4 a, b;
    signed char ab[16], bb[16];
    for (int i = 0; i < 16; i++) { ab[i] = A[rc * 32 + 16 * half + i]; bb[i] = B[(16 * half + i) * 32 + rc]; }
    __builtin_memcpy(&a, ab, 16); __builtin_memcpy(&b, bb, 16);
    i32x16 c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + rc] = c[r];
}
__global__ void rate(int* out, int iters) {
    i32x4 a = {(int)threadIdx.x, 3, 5, 7}, b = {11, (int)threadIdx.x, 1, 2};
    i32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    std::vector<signed char> A(1024), B(1024); std::vector<int> D(1024), R(1024, 0);
    srand(1); for (auto& x : A) x = (signed char)(rand() % 256 - 128); for (auto& x : B) x = (signed char)(rand() % 256 - 128);
    for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) { int s = 0; for (int k = 0; k < 32; k++) s += (int)A[m * 32 + k] * (int)B[k * 32 + n]; R[m * 32 + n] = s; }
    signed char *dA, *dB; int* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; i++) bad += D[i] != R[i];
    printf("i8 32x32x32 layout check: %d mismatches of 1024\n", bad);
    int* dO; hipMalloc(&dO, 1024 * 256 * 4 * 4);
    for (int wpb : {256, 512, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000, blocks = 1024;
        hipLaunchKernelGGL(rate, dim3(blocks), dim3(wpb), 0, 0, dO, 100);
        hipEventRecord(e0); hipLaunchKernelGGL(rate, dim3(blocks), dim3(wpb), 0, 0, dO, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double ops = (double)blocks * (wpb / 64) * iters * 4.0 * 2.0 * 32 * 32 * 32;
        printf("threads/block %4d: %.1f TOPS\n", wpb, ops / (ms * 1e-3) / 1e12);
    }
    return bad != 0;
}
