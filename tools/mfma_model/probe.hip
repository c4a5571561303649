// probe.hip -- runs v_mfma_f32_32x32x16_f16 on a file of (a[16] fp16, b[16] fp16, c fp32) dot-product cases and writes the fp32 results.
// Test infrastructure for oracle/mfma_f16_model.h (the CPU restatement of the instruction that CALD_PRECISION_F16X3 is built on).
//   build:  hipcc --offload-arch=gfx950 -O2 tools/mfma_model/probe.hip -o tools/mfma_model/probe.bin
//   run:    probe.bin cases.bin out.bin      cases.bin = int64 n | n x 16 u16 (A) | n x 16 u16 (B) | n x u32 (C);  out.bin = n x u32 (D)
// One wave evaluates 32 cases per instruction: case i of a block of 32 is row i of A and column i of B, D[i][i] is its result.
// A second instruction per block puts case i's b vector in column (i + 5) & 31 instead: the result must not depend on where in the
// tile a dot product sits (the count of differences is printed; it has always been 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void probe_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, const uint32_t* __restrict__ C,
                                                    uint32_t* __restrict__ D, long long n32) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long blk = (long long)blockIdx.x * 4 + wave;
    const bool live = blk < n32;
    const long long base = (live ? blk : 0) * 32;
    const int l31 = lane & 31, kh = lane >> 5;
    const h8 a = __builtin_bit_cast(h8, *reinterpret_cast<const u32x4*>(A + (base + l31) * 16 + 8 * kh));
    const h8 b = __builtin_bit_cast(h8, *reinterpret_cast<const u32x4*>(B + (base + l31) * 16 + 8 * kh));
    const h8 b5 = __builtin_bit_cast(h8, *reinterpret_cast<const u32x4*>(B + (base + ((l31 - 5) & 31)) * 16 + 8 * kh));
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; r++) c[r] = __builtin_bit_cast(float, C[base + 4 * kh + (r & 3) + 8 * (r >> 2)]);
    const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    const f32x16 d5 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b5, c, 0, 0, 0);
    // every lane spills its 16 results to the wave's LDS tile [row][col]; lanes 0..31 then read the diagonal (and the shifted one)
    __shared__ float tile[4][2][32 * 33];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = 4 * kh + (r & 3) + 8 * (r >> 2);
        tile[wave][0][row * 33 + l31] = d[r];
        tile[wave][1][row * 33 + l31] = d5[r];
    }
    __syncthreads();
    if (live && lane < 32) {
        D[base + lane] = __builtin_bit_cast(uint32_t, tile[wave][0][lane * 33 + lane]);
        D[n32 * 32 + base + lane] = __builtin_bit_cast(uint32_t, tile[wave][1][lane * 33 + ((lane + 5) & 31)]);
    }
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s cases.bin out.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("cases"); return 2; }
    int64_t n = 0;
    if (fread(&n, 8, 1, f) != 1 || n <= 0) return 2;
    const int64_t n32 = (n + 31) / 32, np = n32 * 32;
    std::vector<uint16_t> hA(np * 16, 0), hB(np * 16, 0);
    std::vector<uint32_t> hC(np, 0), hD(2 * np, 0);
    if (fread(hA.data(), 2, n * 16, f) != (size_t)n * 16 || fread(hB.data(), 2, n * 16, f) != (size_t)n * 16 || fread(hC.data(), 4, n, f) != (size_t)n) {
        fprintf(stderr, "short read\n"); return 2;
    }
    fclose(f);
    uint16_t *dA, *dB; uint32_t *dC, *dD;
    hipMalloc(&dA, np * 32); hipMalloc(&dB, np * 32); hipMalloc(&dC, np * 4); hipMalloc(&dD, 2 * np * 4);
    hipMemcpy(dA, hA.data(), np * 32, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), np * 32, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC.data(), np * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_kernel, dim3((unsigned)((n32 + 3) / 4)), dim3(256), 0, 0, dA, dB, dC, dD, (long long)n32);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    hipMemcpy(hD.data(), dD, 2 * np * 4, hipMemcpyDeviceToHost);
    long long moved = 0;
    for (int64_t i = 0; i < n; i++) moved += hD[i] != hD[np + i];
    printf("%lld cases; results that depend on the tile position: %lld\n", (long long)n, moved);
    f = fopen(argv[2], "wb");
    fwrite(hD.data(), 4, n, f);
    fclose(f);
    return moved ? 3 : 0;
}
