// conv_i3.hip -- CALD_PRECISION_I8X3: implicit-GEMM conv / linear on the int8 matrix pipe with EXACT integer accumulation and
// block-floating-point operands (one exponent per pixel / per output channel).
//
// The arithmetic (restated bit for bit by the CPU oracle, orc_conv2d_i8x3 under oracle/):
//   * activations: every PIXEL p of the layer's input (its Cin channels) gets the exponent e_p of its largest magnitude
//     (max |x| = f * 2^e_p, f in [0.5, 1); all-zero pixel: 0) and is quantised to 23-bit fixed point,
//     q = rint(x * 2^(22 - e_p));  weights: per output channel n, q = rint(w * 2^(22 - e_w[n])), fixed at model finalize;
//   * both are written as three balanced signed base-256 digits (d0 + 256 d1 + 65536 d2, each in [-128, 127]);
//   * per filter TAP the six digit products of weight >= 2^16 -- d2.d2 | d2.d1 + d1.d2 | d2.d0 + d0.d2 + d1.d1 -- are summed
//     over the tap's Cin channels in int32 by v_mfma_i32_32x32x32_i8: integer sums are exact and order-free, so no summation
//     order is part of the contract (the three dropped products are < 2^-23 of the pixel's / channel's maxima per term);
//   * the tap's sums are folded into one float32 accumulator per output, taps in (kh, kw) order:
//         T = fmaf((float)S2, 65536, fmaf((float)S1, 256, (float)S0));   acc = fmaf(T, 2^(e_p - 22), acc)      (p = the tap's pixel)
//     and the result is acc * 2^(e_w[n] - 22 + 16) (exact power of two), followed by the exact mode's fp32 epilogue
//     (+bias) -> (*bn_scale, +bn_shift) -> (+residual | +upsampled) -> ReLU.
// Every step is either exact integer arithmetic or a float32 operation in a fixed order, so a CPU reproduces it bit for bit
// (tests compare tobytes()-equal) -- which the split-fp16 mode (conv_h3.hip) cannot offer -- and the per-pixel exponent keeps
// ~19-20 significant bits on typical activations (a per-LAYER exponent, tried first, kept ~14: 14 % of the images moved by
// more than 1e-4).  No calibration, no saturation: the exponents come from the data.
//
// Data path: quantize_pixels_kernel writes the three int8 digit planes [pixel][Cin] and the per-pixel scale 2^(e_p - 22) in one
// streaming pass over the fp32 tensor; the GEMM loop is copy + MFMA: 16-byte buffer loads (hardware zero fill for
// out-of-image taps) -> ds_write_b128 -> ds_read_b128 -> MFMA.  k-tiles walk (kh, kw, 32-channel chunk); after the last chunk
// of a tap the int32 sums are folded with the tap's per-row scales (staged in LDS) and the next tap starts from zero.
// Tiles: 128 x 128 x 32 with 512 threads (8 waves of 64 x 32) for K >= 512, 128 x 64 / 256 threads otherwise; two LDS buffers,
// one barrier per k-tile; XCD-contiguous tile map for filters with a spatial extent.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// fp32 tensor [P][C] -> three digit planes [P][C] (plane p at dst + p * plane_stride) + per-pixel scale 2^(e_p - 22).
// G lanes per pixel (G = 16 / 32 / 64 by C), 256 / G pixels per workgroup pass.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned i3_pack_digit(int a, int b, int c, int d) {
    return (unsigned)(a & 255) | ((unsigned)(b & 255) << 8) | ((unsigned)(c & 255) << 16) | ((unsigned)(d & 255) << 24);
}
__device__ __forceinline__ void i3_digits(float x, float scale, int& d0, int& d1, int& d2) {
    const int q = (int)rintf(x * scale);               // x * scale is exact (power of two); |q| <= 2^22; round to nearest even
    d0 = (int)(signed char)(q & 255);
    const int q1 = (q - d0) >> 8;
    d1 = (int)(signed char)(q1 & 255);
    d2 = (q1 - d1) >> 8;
}
template <int G>
__global__ __launch_bounds__(256) void quantize_pixels_kernel(const float4* __restrict__ src, long long P, int C4, unsigned* __restrict__ dst,
                                                              long long plane_stride4, float* __restrict__ rowscale) {
    const int sub = threadIdx.x % G, grp = threadIdx.x / G;
    constexpr int PPB = 256 / G;
    for (long long p = (long long)blockIdx.x * PPB + grp; p < P; p += (long long)gridDim.x * PPB) {
        const float4* row = src + p * C4;
        float m = 0.0f;
        for (int i = sub; i < C4; i += G) {
            const float4 v = row[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        // m = f * 2^e, f in [0.5, 1)  ->  e = biased exponent - 126 (zero: 0; tiny / huge maxima: clamped, digits saturate to 0 / garbage-free)
        const unsigned bits = __builtin_bit_cast(unsigned, m);
        int e = (int)((bits >> 23) & 255u) - 126;
        if (m == 0.0f) e = 0;
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        const float qs = __builtin_bit_cast(float, (unsigned)(22 - e + 127) << 23);      // 2^(22 - e)
        for (int i = sub; i < C4; i += G) {
            const float4 v = row[i];
            int a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
            i3_digits(v.x, qs, a0, a1, a2); i3_digits(v.y, qs, b0, b1, b2);
            i3_digits(v.z, qs, c0, c1, c2); i3_digits(v.w, qs, d0, d1, d2);
            const long long o = p * C4 + i;
            dst[o] = i3_pack_digit(a0, b0, c0, d0);
            dst[o + plane_stride4] = i3_pack_digit(a1, b1, c1, d1);
            dst[o + 2 * plane_stride4] = i3_pack_digit(a2, b2, c2, d2);
        }
        if (sub == 0) rowscale[p] = __builtin_bit_cast(float, (unsigned)(e - 22 + 127) << 23);   // 2^(e - 22)
    }
}
void launch_quantize_pixels(const float* src, long long P, int C, signed char* dst, long long plane_stride, float* rowscale, hipStream_t st) {
    if (P <= 0 || C < 4) return;
    const int C4 = C / 4;
    const int G = C4 >= 64 ? 64 : (C4 >= 32 ? 32 : 16);
    long long blocks = (P + (256 / G) - 1) / (256 / G); if (blocks > 65536) blocks = 65536;
    const float4* s4 = reinterpret_cast<const float4*>(src); unsigned* d4 = reinterpret_cast<unsigned*>(dst);
    if (G == 64) hipLaunchKernelGGL((quantize_pixels_kernel<64>), dim3((unsigned)blocks), dim3(256), 0, st, s4, P, C4, d4, plane_stride / 4, rowscale);
    else if (G == 32) hipLaunchKernelGGL((quantize_pixels_kernel<32>), dim3((unsigned)blocks), dim3(256), 0, st, s4, P, C4, d4, plane_stride / 4, rowscale);
    else hipLaunchKernelGGL((quantize_pixels_kernel<16>), dim3((unsigned)blocks), dim3(256), 0, st, s4, P, C4, d4, plane_stride / 4, rowscale);
}

// ---------------------------------------------------------------------------------------------
// the GEMM
// ---------------------------------------------------------------------------------------------
// WN = 2: 128 x 64 tiles, 256 threads (layers with 64 output channels, short chains);
// WN = 4: 128 x 128 tiles, 512 threads = 8 waves of 64 x 32 -- 1.5 x the digit-MACs per byte moved from L2 into LDS.
template <int EPI, int WN>
__device__ __forceinline__ void conv_i3_body(const ConvArgs& a, const int blk) {
    constexpr int BM = 128, BN = 32 * WN, BK = 32, TM = 2;
    constexpr int PLANE_A = BM * 32, PLANE_B = BN * 32, TILE_B = 3 * PLANE_A + 3 * PLANE_B;       // bytes
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_B];
    __shared__ __attribute__((aligned(16))) float s_scale[2][BM];        // per-row scale 2^(e_p - 22) of the tap being accumulated

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int NT = a.CoutPad / BN;
    int mt, nt;
    if (a.KH * a.KW > 1) {
        const int MT = a.total_mtiles, CH = (MT + 7) >> 3;
        const int xcd = blk & 7, idx = blk >> 3;
        mt = xcd * CH + idx / NT; nt = idx % NT;
        if (idx / NT >= CH || mt >= MT) return;
    } else {
        const int b = blk, MT = a.total_mtiles, MT8 = MT & ~7;
        if (b < MT8 * NT) { const int xcd = b & 7, idx = b >> 3; mt = (idx / NT) * 8 + xcd; nt = idx % NT; }
        else { const int r = b - MT8 * NT; mt = MT8 + r / NT; nt = r % NT; }
    }
    const int n0 = nt * BN;
    int v = 0;
    while (v + 1 < a.V && a.seg_out[v + 1].tile_start <= mt) v++;
    const LevelSeg so = a.seg_out[v];
    const LevelSeg si = a.seg_in[v];
    const int Ho = so.H, Wo = so.W, Hi = si.H, Wi = si.W;
    int Mv = Ho * Wo;
    if (a.dyn_rows) { const int d = a.dyn_rows[v]; Mv = d < Mv ? d : Mv; }
    const int m0 = (mt - so.tile_start) * BM;
    if (m0 >= Mv) return;
    const int Cin = a.Cin, KW = a.KW, KH = a.KH, CoutPad = a.CoutPad;
    const int CC = Cin / BK;                            // 32-channel chunks per tap

    // ---- A gather: row = tid >> 1, 16-byte half = tid & 1, the three planes; half 0 also fetches the row's scale per tap ----
    const bool is_a = WN == 2 || tid < 256;          // WN = 4: waves 0-3 stage A, waves 4-7 stage B (three pieces each)
    const int arow = (tid & 255) >> 1, ahalf = tid & 1;
    unsigned rowmask = 0;
    int rowvoff, rowpix;
    {
        const int m = m0 + arow;
        const int oy = m / Wo, ox = m - oy * Wo;
        const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
        if (m < Mv)
            for (int t = 0; t < KH * KW; t++) {
                const int th = t / KW, tw = t - th * KW;
                const int iy = iy0 + th, ix = ix0 + tw;
                if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) rowmask |= 1u << t;
            }
        rowpix = (oy * a.stride) * Wi + ox * a.stride;
        rowvoff = rowpix * Cin + 16 * ahalf;
    }
    const signed char* pl0 = a.i8_in + si.pix_off * (long long)Cin - (long long)a.pad * (Wi + 1) * Cin;
    const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc((void*)pl0, 0, 0x7FFE0000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA1 = __builtin_amdgcn_make_buffer_rsrc((void*)(pl0 + a.i8_plane_stride), 0, 0x7FFE0000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(pl0 + 2 * a.i8_plane_stride), 0, 0x7FFE0000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.i8_rowscale + si.pix_off - (long long)a.pad * (Wi + 1)), 0, 0x7FFE0000, 0x00020000);
    // B: packed [kt][plane][CoutPad][32 B], kt = tap * CC + chunk
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(a.w8) + (long long)n0 * 32), 0, 0x7FFE0000, 0x00020000);
    // WN = 2: piece t: plane t >> 7, column (t & 127) >> 1; threads < 128 also plane 2.  WN = 4: thread 256 + t: column t >> 1, all planes
    const int b_p = WN == 2 ? (tid >> 7) : 0, b_nl = WN == 2 ? ((tid & 127) >> 1) : ((tid & 255) >> 1), b_h = tid & 1;
    const bool b_two = WN == 2 && tid < 128;
    const int bvoff0 = b_p * CoutPad * 32 + b_nl * 32 + b_h * 16, bvoff1 = 2 * CoutPad * 32 + b_nl * 32 + b_h * 16;
    const int aw_off = arow * 32 + ((ahalf ^ ((arow >> 3) & 1)) * 16);
    const int bw_off0 = 3 * PLANE_A + b_p * PLANE_B + b_nl * 32 + ((b_h ^ ((b_nl >> 3) & 1)) * 16);
    const int bw_off1 = 3 * PLANE_A + 2 * PLANE_B + b_nl * 32 + ((b_h ^ ((b_nl >> 3) & 1)) * 16);
    int u_tap = 0, u_kh = 0, u_kw = 0, u_cc = 0, u_kt = 0;     // loader cursor: tile u_kt = (tap u_tap, chunk u_cc)
    i32x4 ra0, ra1, ra2, rb0, rb1 = {0, 0, 0, 0};     // WN = 4, B waves: ra0..ra2 carry the three B planes
    float rs_new = 0.0f; bool rs_have = false; int rs_tap = 0;

#define I3_LOAD()                                                                                          \
    {                                                                                                      \
        const int soffB = u_kt * 3 * CoutPad * 32;                                                         \
        const bool ok = (rowmask >> u_tap) & 1u;                                                           \
        const int soffA = (u_kh * Wi + u_kw) * Cin + u_cc * BK;                                            \
        const int v0 = ok ? rowvoff : 0x7FFF0000;                                                          \
        if (is_a) {                                                                                        \
            ra0 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA0, v0, soffA, 0));    \
            ra1 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA1, v0, soffA, 0));    \
            ra2 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA2, v0, soffA, 0));    \
            rs_have = u_cc == 0;                                                                           \
            if (rs_have) {      /* first chunk of a tap: the row's scale at that tap's pixel (out of image: 0) */ \
                rs_tap = u_tap;                                                                            \
                rs_new = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsS, ok ? rowpix * 4 : 0x7FFF0000, (u_kh * Wi + u_kw) * 4, 0)); \
            }                                                                                              \
        }                                                                                                  \
        if (WN == 2) {                                                                                     \
            rb0 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff0, soffB, 0)); \
            if (b_two) rb1 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff1, soffB, 0)); \
        } else if (!is_a) {                                                                                \
            ra0 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff0, soffB, 0)); \
            ra1 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff0 + CoutPad * 32, soffB, 0)); \
            ra2 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff0 + 2 * CoutPad * 32, soffB, 0)); \
        }                                                                                                  \
        u_kt++;                                                                                            \
        u_cc++; if (u_cc == CC) { u_cc = 0; u_tap++; u_kw++; if (u_kw == KW) { u_kw = 0; u_kh++; } }       \
    }
#define I3_STORE(BUF)                                                                                      \
    {                                                                                                      \
        unsigned char* tb = smem + (BUF) * TILE_B;                                                         \
        if (is_a) {                                                                                        \
            *reinterpret_cast<i32x4*>(tb + aw_off) = ra0;                                                  \
            *reinterpret_cast<i32x4*>(tb + PLANE_A + aw_off) = ra1;                                        \
            *reinterpret_cast<i32x4*>(tb + 2 * PLANE_A + aw_off) = ra2;                                    \
            if (rs_have && ahalf == 0) s_scale[rs_tap & 1][arow] = rs_new;                                 \
        }                                                                                                  \
        if (WN == 2) {                                                                                     \
            *reinterpret_cast<i32x4*>(tb + bw_off0) = rb0;                                                 \
            if (b_two) *reinterpret_cast<i32x4*>(tb + bw_off1) = rb1;                                      \
        } else if (!is_a) {                                                                                \
            *reinterpret_cast<i32x4*>(tb + bw_off0) = ra0;                                                 \
            *reinterpret_cast<i32x4*>(tb + bw_off0 + PLANE_B) = ra1;                                       \
            *reinterpret_cast<i32x4*>(tb + bw_off0 + 2 * PLANE_B) = ra2;                                   \
        }                                                                                                  \
    }

    i32x16 acc0[TM], acc1[TM], acc2[TM];        // digit-product sums of the current tap, weights 2^16, 2^24, 2^32
    float accf[TM][16];                         // the output's float32 accumulator over the taps
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) accf[i][r] = 0.0f;
    const i32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    const int KT = a.Kpad / BK;
    I3_LOAD();
    I3_STORE(0);
    if (KT > 1) I3_LOAD();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int kh_lane = lane >> 5, l31 = lane & 31;
    int fo_a[TM], fo_b;
#pragma unroll
    for (int t = 0; t < TM; t++) { const int m = wm * 64 + t * 32 + l31; fo_a[t] = m * 32 + ((kh_lane ^ ((m >> 3) & 1)) * 16); }
    { const int n = wn * 32 + l31; fo_b = 3 * PLANE_A + n * 32 + ((kh_lane ^ ((n >> 3) & 1)) * 16); }

#define I3_MFMA(ACC, FA, FB)                                                                               \
    _Pragma("unroll") for (int i = 0; i < TM; i++) ACC[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(FA[i], FB, ACC[i], 0, 0, 0);
#define I3_MFMA0(ACC, FA, FB)      /* first product of a tap into this accumulator set: starts from zero */  \
    _Pragma("unroll") for (int i = 0; i < TM; i++) ACC[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(FA[i], FB, zero16, 0, 0, 0);
    int cur = 0, c_cc = 0, c_tap = 0, kt = 0;   // the tile being computed: (tap c_tap, chunk c_cc)
    // Fragments of tile kt + 1 are read right after the barrier of tile kt, under its last MFMA group and the tap fold, instead of at
    // the top of the next iteration (where every wave of the workgroup waited out the LDS latency together); two fragment sets, the
    // loop unrolled by two so that no register is copied.
#define I3_READ(A0, A1, A2, B0, B1, B2, TB)                                                                \
    {                                                                                                      \
        _Pragma("unroll") for (int t = 0; t < TM; t++) {                                                   \
            A2[t] = *reinterpret_cast<const i32x4*>((TB) + 2 * PLANE_A + fo_a[t]);                         \
            A1[t] = *reinterpret_cast<const i32x4*>((TB) + PLANE_A + fo_a[t]);                             \
            A0[t] = *reinterpret_cast<const i32x4*>((TB) + fo_a[t]);                                       \
        }                                                                                                  \
        B2 = *reinterpret_cast<const i32x4*>((TB) + 2 * PLANE_B + fo_b);                                   \
        B1 = *reinterpret_cast<const i32x4*>((TB) + PLANE_B + fo_b);                                       \
        B0 = *reinterpret_cast<const i32x4*>((TB) + fo_b);                                                 \
    }
#define I3_TILE(A0, A1, A2, B0, B1, B2, NA0, NA1, NA2, NB0, NB1, NB2)                                      \
    {                                                                                                      \
        const bool has1 = kt + 1 < KT, has2 = kt + 2 < KT;                                                 \
        if (c_cc == 0) {                                                                                   \
            I3_MFMA0(acc2, A2, B2)                                                                         \
            I3_MFMA0(acc1, A2, B1)                                                                         \
            if (has1) I3_STORE(cur ^ 1)                                                                    \
            I3_MFMA(acc1, A1, B2)                                                                          \
            I3_MFMA0(acc0, A2, B0)                                                                         \
        } else {                                                                                           \
            I3_MFMA(acc2, A2, B2)                                                                          \
            I3_MFMA(acc1, A2, B1)                                                                          \
            if (has1) I3_STORE(cur ^ 1)                                                                    \
            I3_MFMA(acc1, A1, B2)                                                                          \
            I3_MFMA(acc0, A2, B0)                                                                          \
        }                                                                                                  \
        if (has2) I3_LOAD()                                                                                \
        I3_MFMA(acc0, A0, B2)                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                      \
        cur ^= 1;                                                                                          \
        if (has1) I3_READ(NA0, NA1, NA2, NB0, NB1, NB2, smem + cur * TILE_B)                               \
        I3_MFMA(acc0, A1, B1)                                                                              \
        if (++c_cc == CC) {                                                                                \
            /* the tap is complete: fold its exact integer sums into the float accumulators with the per-row scales */ \
            const float* sp = s_scale[c_tap & 1];                                                          \
            _Pragma("unroll") for (int i = 0; i < TM; i++) {                                               \
                const int rb = wm * 64 + i * 32 + 4 * kh_lane;                                             \
                _Pragma("unroll") for (int q = 0; q < 4; q++) {                                            \
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sp + rb + 8 * q);                     \
                    _Pragma("unroll") for (int j = 0; j < 4; j++) {                                        \
                        const int r = 4 * q + j;                                                           \
                        const float T = fmaf((float)acc2[i][r], 65536.0f, fmaf((float)acc1[i][r], 256.0f, (float)acc0[i][r])); \
                        accf[i][r] = fmaf(T, s4[j], accf[i][r]);                                           \
                    }                                                                                      \
                }                                                                                          \
            }                                                                                              \
            c_cc = 0; c_tap++;                                                                             \
        }                                                                                                  \
        kt++;                                                                                              \
    }
    i32x4 fa0[TM], fa1[TM], fa2[TM], fb0, fb1, fb2, ga0[TM], ga1[TM], ga2[TM], gb0, gb1, gb2;
    I3_READ(fa0, fa1, fa2, fb0, fb1, fb2, smem)
    while (kt + 1 < KT) {
        I3_TILE(fa0, fa1, fa2, fb0, fb1, fb2, ga0, ga1, ga2, gb0, gb1, gb2)
        I3_TILE(ga0, ga1, ga2, gb0, gb1, gb2, fa0, fa1, fa2, fb0, fb1, fb2)
    }
    if (kt < KT) I3_TILE(fa0, fa1, fa2, fb0, fb1, fb2, ga0, ga1, ga2, gb0, gb1, gb2)
#undef I3_READ
#undef I3_TILE
#undef I3_LOAD
#undef I3_STORE
#undef I3_MFMA
#undef I3_MFMA0

    // ---- epilogue: * 2^(e_w[n] - 22 + 16) (exact), then the exact mode's fp32 epilogue ----
    const int out_ld = a.out_ld;
    float* __restrict__ out_v = a.out + so.pix_off * (long long)out_ld;
    const float* __restrict__ ex_v = nullptr;
    int upH = 1, upW = 1;
    float uph_s = 0.f, upw_s = 0.f;
    if (EPI == 1) ex_v = a.residual + so.pix_off * (long long)out_ld;
    if (EPI == 2) {
        const LevelSeg su = a.seg_up[v];
        ex_v = a.up + su.pix_off * (long long)out_ld;
        upH = su.H; upW = su.W;
        uph_s = (float)upH / (float)Ho; upw_s = (float)upW / (float)Wo;
    }
    const bool relu = a.relu != 0, has_bias = a.bias != nullptr, has_bn = a.scale != nullptr;
    const int Mlast = Mv - 1;
    const int n = n0 + wn * 32 + l31;
    const bool nok = n < a.Cout;
    const int nc = nok ? n : 0;
    const float bs = has_bias ? a.bias[nc] : 0.0f;
    const float sc = has_bn ? a.scale[nc] : 1.0f;
    const float sh = has_bn ? a.shift[nc] : 0.0f;
    const float unscale = a.w8_unscale[nc];                                          // 2^(e_w[n] - 22 + 16)
    // full tiles: one byte offset per lane, the 16 rows of an accumulator tile through the scalar offset of the buffer instruction
    const bool full_tile = m0 + BM <= Mv;
    const int row_b = out_ld * 4;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)out_v, 0, 0x7FFE0000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 1 ? ex_v : out_v), 0, 0x7FFE0000, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mbase = m0 + wm * 64 + i * 32 + 4 * kh_lane;
        const int vo = nok ? (mbase * out_ld + n) * 4 : 0x7FFF0000;
        float extra[16];
        if (EPI == 1 && full_tile) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                extra[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, vo, ((r & 3) + 8 * (r >> 2)) * row_b, 0));
        } else if (EPI != 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int m = mbase + (r & 3) + 8 * (r >> 2);
                m = m < Mlast ? m : Mlast;
                if (EPI == 1) {
                    extra[r] = ex_v[(long long)m * out_ld + nc];
                } else {
                    const int oy = m / Wo, ox = m - oy * Wo;
                    int sy = (int)floorf((float)oy * uph_s); sy = sy > upH - 1 ? upH - 1 : sy;
                    int sx = (int)floorf((float)ox * upw_s); sx = sx > upW - 1 ? upW - 1 : sx;
                    extra[r] = ex_v[(long long)(sy * upW + sx) * out_ld + nc];
                }
            }
        }
        float val[16];
#pragma unroll
        for (int r = 0; r < 16; r++) val[r] = accf[i][r] * unscale;
        if (has_bias) {
#pragma unroll
            for (int r = 0; r < 16; r++) val[r] = val[r] + bs;
        }
        if (has_bn) {
#pragma unroll
            for (int r = 0; r < 16; r++) { val[r] = val[r] * sc; val[r] = val[r] + sh; }
        }
        if (EPI != 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) val[r] = val[r] + extra[r];
        }
        if (relu) {
#pragma unroll
            for (int r = 0; r < 16; r++) val[r] = val[r] > 0.0f ? val[r] : 0.0f;
        }
        if (full_tile) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val[r]), rsO, vo, ((r & 3) + 8 * (r >> 2)) * row_b, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m < Mv && nok) out_v[(long long)m * out_ld + n] = val[r];
            }
        }
    }
}

template <int EPI, int WN>
__global__ __launch_bounds__(128 * WN, 2) void conv_i3_kernel(const ConvArgs a) { conv_i3_body<EPI, WN>(a, blockIdx.x); }
template <int WN>
__global__ __launch_bounds__(128 * WN, 2) void conv_i3_group_kernel(const ConvGroup g) {
    int i = 0;
    while (i + 1 < g.n && g.blk0[i + 1] <= (int)blockIdx.x) i++;
    conv_i3_body<0, WN>(g.p[i], (int)blockIdx.x - g.blk0[i]);
}

static inline bool i3_covers(const ConvArgs& a) {
    return a.w8 && a.i8_in && a.i8_rowscale && a.CoutPad % 64 == 0 && a.Cin % 32 == 0 && a.KH * a.KW <= 32 && !a.in_relu;
}
static inline int i3_grid_mtiles(const ConvArgs& a) { return a.KH * a.KW > 1 ? 8 * ((a.total_mtiles + 7) / 8) : a.total_mtiles; }
// 128 x 128 tiles when the channel count allows it and the launch still fills the chip (one 512-thread workgroup per CU)
static inline bool i3_wide(const ConvArgs& a) {
    static const int wide_env = getenv("CALD_I3_WIDE") ? atoi(getenv("CALD_I3_WIDE")) : 1;
    // short chains (K < 512) are bound by HBM / per-workgroup overheads, where the smaller workgroup does better (measured)
    return wide_env && a.CoutPad % 128 == 0 && a.Kpad >= 512 && (long long)a.total_mtiles * (a.CoutPad / 128) >= 512;
}

bool launch_conv_i3_group(const ConvArgs* p, int n, hipStream_t stream) {
    if (n < 1 || n > CALD_MAX_GROUP) return false;
    bool wide = true;
    for (int i = 0; i < n; i++) { if (!i3_covers(p[i]) || p[i].residual || p[i].up) return false; wide = wide && p[i].CoutPad % 128 == 0; }
    static const int wide_env = getenv("CALD_I3_WIDE") ? atoi(getenv("CALD_I3_WIDE")) : 1;
    wide = wide && wide_env;
    ConvGroup g; g.n = n; int blk = 0;
    for (int i = 0; i < n; i++) { g.blk0[i] = blk; blk += i3_grid_mtiles(p[i]) * (p[i].CoutPad / (wide ? 128 : 64)); g.p[i] = p[i]; }
    g.blk0[n] = blk;
    if (wide) hipLaunchKernelGGL((conv_i3_group_kernel<4>), dim3((unsigned)blk), dim3(512), 0, stream, g);
    else hipLaunchKernelGGL((conv_i3_group_kernel<2>), dim3((unsigned)blk), dim3(256), 0, stream, g);
    return true;
}
bool launch_conv_i3(const ConvArgs& a, hipStream_t stream) {
    if (!i3_covers(a)) return false;
    if (i3_wide(a)) {
        const dim3 grid((unsigned)(i3_grid_mtiles(a) * (a.CoutPad / 128))), block(512);
        if (a.residual) hipLaunchKernelGGL((conv_i3_kernel<1, 4>), grid, block, 0, stream, a);
        else if (a.up) hipLaunchKernelGGL((conv_i3_kernel<2, 4>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((conv_i3_kernel<0, 4>), grid, block, 0, stream, a);
        return true;
    }
    const dim3 grid((unsigned)(i3_grid_mtiles(a) * (a.CoutPad / 64))), block(256);
    if (a.residual) hipLaunchKernelGGL((conv_i3_kernel<1, 2>), grid, block, 0, stream, a);
    else if (a.up) hipLaunchKernelGGL((conv_i3_kernel<2, 2>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((conv_i3_kernel<0, 2>), grid, block, 0, stream, a);
    return true;
}
