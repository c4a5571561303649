// conv_i3.hip -- CALD_PRECISION_I8X3: implicit-GEMM conv / linear on the int8 matrix pipe with EXACT integer accumulation.
//
// The arithmetic (restated bit for bit by the CPU oracle, orc_conv2d_i8x3 under oracle/):
//   * the layer's input tensor is quantised to 24-bit fixed point with ONE static power-of-two exponent e_x per layer
//     (calibrated once per model, cald_model_calibrate): q_x = clamp(rint(x * 2^(22 - e_x)), +-0x7F7F7F);
//     weights per output channel n: q_w = clamp(rint(w * 2^(22 - e_w[n])), +-0x7F7F7F), fixed at model finalize;
//   * both are written as three balanced signed base-256 digits (d0 + 256 d1 + 65536 d2, each in [-128, 127]); the six digit
//     products of weight >= 2^16 -- d2.d2 | d2.d1 + d1.d2 | d2.d0 + d0.d2 + d1.d1 -- are accumulated in int32 by
//     v_mfma_i32_32x32x32_i8 (integer sums are exact and order-free, so no summation order is part of the contract); the
//     three dropped products are < 2^-23 of |x|max |w|max per term;
//   * T = 2^32 S2 + 2^24 S1 + 2^16 S0 is formed exactly in double, scaled by the exact power of two 2^(e_x + e_w[n] - 44) and
//     rounded ONCE to float32; then the exact mode's fp32 epilogue: (+bias) -> (*bn_scale, +bn_shift) -> (+residual |
//     +upsampled) -> ReLU.
// That makes the mode reproducible on a CPU (tests compare tobytes()-equal), which the split-fp16 mode (conv_h3.hip) is not.
//
// Data path: the quantised digits of the input come as three int8 planes [pixel][Cin] written by quantize_planes_kernel
// (one streaming pass over the fp32 tensor), so the GEMM loop is copy + MFMA only: 16-byte buffer loads (hardware zero
// fill for out-of-image taps) -> ds_write_b128 -> ds_read_b128 -> MFMA.  128 x 64 x 32 tiles, 4 waves (64 x 32 each, 96
// accumulator registers), two LDS buffers, one barrier per k-tile; k-tiles walk (32-channel chunk, kh, kw) so the taps of a
// chunk re-touch the same bytes in L1 / L2; XCD-contiguous tile map for filters with a spatial extent.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define I3_QMAX 0x7F7F7F     /* 127 * (1 + 256 + 65536): the largest magnitude three balanced digits can hold */

// ---------------------------------------------------------------------------------------------
// fp32 tensor -> three digit planes.  n4 = elements / 4; plane p at dst + p * plane_stride.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void i3_digits(float x, float scale, int& d0, int& d1, int& d2) {
    float t = x * scale;                               // exact: scale is a power of two (no overflow: |x| 2^(22-e) stays finite)
    t = t > (float)I3_QMAX ? (float)I3_QMAX : (t < -(float)I3_QMAX ? -(float)I3_QMAX : t);
    const int q = (int)rintf(t);                       // round to nearest even
    d0 = (int)(signed char)(q & 255);
    const int q1 = (q - d0) >> 8;
    d1 = (int)(signed char)(q1 & 255);
    d2 = (q1 - d1) >> 8;
}
__global__ __launch_bounds__(256) void quantize_planes_kernel(const float4* __restrict__ src, long long n4, float scale,
                                                              unsigned* __restrict__ dst, long long plane_stride4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = src[i];
        int a0, a1, a2, b0, b1, b2, c0, c1, c2, e0, e1, e2;
        i3_digits(v.x, scale, a0, a1, a2); i3_digits(v.y, scale, b0, b1, b2);
        i3_digits(v.z, scale, c0, c1, c2); i3_digits(v.w, scale, e0, e1, e2);
        dst[i] = (unsigned)(a0 & 255) | ((unsigned)(b0 & 255) << 8) | ((unsigned)(c0 & 255) << 16) | ((unsigned)(e0 & 255) << 24);
        dst[i + plane_stride4] = (unsigned)(a1 & 255) | ((unsigned)(b1 & 255) << 8) | ((unsigned)(c1 & 255) << 16) | ((unsigned)(e1 & 255) << 24);
        dst[i + 2 * plane_stride4] = (unsigned)(a2 & 255) | ((unsigned)(b2 & 255) << 8) | ((unsigned)(c2 & 255) << 16) | ((unsigned)(e2 & 255) << 24);
    }
}
void launch_quantize_planes(const float* src, long long n, int exp, signed char* dst, long long plane_stride, hipStream_t st) {
    const long long n4 = n / 4;
    if (n4 <= 0) return;
    long long blocks = (n4 + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(quantize_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(src), n4,
                       ldexpf(1.0f, 22 - exp), reinterpret_cast<unsigned*>(dst), plane_stride / 4);
}

// max |x| of a tensor (calibration): bit pattern of a non-negative float orders like the float
__global__ __launch_bounds__(256) void absmax_kernel(const float4* __restrict__ src, long long n4, unsigned* out) {
    float m = 0.0f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = src[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(out, __builtin_bit_cast(unsigned, m));
}
void launch_absmax(const float* src, long long n, unsigned* out, hipStream_t st) {
    const long long n4 = n / 4;
    if (n4 <= 0) return;
    long long blocks = (n4 + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(src), n4, out);
}

// ---------------------------------------------------------------------------------------------
// the GEMM
// ---------------------------------------------------------------------------------------------
// WN = 2: 128 x 64 tiles, 256 threads (layers with 64 output channels, and the tail-quantised small launches);
// WN = 4: 128 x 128 tiles, 512 threads = 8 waves of 64 x 32 -- 1.5 x the digit-MACs per byte moved from L2 into LDS, which is
//         what bounds this kernel (the 128 x 64 tile sustains ~9 TB/s of L2 -> LDS traffic at 265 TF-eq).
template <int EPI, int WN>
__device__ __forceinline__ void conv_i3_body(const ConvArgs& a, const int blk) {
    constexpr int BM = 128, BN = 32 * WN, BK = 32, TM = 2;
    constexpr int PLANE_A = BM * 32, PLANE_B = BN * 32, TILE_B = 3 * PLANE_A + 3 * PLANE_B;       // bytes: 12288 + 6144
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_B];

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

    // ---- A gather: row = tid >> 1, 16-byte half = tid & 1, the three planes ----
    const bool is_a = WN == 2 || tid < 256;          // WN = 4: waves 0-3 stage A, waves 4-7 stage B (three pieces each)
    const int arow = (tid & 255) >> 1, ahalf = tid & 1;
    unsigned rowmask = 0;
    int rowvoff;
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
        rowvoff = ((oy * a.stride) * Wi + ox * a.stride) * Cin + 16 * ahalf;
    }
    const signed char* pl0 = a.i8_in + si.pix_off * (long long)Cin - (long long)a.pad * (Wi + 1) * Cin;
    const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc((void*)pl0, 0, 0x7FFE0000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA1 = __builtin_amdgcn_make_buffer_rsrc((void*)(pl0 + a.i8_plane_stride), 0, 0x7FFE0000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(pl0 + 2 * a.i8_plane_stride), 0, 0x7FFE0000, 0x00020000);
    // B: packed [kt][plane][CoutPad][32 B]; piece t: plane t >> 7, column (t & 127) >> 1, half t & 1; threads < 128 also plane 2
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(a.w8) + (long long)n0 * 32), 0, 0x7FFE0000, 0x00020000);
    // WN = 2: piece t: plane t >> 7, column (t & 127) >> 1; threads < 128 also plane 2.  WN = 4: thread 256 + t: column t >> 1, all planes
    const int b_p = WN == 2 ? (tid >> 7) : 0, b_nl = WN == 2 ? ((tid & 127) >> 1) : ((tid & 255) >> 1), b_h = tid & 1;
    const bool b_two = WN == 2 && tid < 128;
    const int bvoff0 = b_p * CoutPad * 32 + b_nl * 32 + b_h * 16, bvoff1 = 2 * CoutPad * 32 + b_nl * 32 + b_h * 16;
    const int aw_off = arow * 32 + ((ahalf ^ ((arow >> 3) & 1)) * 16);
    const int bw_off0 = 3 * PLANE_A + b_p * PLANE_B + b_nl * 32 + ((b_h ^ ((b_nl >> 3) & 1)) * 16);
    const int bw_off1 = 3 * PLANE_A + 2 * PLANE_B + b_nl * 32 + ((b_h ^ ((b_nl >> 3) & 1)) * 16);
    int u_kh = 0, u_kw = 0, u_ci = 0, u_kt = 0;
    i32x4 ra0, ra1, ra2, rb0, rb1 = {0, 0, 0, 0};     // WN = 4, B waves: ra0..ra2 carry the three B planes

#define I3_LOAD()                                                                                          \
    {                                                                                                      \
        const int soffB = u_kt * 3 * CoutPad * 32;                                                         \
        const unsigned u_bit = 1u << (u_kh * KW + u_kw);                                                   \
        const int soffA = (u_kh * Wi + u_kw) * Cin + u_ci;                                                 \
        const int v0 = (rowmask & u_bit) ? rowvoff : 0x7FFF0000;                                           \
        if (is_a) {                                                                                        \
            ra0 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA0, v0, soffA, 0));    \
            ra1 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA1, v0, soffA, 0));    \
            ra2 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA2, v0, soffA, 0));    \
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
        u_kw++; if (u_kw == KW) { u_kw = 0; u_kh++; if (u_kh == KH) { u_kh = 0; u_ci += BK; } }            \
    }
#define I3_STORE(BUF)                                                                                      \
    {                                                                                                      \
        unsigned char* tb = smem + (BUF) * TILE_B;                                                         \
        if (is_a) {                                                                                        \
            *reinterpret_cast<i32x4*>(tb + aw_off) = ra0;                                                  \
            *reinterpret_cast<i32x4*>(tb + PLANE_A + aw_off) = ra1;                                        \
            *reinterpret_cast<i32x4*>(tb + 2 * PLANE_A + aw_off) = ra2;                                    \
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

    i32x16 acc0[TM], acc1[TM], acc2[TM];        // digit-product sums of weight 2^16, 2^24, 2^32
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc0[i][r] = 0; acc1[i][r] = 0; acc2[i][r] = 0; }

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
    int cur = 0;
    for (int kt = 0; kt < KT; kt++) {
        const unsigned char* tc = smem + cur * TILE_B;
        const bool has1 = kt + 1 < KT, has2 = kt + 2 < KT;
        i32x4 a0[TM], a1[TM], a2[TM], b0, b1, b2;
#pragma unroll
        for (int t = 0; t < TM; t++) {
            a2[t] = *reinterpret_cast<const i32x4*>(tc + 2 * PLANE_A + fo_a[t]);
            a1[t] = *reinterpret_cast<const i32x4*>(tc + PLANE_A + fo_a[t]);
            a0[t] = *reinterpret_cast<const i32x4*>(tc + fo_a[t]);
        }
        b2 = *reinterpret_cast<const i32x4*>(tc + 2 * PLANE_B + fo_b);
        b1 = *reinterpret_cast<const i32x4*>(tc + PLANE_B + fo_b);
        b0 = *reinterpret_cast<const i32x4*>(tc + fo_b);
        I3_MFMA(acc2, a2, b2)
        I3_MFMA(acc1, a2, b1)
        if (has1) I3_STORE(cur ^ 1)
        I3_MFMA(acc1, a1, b2)
        I3_MFMA(acc0, a2, b0)
        if (has2) I3_LOAD()
        I3_MFMA(acc0, a0, b2)
        I3_MFMA(acc0, a1, b1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
#undef I3_LOAD
#undef I3_STORE
#undef I3_MFMA

    // ---- epilogue: exact combine in double, one rounding to float32, then the exact mode's fp32 epilogue ----
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
    const double unscale = (double)a.i8_in_unscale * (double)a.w8_unscale[nc];      // 2^(e_x - 22 + 16) * 2^(e_w[n] - 22)
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
        for (int r = 0; r < 16; r++) {
            // T / 2^16 = 65536 S2 + 256 S1 + S0: |T / 2^16| < 2^47, exact in double; the factor 2^16 lives in `unscale`
            const double T = fma((double)acc2[i][r], 65536.0, fma((double)acc1[i][r], 256.0, (double)acc0[i][r]));
            val[r] = (float)(T * unscale);
        }
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
    return a.w8 && a.i8_in && a.CoutPad % 64 == 0 && a.Cin % 32 == 0 && a.KH * a.KW <= 32 && !a.in_relu;
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
