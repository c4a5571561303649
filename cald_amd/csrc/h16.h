// h16.h -- the split-plane activation format of CALD_PRECISION_F16X3 (conv_h3.hip / conv_h4.hip), device side.
//
// A tensor [pixel][C] (C % 16 == 0) kept "in split form" occupies the same 4 * C bytes per pixel as the fp32 tensor would,
// laid out per 16-channel chunk as   [16 x fp16 hi | 16 x fp16 lo]   (64 bytes):   16 * x = hi + lo (+ 2^-22 relative).
// Element (pixel, c): hi at byte (pixel * C + (c & ~15)) * 4 + (c & 15) * 2, lo 32 bytes further.  The layout is the MFMA
// operand layout of v_mfma_f32_32x32x16_f16 for one k-step (16 consecutive k of one row = one chunk), so a consumer moves
// the 64 bytes of a (pixel, chunk) to LDS as four 16-byte pieces -- with buffer_load ... lds, no register, no VALU -- and a
// fragment is one ds_read_b128.  (Rounds 2-3 stored one word {hi, lo} per element: the loader then needed four byte
// permutes and two 8-byte LDS stores per float4, and could not use the LDS-DMA path at all.)
//
// Producers (conv epilogues, maxpool, RoIAlign) write it with DWORD stores: lanes l and l ^ 1 hold neighbouring channels,
// the even lane stores {hi(c), hi(c + 1)}, the odd lane {lo(c - 1), lo(c)} after one DPP exchange.  Epilogue reads of a
// split tensor (residual, FPN top-down) use the mirror image: one dword load per lane + one exchange.
#pragma once
#include "common.h"

// hi | lo << 16 of 16 * x
__device__ __forceinline__ unsigned h16_split(const float x) { return split16_word(x); }
__device__ __forceinline__ float h16_join(const unsigned w) {
    const _Float16 hi = __builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
    const _Float16 lo = __builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
    return ((float)hi + (float)lo) * 0.0625f;
}
// the other lane of the (even, odd) pair
__device__ __forceinline__ unsigned h16_partner(const unsigned w) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}
// Lane holding channel c (its pair partner holds c ^ 1): the dword this lane stores and its byte offset inside the pixel row.
//   even c: {hi(c), hi(c + 1)} at chunk + 2 (c & 15);   odd c: {lo(c - 1), lo(c)} at chunk + 32 + 2 ((c - 1) & 15)
__device__ __forceinline__ unsigned h16_pair_word(const unsigned own, const unsigned partner, const bool odd) {
    return odd ? __builtin_amdgcn_perm(partner, own, 0x03020706u) : __builtin_amdgcn_perm(partner, own, 0x05040100u);
}
__device__ __forceinline__ int h16_pair_off(const int c) {   // bytes from the start of the pixel's row of C channels
    return (c & ~15) * 4 + ((c & 1) ? 32 + ((c - 1) & 15) * 2 : (c & 15) * 2);
}
// mirror image: `own` = the dword loaded at h16_pair_off(c), `partner` = the pair partner's; returns hi | lo << 16 of channel c
__device__ __forceinline__ unsigned h16_unpair_word(const unsigned own, const unsigned partner, const bool odd) {
    // even: hi = own.lo16, lo = partner.lo16;   odd: hi = partner.hi16, lo = own.hi16
    return odd ? __builtin_amdgcn_perm(partner, own, 0x03020706u) : __builtin_amdgcn_perm(partner, own, 0x05040100u);
}
// scalar (non-paired) access, for kernels whose lanes do not sit on channel pairs
__device__ __forceinline__ void h16_store1(unsigned char* row, const int c, const float x) {
    const unsigned w = h16_split(x);
    unsigned short* p = reinterpret_cast<unsigned short*>(row + (c & ~15) * 4 + (c & 15) * 2);
    p[0] = (unsigned short)(w & 0xffffu); p[16] = (unsigned short)(w >> 16);
}
__device__ __forceinline__ float h16_load1(const unsigned char* row, const int c) {
    const unsigned short* p = reinterpret_cast<const unsigned short*>(row + (c & ~15) * 4 + (c & 15) * 2);
    return h16_join((unsigned)p[0] | ((unsigned)p[16] << 16));
}
// four consecutive channels c .. c + 3 (c % 4 == 0) of one pixel: two 8-byte stores
__device__ __forceinline__ void h16_store4(unsigned char* row, const int c, const float4 v) {
    const unsigned w0 = h16_split(v.x), w1 = h16_split(v.y), w2 = h16_split(v.z), w3 = h16_split(v.w);
    uint2 hi, lo;
    hi.x = __builtin_amdgcn_perm(w1, w0, 0x05040100u); hi.y = __builtin_amdgcn_perm(w3, w2, 0x05040100u);
    lo.x = __builtin_amdgcn_perm(w1, w0, 0x07060302u); lo.y = __builtin_amdgcn_perm(w3, w2, 0x07060302u);
    unsigned char* p = row + (c & ~15) * 4 + (c & 15) * 2;
    *reinterpret_cast<uint2*>(p) = hi;
    *reinterpret_cast<uint2*>(p + 32) = lo;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Fused fp32 epilogue shared by conv_h3.hip / conv_h4.hip: un-scale, +bias, FrozenBN, +residual | +nearest-upsampled top-down,
// ReLU -- the exact mode's operation order -- then fp32 and / or split-form stores.  The wave owns TM x TN accumulator tiles of
// 32 x 32 whose first row / column are m_w0 (inside the view) / n_w0; lane l holds column l & 31, rows 4 (l >> 5) + (r & 3) + 8 (r >> 2).
//   EPI 0: nothing extra; 1: residual (same geometry as out); 2: FPN top-down (nearest up-sampling of the coarser level).
// residual / up are fp32 tensors, or split-form ones when a.ex16 is set (then a.residual / a.up point at the split tensor).
// ---------------------------------------------------------------------------------------------------------------------------
typedef float h16_f32x16 __attribute__((ext_vector_type(16)));
template <int EPI, int TM, int TN>
__device__ __forceinline__ void h16_epilogue(const ConvArgs& a, const h16_f32x16 (&acc)[TM][TN], const LevelSeg so, const int v,
                                             const int m_w0, const int n_w0, const int Mv, const int lane) {
    // Addresses stay off the VALU (conv_p4.hip's scheme): one byte offset per lane and accumulator tile, the 16 rows of a tile through
    // the SCALAR offset of the buffer instruction; the descriptors end at the view's last valid row, so rows past it (and the lanes of
    // padded output channels, whose offset is out of range) load zeros and store nothing -- no per-element tests or branches.  All 16
    // loads of a tile are issued before the first use.
    const int l31 = lane & 31, kh_lane = lane >> 5;
    const int out_ld = a.out_ld, Wo = so.W, Ho = so.H, row_b = out_ld * 4;
    const unsigned valid_b = (unsigned)Mv * (unsigned)row_b;
    float* const out_v = a.out ? a.out + so.pix_off * (long long)out_ld : nullptr;
    unsigned* const out16_v = a.out16 ? a.out16 + so.pix_off * (long long)out_ld : nullptr;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(out_v ? (void*)out_v : (void*)out16_v), 0, valid_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(out16_v ? (void*)out16_v : (void*)out_v), 0, valid_b, 0x00020000);
    const bool ex16 = a.ex16 != 0;
    const float* ex_v = nullptr;
    int upH = 1, upW = 1;
    float uph_s = 0.f, upw_s = 0.f;
    unsigned ex_b = valid_b;
    if (EPI == 1) ex_v = a.residual + so.pix_off * (long long)out_ld;
    if (EPI == 2) {
        const LevelSeg su = a.seg_up[v];
        ex_v = a.up + su.pix_off * (long long)out_ld;
        upH = su.H; upW = su.W;
        uph_s = (float)upH / (float)Ho; upw_s = (float)upW / (float)Wo;
        ex_b = (unsigned)(upH * upW) * (unsigned)row_b;
    }
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI != 0 ? (void*)ex_v : (void*)out_v), 0, ex_b, 0x00020000);
    const bool relu = a.relu != 0, has_bias = a.bias != nullptr, has_bn = a.scale != nullptr;
    const int Mlast = Mv - 1;
    const float unscale = a.w16_unscale;
    const bool odd = (lane & 1) != 0;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n_w0 + j * 32 + l31;
        const bool nok = n < a.Cout;
        const int nc = nok ? n : 0;
        const float bs = has_bias ? a.bias[nc] : 0.0f;
        const float sc = has_bn ? a.scale[nc] : 1.0f;
        const float sh = has_bn ? a.shift[nc] : 0.0f;
        const int poff = h16_pair_off(nc);          // split form: byte offset of this lane's dword inside a pixel's row of channels
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int mrow = m_w0 + i * 32;                                   // wave-uniform; this lane's rows: mrow + 4 kh_lane + (r & 3) + 8 (r >> 2)
            const int vrow = 4 * kh_lane * row_b;
            const int vo32 = nok ? vrow + n * 4 : 0x7FFF0000, vo16 = nok ? vrow + poff : 0x7FFF0000;
            float extra[16];
            if (EPI == 1) {
                // (issuing tile t + 1's loads before tile t is processed was measured: no gain on the HBM-bound expand layers, round 4)
                unsigned raw[16];
#pragma unroll
                for (int r = 0; r < 16; r++)
                    raw[r] = __builtin_amdgcn_raw_buffer_load_b32(rsX, ex16 ? vo16 : vo32, (mrow + (r & 3) + 8 * (r >> 2)) * row_b, 0);
#pragma unroll
                for (int r = 0; r < 16; r++)
                    extra[r] = ex16 ? h16_join(h16_unpair_word(raw[r], h16_partner(raw[r]), odd)) : __builtin_bit_cast(float, raw[r]);
            } else if (EPI == 2) {
                unsigned raw[16];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    int m = mrow + 4 * kh_lane + (r & 3) + 8 * (r >> 2);
                    m = m < Mlast ? m : Mlast;
                    const int oy = m / Wo, ox = m - oy * Wo;
                    int sy = (int)floorf((float)oy * uph_s); sy = sy > upH - 1 ? upH - 1 : sy;
                    int sx = (int)floorf((float)ox * upw_s); sx = sx > upW - 1 ? upW - 1 : sx;
                    raw[r] = __builtin_amdgcn_raw_buffer_load_b32(rsX, (sy * upW + sx) * row_b + (ex16 ? poff : nc * 4), 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; r++)
                    extra[r] = ex16 ? h16_join(h16_unpair_word(raw[r], h16_partner(raw[r]), odd)) : __builtin_bit_cast(float, raw[r]);
            }
            float val[16];
#pragma unroll
            for (int r = 0; r < 16; r++) val[r] = acc[i][j][r] * unscale;
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
            if (out_v) {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val[r]), rsO, vo32, (mrow + (r & 3) + 8 * (r >> 2)) * row_b, 0);
            }
            if (out16_v) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const unsigned w = h16_split(val[r]);
                    __builtin_amdgcn_raw_buffer_store_b32(h16_pair_word(w, h16_partner(w), odd), rsS, vo16, (mrow + (r & 3) + 8 * (r >> 2)) * row_b, 0);
                }
            }
        }
    }
}
