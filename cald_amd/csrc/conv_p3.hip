// conv_p3.hip -- software-pipelined variant of the implicit-GEMM fp32 MFMA convolution (128x128x16 tile).
//
// Same arithmetic as conv_mfma.hip (one k-ordered fma chain per output, identical epilogue) -- results are
// bit-identical.  What changes is the schedule inside a workgroup:
//   * THREE LDS buffers and ONE raw s_barrier per k-tile, placed inside the MFMA burst (k-step 5 of 8):
//       burst t:  k-step 1: ds_write tile t+1 (registers loaded during burst t-1) -> buffer (t+1)%3
//                 k-step 2: global loads of tile t+2 -> registers (in flight for a whole burst)
//                 k-step 5: s_waitcnt lgkmcnt(0); s_barrier         (tile t+1 is now visible)
//                 k-step 7: prefetch the first fragments of tile t+1
//     so a wave's MFMA stream is continuous across k-tiles: no LDS-read or global latency sits between two
//     bursts, and the barrier only costs the skew between the four waves.
//   * hazards: buffer (t+1)%3 last held tile t-2, whose readers finished before barrier t-1, which the
//     writer has passed; tile t+1 is read only after barrier t, which every writer reaches after its writes.
//   The fp32 MFMA pipe loses ~20 % when four waves per SIMD interleave (tools/mfma_valu_probe.hip); this
//   schedule is meant to run at <= 3 waves per SIMD (49.5 KB LDS per workgroup).
#include "common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int EPI>
__global__ __launch_bounds__(256, 3) void conv_p3_kernel(const ConvArgs a) {
    constexpr int BM = 128, BN = 128, BK = 16, SA = 130, SB = 128, TM = 2, TN = 2, WN = 2;
    constexpr int TILE_F = BK * SA + BK * SB;
    __shared__ __attribute__((aligned(16))) float smem[3 * TILE_F];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int NT = a.CoutPad / BN;
    int mt, nt;
    {
        const int b = blockIdx.x, MT = a.total_mtiles, MT8 = MT & ~7;
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
    const float* __restrict__ in_v = a.in + si.pix_off * (long long)a.Cin;
    const float* __restrict__ wgt = a.w;
    const int Cin = a.Cin, KW = a.KW, KH = a.KH, CoutPad = a.CoutPad;
    const bool in_relu = a.in_relu != 0;

    // A gather through a buffer resource (SRSRC + 32-bit voffset + wave-uniform soffset): the per-tile address
    // work is 3 VALU per row (tap-valid test -> voffset or an out-of-range offset that the hardware returns as
    // zeros) instead of 64-bit pointer arithmetic; the B (weights) loads need no VALU at all.
    // The resource base is shifted by -pad rows/cols so that voffset and soffset are both non-negative.
    const int g = tid & 3, arow = tid >> 2;
    unsigned rowmask[2];
    int rowvoff[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int m = m0 + arow + 64 * p;
        const int oy = m / Wo, ox = m - oy * Wo;
        const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
        unsigned msk = 0;
        if (m < Mv)
            for (int t = 0; t < KH * KW; t++) {
                const int th = t / KW, tw = t - th * KW;
                const int iy = iy0 + th, ix = ix0 + tw;
                if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) msk |= 1u << t;
            }
        rowmask[p] = msk;
        rowvoff[p] = (((oy * a.stride) * Wi + ox * a.stride) * Cin + 4 * g) * 4;       // bytes, >= 0
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(in_v - (long long)a.pad * (Wi + 1) * Cin), 0, 0x7FFE0000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(wgt + n0), 0, 0x7FFE0000, 0x00020000);
    int u_kh = 0, u_kw = 0, u_ci = 0, u_kt = 0;      // wave-uniform cursor of the NEXT tile to load
    const int bk0 = tid >> 5, bc0 = tid & 31;        // B: rows bk0 and bk0+8, float4 column bc0
    const int bvoff0 = (bk0 * CoutPad + 4 * bc0) * 4, bvoff1 = bvoff0 + 8 * CoutPad * 4;

    float ax0, ay0, az0, aw0, ax1, ay1, az1, aw1, bx0, by0, bz0, bw0, bx1, by1, bz1, bw1;

#define P3_LOAD()                                                                                          \
    {                                                                                                      \
        const unsigned u_bit = 1u << (u_kh * KW + u_kw);                                                   \
        const int soffA = ((u_kh * Wi + u_kw) * Cin + u_ci) * 4;                                           \
        const int soffB = u_kt * BK * CoutPad * 4;                                                         \
        const int v0 = (rowmask[0] & u_bit) ? rowvoff[0] : 0x7FFF0000;                                     \
        const int v1 = (rowmask[1] & u_bit) ? rowvoff[1] : 0x7FFF0000;                                     \
        const f32x4 t0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, v0, soffA, 0)); \
        const f32x4 t1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, v1, soffA, 0)); \
        const f32x4 t2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff0, soffB, 0)); \
        const f32x4 t3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff1, soffB, 0)); \
        ax0 = t0[0]; ay0 = t0[1]; az0 = t0[2]; aw0 = t0[3]; ax1 = t1[0]; ay1 = t1[1]; az1 = t1[2]; aw1 = t1[3]; \
        bx0 = t2[0]; by0 = t2[1]; bz0 = t2[2]; bw0 = t2[3]; bx1 = t3[0]; by1 = t3[1]; bz1 = t3[2]; bw1 = t3[3]; \
        u_kt++; u_ci += BK;                                                                                \
        if (u_ci >= Cin) { u_ci = 0; u_kw++; if (u_kw == KW) { u_kw = 0; u_kh++; } }                       \
    }
#define P3_STORE(BUF)                                                                                      \
    {                                                                                                      \
        if (in_relu) {                                                                                     \
            ax0 = ax0 < 0.f ? 0.f : ax0; ay0 = ay0 < 0.f ? 0.f : ay0; az0 = az0 < 0.f ? 0.f : az0; aw0 = aw0 < 0.f ? 0.f : aw0; \
            ax1 = ax1 < 0.f ? 0.f : ax1; ay1 = ay1 < 0.f ? 0.f : ay1; az1 = az1 < 0.f ? 0.f : az1; aw1 = aw1 < 0.f ? 0.f : aw1; \
        }                                                                                                  \
        float* dA = smem + (BUF) * TILE_F + (4 * g) * SA + arow;                                           \
        dA[0] = ax0; dA[SA] = ay0; dA[2 * SA] = az0; dA[3 * SA] = aw0;                                     \
        dA[64] = ax1; dA[SA + 64] = ay1; dA[2 * SA + 64] = az1; dA[3 * SA + 64] = aw1;                     \
        float* dB = smem + (BUF) * TILE_F + BK * SA;                                                       \
        *reinterpret_cast<float4*>(dB + bk0 * SB + 4 * bc0) = make_float4(bx0, by0, bz0, bw0);             \
        *reinterpret_cast<float4*>(dB + (bk0 + 8) * SB + 4 * bc0) = make_float4(bx1, by1, bz1, bw1);       \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int KT = a.Kpad / BK;
    P3_LOAD();
    P3_STORE(0);
    if (KT > 1) P3_LOAD();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int kh_lane = lane >> 5, l31 = lane & 31;
    const int fa = kh_lane * SA + wm * TM * 32 + l31;          // fragment offsets inside a tile buffer
    const int fb = BK * SA + kh_lane * SB + wn * TN * 32 + l31;
    float av[2][TM], bv[2][TN];
    {
        const float* t0 = smem;
#pragma unroll
        for (int i = 0; i < TM; i++) av[0][i] = t0[fa + i * 32];
#pragma unroll
        for (int j = 0; j < TN; j++) bv[0][j] = t0[fb + j * 32];
    }
    int cur = 0;
    for (int kt = 0; kt < KT; kt++) {
        const int nxtb = cur == 2 ? 0 : cur + 1;
        const float* tc = smem + cur * TILE_F;
        const float* tn = smem + nxtb * TILE_F;
        const bool has1 = kt + 1 < KT, has2 = kt + 2 < KT;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ks++) {
            const int c_ = ks & 1, n_ = c_ ^ 1;
            if (ks + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; i++) av[n_][i] = tc[fa + (2 * ks + 2) * SA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; j++) bv[n_][j] = tc[fb + (2 * ks + 2) * SB + j * 32];
            } else if (has1) {     // first fragments of the next tile (visible since the barrier at k-step 5)
#pragma unroll
                for (int i = 0; i < TM; i++) av[n_][i] = tn[fa + i * 32];
#pragma unroll
                for (int j = 0; j < TN; j++) bv[n_][j] = tn[fb + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c_][i], bv[c_][j], acc[i][j], 0, 0, 0);
            if (ks == 1 && has1) P3_STORE(nxtb);
            if (ks == 2 && has2) P3_LOAD();
            if (ks == 5) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        cur = nxtb;
    }
#undef P3_LOAD
#undef P3_STORE

    // ---- fused epilogue: identical to conv_mfma.hip ----
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
    const bool relu = a.relu != 0;
    const int Mlast = Mv - 1;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        const bool nok = n < a.Cout;
        const int nc = nok ? n : 0;
        const float bs = a.bias ? a.bias[nc] : 0.0f;
        const float sc = a.scale ? a.scale[nc] : 1.0f;
        const float sh = a.scale ? a.shift[nc] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int mbase = m0 + wm * TM * 32 + i * 32 + 4 * kh_lane;
            float extra[16];
            if (EPI != 0) {
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
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                float val = acc[i][j][r];
                val = val + bs;
                val = val * sc;
                val = val + sh;
                if (EPI != 0) val = val + extra[r];
                if (relu) val = val > 0.0f ? val : 0.0f;
                if (m < Mv && nok) out_v[(long long)m * out_ld + n] = val;
            }
        }
    }
}

// returns true if this variant handled the launch
bool launch_conv_p3(const ConvArgs& a, hipStream_t stream) {
    if (a.CoutPad % 128 != 0 || a.Cin % 16 != 0 || a.KH * a.KW > 32) return false;
    dim3 grid((unsigned)(a.total_mtiles * (a.CoutPad / 128))), block(256);
    static const int dl = getenv("CALD_CONV_DYNLDS") ? atoi(getenv("CALD_CONV_DYNLDS")) : 0;
    if (a.residual) hipLaunchKernelGGL((conv_p3_kernel<1>), grid, block, dl, stream, a);
    else if (a.up) hipLaunchKernelGGL((conv_p3_kernel<2>), grid, block, dl, stream, a);
    else hipLaunchKernelGGL((conv_p3_kernel<0>), grid, block, dl, stream, a);
    return true;
}
