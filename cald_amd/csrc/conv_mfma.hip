// conv_mfma.hip -- implicit-GEMM convolution / linear layer on the CDNA4 fp32 matrix cores.
//
// Replaces every cuDNN/cuBLAS call on the reference's detector forward (SURVEY.md section 2a:
// ResNet body, FPN, RPN head, TwoMLPHead, FastRCNNPredictor; reference call sites
// detection/frcnn_la.py:258, :261, :113-114).
//
//   GEMM view:  M = output pixels of all views of the ragged batch (each view padded to a multiple
//               of the 128-row tile), N = Cout, K = KH*KW*Cin in (kh, kw, cin) order.
//   MFMA:       v_mfma_f32_32x32x2_f32 -- exact fp32, one k-ordered fma chain per output, so the
//               result is bit-identical to the CPU oracle's fmaf chain (DESIGN.md contract).
//   Tile:       128 x BN x 16, 256 threads = 4 waves; each wave owns TM x TN 32x32 accumulators.
//   LDS:        A staged k-major [16][130] (stride 130 makes the transposing ds_write_b32 of the
//               im2col gather conflict-free), B [16][BN]; double buffered, one barrier per k-tile.
//   Pipeline:   global loads of k-tile t+1 are issued (unconditionally, clamped address + select)
//               before the MFMAs of tile t and land in registers while they run; LDS fragments of
//               k-step s+1 are read while the MFMAs of step s issue.
//   Mapping:    blocks that share an A tile (same M tile, different N tile) are placed on one XCD
//               (block b runs on XCD b % 8) so the im2col gather is fetched into one L2 only.
//   Epilogue:   (+bias) -> (*bn_scale, +bn_shift) -> (+residual) -> (+nearest-upsampled top-down)
//               -> ReLU, fused; residual rows are loaded as a batch before the stores.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WM, int WN, int TM, int TN, int EPI>
__global__ __launch_bounds__(256, 3) void conv_mfma_f32_kernel(const ConvArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 16;
    constexpr int SA = BM + 2;
    constexpr int SB = BN;
    constexpr int B_F4 = BK * BN / 4;                 // float4 in one B tile: 512 / 256 / 128
    constexpr int PB = (B_F4 + 255) / 256;
    static_assert(WM * WN == 4 && BM == 128, "4 waves, 128-row tile");
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * SA + 2 * BK * SB];
    float* const sA = smem;
    float* const sB = smem + 2 * BK * SA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int NT = a.CoutPad / BN;
    // ---- XCD-aware block -> (M tile, N tile) map ----
    int mt, nt;
    {
        const int b = blockIdx.x, MT = a.total_mtiles, MT8 = MT & ~7;
        if (b < MT8 * NT) {
            const int xcd = b & 7, idx = b >> 3;
            mt = (idx / NT) * 8 + xcd; nt = idx % NT;
        } else {
            const int r = b - MT8 * NT;
            mt = MT8 + r / NT; nt = r % NT;
        }
    }
    const int n0 = nt * BN;

    // ---- which view does this M tile belong to (wave-uniform scan of the plan) ----
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

    // ---- per-thread A gather state: rows (tid>>2) and (tid>>2)+64, k group g = tid&3 ----
    const int g = tid & 3, arow = tid >> 2;
    const int mA0 = m0 + arow, mA1 = mA0 + 64;
    const bool rv0 = mA0 < Mv, rv1 = mA1 < Mv;
    const int oyA0 = mA0 / Wo, oxA0 = mA0 - oyA0 * Wo;
    const int oyA1 = mA1 / Wo, oxA1 = mA1 - oyA1 * Wo;
    const int iy00 = oyA0 * a.stride - a.pad, ix00 = oxA0 * a.stride - a.pad;
    const int iy01 = oyA1 * a.stride - a.pad, ix01 = oxA1 * a.stride - a.pad;
    int ci = (4 * g) % Cin;
    int kh, kw;
    { const int tap = (4 * g) / Cin; kh = tap / KW; kw = tap - kh * KW; }
    // B tile mapping
    const int bk0 = tid / (BN / 4), bc0 = tid % (BN / 4);            // first float4
    const int bk1 = (tid + 256) / (BN / 4), bc1 = (tid + 256) % (BN / 4);
    const bool b0ok = (B_F4 >= 256) || (tid < B_F4);

    float4 ra0, ra1, rb0, rb1;
    const int KT = a.Kpad / BK;

#define LOAD_TILE(KTI)                                                                                     \
    {                                                                                                      \
        const int iy0 = iy00 + kh, ix0 = ix00 + kw, iy1 = iy01 + kh, ix1 = ix01 + kw;                      \
        const bool ok0 = rv0 && kh < KH && iy0 >= 0 && iy0 < Hi && ix0 >= 0 && ix0 < Wi;                   \
        const bool ok1 = rv1 && kh < KH && iy1 >= 0 && iy1 < Hi && ix1 >= 0 && ix1 < Wi;                   \
        const long long o0 = ok0 ? ((long long)(iy0 * Wi + ix0) * Cin + ci) : 0ll;                         \
        const long long o1 = ok1 ? ((long long)(iy1 * Wi + ix1) * Cin + ci) : 0ll;                         \
        const float4 t0 = *reinterpret_cast<const float4*>(in_v + o0);                                     \
        const float4 t1 = *reinterpret_cast<const float4*>(in_v + o1);                                     \
        ra0.x = ok0 ? t0.x : 0.0f; ra0.y = ok0 ? t0.y : 0.0f; ra0.z = ok0 ? t0.z : 0.0f; ra0.w = ok0 ? t0.w : 0.0f; \
        ra1.x = ok1 ? t1.x : 0.0f; ra1.y = ok1 ? t1.y : 0.0f; ra1.z = ok1 ? t1.z : 0.0f; ra1.w = ok1 ? t1.w : 0.0f; \
        if (in_relu) {                                                                                     \
            ra0.x = ra0.x < 0.0f ? 0.0f : ra0.x; ra0.y = ra0.y < 0.0f ? 0.0f : ra0.y; ra0.z = ra0.z < 0.0f ? 0.0f : ra0.z; ra0.w = ra0.w < 0.0f ? 0.0f : ra0.w; \
            ra1.x = ra1.x < 0.0f ? 0.0f : ra1.x; ra1.y = ra1.y < 0.0f ? 0.0f : ra1.y; ra1.z = ra1.z < 0.0f ? 0.0f : ra1.z; ra1.w = ra1.w < 0.0f ? 0.0f : ra1.w; \
        }                                                                                                  \
        if (b0ok) rb0 = *reinterpret_cast<const float4*>(wgt + (long long)((KTI) * BK + bk0) * CoutPad + n0 + 4 * bc0); \
        if (PB > 1) rb1 = *reinterpret_cast<const float4*>(wgt + (long long)((KTI) * BK + bk1) * CoutPad + n0 + 4 * bc1); \
        ci += BK;                                                                                          \
        while (ci >= Cin) { ci -= Cin; kw++; if (kw == KW) { kw = 0; kh++; } }                             \
    }
#define STORE_TILE(BUF)                                                                                    \
    {                                                                                                      \
        float* dA = sA + (BUF) * BK * SA + (4 * g) * SA + arow;                                            \
        dA[0] = ra0.x; dA[SA] = ra0.y; dA[2 * SA] = ra0.z; dA[3 * SA] = ra0.w;                             \
        dA[64] = ra1.x; dA[SA + 64] = ra1.y; dA[2 * SA + 64] = ra1.z; dA[3 * SA + 64] = ra1.w;             \
        float* dB = sB + (BUF) * BK * SB;                                                                  \
        if (b0ok) *reinterpret_cast<float4*>(dB + bk0 * SB + 4 * bc0) = rb0;                               \
        if (PB > 1) *reinterpret_cast<float4*>(dB + bk1 * SB + 4 * bc1) = rb1;                             \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    LOAD_TILE(0);
    STORE_TILE(0);
    __syncthreads();
    const int kh_lane = lane >> 5, l31 = lane & 31;
    for (int kt = 0; kt < KT; kt++) {
        const int buf = kt & 1;
        const bool more = kt + 1 < KT;
        if (more) LOAD_TILE(kt + 1);
        const float* cA = sA + buf * BK * SA + kh_lane * SA + wm * TM * 32 + l31;
        const float* cB = sB + buf * BK * SB + kh_lane * SB + wn * TN * 32 + l31;
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TM; i++) av[0][i] = cA[i * 32];
#pragma unroll
        for (int j = 0; j < TN; j++) bv[0][j] = cB[j * 32];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ks++) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; i++) av[nxt][i] = cA[(2 * ks + 2) * SA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; j++) bv[nxt][j] = cB[(2 * ks + 2) * SB + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
        }
        if (more) STORE_TILE(buf ^ 1);
        __syncthreads();
    }
#undef LOAD_TILE
#undef STORE_TILE

    // ---- fused epilogue (EPI: 0 = plain, 1 = +residual, 2 = +nearest-upsampled top-down) ----
    // bias / FrozenBN are applied unconditionally with neutral constants when absent: acc is never
    // -0.0 (the chain starts from +0), so x + 0.0f and x * 1.0f + 0.0f are exact identities.
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
                for (int r = 0; r < 16; r++) {   // all 16 loads of this 32x32 tile in flight together
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

template <int WM, int WN, int TM, int TN>
static void launch_cfg(const ConvArgs& a, int bn, hipStream_t stream) {
    dim3 grid((unsigned)(a.total_mtiles * (a.CoutPad / bn))), block(256);
    if (a.residual) hipLaunchKernelGGL((conv_mfma_f32_kernel<WM, WN, TM, TN, 1>), grid, block, 0, stream, a);
    else if (a.up) hipLaunchKernelGGL((conv_mfma_f32_kernel<WM, WN, TM, TN, 2>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((conv_mfma_f32_kernel<WM, WN, TM, TN, 0>), grid, block, 0, stream, a);
}

void launch_conv(const ConvArgs& a, hipStream_t stream) {
    if (a.CoutPad % 128 == 0) launch_cfg<2, 2, 2, 2>(a, 128, stream);
    else if (a.CoutPad % 64 == 0) launch_cfg<2, 2, 2, 1>(a, 64, stream);
    else launch_cfg<4, 1, 1, 1>(a, 32, stream);
}
