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
//   Epilogue:   (+bias) -> (*bn_scale, +bn_shift) -> (+residual) -> (+nearest-upsampled top-down)
//               -> ReLU, fused; NHWC stores are 128 B contiguous per half-wave.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_mfma_f32_kernel(ConvArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 16;
    constexpr int SA = BM + 2;
    constexpr int SB = BN;
    constexpr int PA = BM / 64;                       // float4 A loads per thread per k-tile
    constexpr int B_F4 = BK * BN / 4;                 // float4 in one B tile
    constexpr int PB = (B_F4 + 255) / 256;
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * SA + 2 * BK * SB];
    float* sA = smem;
    float* sB = smem + 2 * BK * SA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int NT = a.CoutPad / BN;
    const int nt = blockIdx.x % NT, mt = blockIdx.x / NT;
    const int n0 = nt * BN;

    // ---- which view does this M tile belong to (wave-uniform scan of the plan) ----
    int v = 0;
    while (v + 1 < a.V && a.seg_out[v + 1].tile_start <= mt) v++;
    const LevelSeg so = a.seg_out[v];
    const LevelSeg si = a.seg_in[v];
    const int Ho = so.H, Wo = so.W, Hi = si.H, Wi = si.W;
    int Mv = Ho * Wo;
    if (a.dyn_rows) { int d = a.dyn_rows[v]; Mv = d < Mv ? d : Mv; }
    const int m0 = (mt - so.tile_start) * BM;
    if (m0 >= Mv) return;
    const float* in_v = a.in + si.pix_off * (long long)a.Cin;
    const int Cin = a.Cin, KW = a.KW, KH = a.KH;

    // ---- per-thread A gather state: rows (tid>>2) + 64*p, k group g = tid&3 (4 consecutive k) ----
    const int g = tid & 3;
    int iy0[PA], ix0[PA];
    bool rvalid[PA];
#pragma unroll
    for (int p = 0; p < PA; p++) {
        int m = m0 + (tid >> 2) + 64 * p;
        rvalid[p] = m < Mv;
        int oy = m / Wo, ox = m - oy * Wo;
        iy0[p] = oy * a.stride - a.pad;
        ix0[p] = ox * a.stride - a.pad;
    }
    int ci = (4 * g) % Cin, tap = (4 * g) / Cin;
    int kh = tap / KW, kw = tap - kh * KW;

    float4 ra[PA], rb[PB];
    const int KT = a.Kpad / BK;

    auto load_tile = [&](int kt) {
#pragma unroll
        for (int p = 0; p < PA; p++) {
            int iy = iy0[p] + kh, ix = ix0[p] + kw;
            bool ok = rvalid[p] && kh < KH && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) val = *reinterpret_cast<const float4*>(in_v + ((long long)(iy * Wi + ix) * Cin + ci));
            ra[p] = val;
        }
#pragma unroll
        for (int p = 0; p < PB; p++) {
            int f = tid + 256 * p;
            if (B_F4 >= 256 || f < B_F4) {
                int krow = f / (BN / 4), c4 = f % (BN / 4);
                rb[p] = *reinterpret_cast<const float4*>(a.w + (long long)(kt * BK + krow) * a.CoutPad + n0 + 4 * c4);
            }
        }
        // advance the (kh, kw, ci) cursor by one k-tile
        ci += BK;
        while (ci >= Cin) { ci -= Cin; kw++; if (kw == KW) { kw = 0; kh++; } }
    };
    auto store_tile = [&](int buf) {
        float* dA = sA + buf * BK * SA;
        float* dB = sB + buf * BK * SB;
#pragma unroll
        for (int p = 0; p < PA; p++) {
            int r = (tid >> 2) + 64 * p;
            dA[(4 * g + 0) * SA + r] = ra[p].x;
            dA[(4 * g + 1) * SA + r] = ra[p].y;
            dA[(4 * g + 2) * SA + r] = ra[p].z;
            dA[(4 * g + 3) * SA + r] = ra[p].w;
        }
#pragma unroll
        for (int p = 0; p < PB; p++) {
            int f = tid + 256 * p;
            if (B_F4 >= 256 || f < B_F4) {
                int krow = f / (BN / 4), c4 = f % (BN / 4);
                *reinterpret_cast<float4*>(dB + krow * SB + 4 * c4) = rb[p];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int kh_lane = lane >> 5, l31 = lane & 31;
    for (int kt = 0; kt < KT; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < KT) load_tile(kt + 1);
        const float* cA = sA + buf * BK * SA + wm * TM * 32 + l31;
        const float* cB = sB + buf * BK * SB + wn * TN * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ks++) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) av[i] = cA[(2 * ks + kh_lane) * SA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; j++) bv[j] = cB[(2 * ks + kh_lane) * SB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- fused epilogue ----
    float* out_v = a.out + so.pix_off * (long long)a.out_ld;
    const float* res_v = a.residual ? a.residual + so.pix_off * (long long)a.out_ld : nullptr;
    const float* up_v = nullptr;
    int upH = 0, upW = 0;
    float uph_s = 0.f, upw_s = 0.f;
    if (a.up) {
        const LevelSeg su = a.seg_up[v];
        up_v = a.up + su.pix_off * (long long)a.out_ld;
        upH = su.H; upW = su.W;
        uph_s = (float)upH / (float)Ho; upw_s = (float)upW / (float)Wo;
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        const bool nok = n < a.Cout;
        const float bs = (a.bias && nok) ? a.bias[n] : 0.0f;
        const float sc = (a.scale && nok) ? a.scale[n] : 1.0f;
        const float sh = (a.shift && nok) ? a.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh_lane;
                const int m = m0 + wm * TM * 32 + i * 32 + row;
                if (m < Mv && nok) {
                    float val = acc[i][j][r];
                    if (a.bias) val = val + bs;
                    if (a.scale) { val = val * sc; val = val + sh; }
                    const long long o = (long long)m * a.out_ld + n;
                    if (res_v) val = val + res_v[o];
                    if (up_v) {
                        int oy = m / Wo, ox = m - oy * Wo;
                        int sy = (int)floorf((float)oy * uph_s); if (sy > upH - 1) sy = upH - 1;
                        int sx = (int)floorf((float)ox * upw_s); if (sx > upW - 1) sx = upW - 1;
                        val = val + up_v[(long long)(sy * upW + sx) * a.out_ld + n];
                    }
                    if (a.relu) val = val > 0.0f ? val : 0.0f;
                    out_v[o] = val;
                }
            }
        }
    }
}

void launch_conv(const ConvArgs& a, hipStream_t stream) {
    dim3 block(256);
    if (a.CoutPad % 128 == 0) {
        dim3 grid((unsigned)(a.total_mtiles * (a.CoutPad / 128)));
        hipLaunchKernelGGL((conv_mfma_f32_kernel<2, 2, 2, 2>), grid, block, 0, stream, a);
    } else if (a.CoutPad % 64 == 0) {
        dim3 grid((unsigned)(a.total_mtiles * (a.CoutPad / 64)));
        hipLaunchKernelGGL((conv_mfma_f32_kernel<2, 2, 2, 1>), grid, block, 0, stream, a);
    } else {
        dim3 grid((unsigned)(a.total_mtiles * (a.CoutPad / 32)));
        hipLaunchKernelGGL((conv_mfma_f32_kernel<4, 1, 1, 1>), grid, block, 0, stream, a);
    }
}
