// conv_mfma.hip -- implicit-GEMM convolution / linear layer on the CDNA4 fp32 matrix cores.
//
// Replaces every cuDNN/cuBLAS call on the reference's detector forward (SURVEY.md section 2a:
// ResNet body, FPN, RPN head, TwoMLPHead, FastRCNNPredictor; reference call sites
// detection/frcnn_la.py:258, :261, :113-114).
//
//   GEMM view:  M = output pixels of all views of the ragged batch (each view padded to a multiple
//               of the 128-row tile), N = Cout, K = KH*KW*Cin in the contract's chain order (api.hip conv_k_index).
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
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WM, int WN, int TM, int TN, int EPI, int BK, bool FAST>
__global__ __launch_bounds__(256, (BK == 32 ? 2 : 3)) void conv_mfma_f32_kernel(const ConvArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int SA = BM + (BK == 16 ? 2 : 1);       // makes the transposing ds_write_b32 conflict-free
    constexpr int SB = BN;
    constexpr int KG = BK / 4;                        // float4 k-groups per A row
    constexpr int RP = 256 / KG;                      // A rows loaded per pass
    constexpr int PA = BM / RP;                       // A float4 per thread per k-tile
    constexpr int B_F4 = BK * BN / 4;                 // float4 in one B tile
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
    const int v = seg_find_view(a.seg_out, a.V, mt);
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
    const float* __restrict__ zpage = a.zeros;

    // ---- per-thread A gather state: rows arow + RP*p, k group g (4 consecutive k) ----
    const int g = tid % KG, arow = tid / KG;
    int iy0[PA], ix0[PA];
    bool rvalid[PA];
#pragma unroll
    for (int p = 0; p < PA; p++) {
        const int m = m0 + arow + RP * p;
        rvalid[p] = m < Mv;
        const int oy = m / Wo, ox = m - oy * Wo;
        iy0[p] = oy * a.stride - a.pad;
        ix0[p] = ox * a.stride - a.pad;
    }
    int ci = (4 * g) % Cin;
    int kh, kw;
    { const int tap = (4 * g) / Cin; kh = tap / KW; kw = tap - kh * KW; }
    // FAST path (Cin % BK == 0: a k-tile lies inside one tap): per-row tap-validity bit mask and element
    // offset are computed once; per k-tile only a wave-uniform offset is added (VALU instructions steal
    // issue cycles from the fp32 MFMA pipe, so the main loop keeps them to a minimum).
    unsigned rowmask[PA];
    int rowoff[PA];
    int u_kh = 0, u_kw = 0, u_ci = 0;                 // wave-uniform cursor
    if (FAST) {
#pragma unroll
        for (int p = 0; p < PA; p++) {
            unsigned msk = 0;
            if (rvalid[p])
                for (int t = 0; t < KH * KW; t++) {
                    const int th = t / KW, tw = t - th * KW;
                    const int iy = iy0[p] + th, ix = ix0[p] + tw;
                    if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) msk |= 1u << t;
                }
            rowmask[p] = msk;
            rowoff[p] = (iy0[p] * Wi + ix0[p]) * Cin + 4 * g;
        }
    }

    float rax[PA], ray[PA], raz[PA], raw[PA];
    float rbx[PB], rby[PB], rbz[PB], rbw[PB];
    const int KT = a.Kpad / BK;

#define LOAD_TILE(KTI)                                                                                     \
    {                                                                                                      \
        const int u_tap = u_kh * KW + u_kw;                                                                \
        const int u_off = (u_kh * Wi + u_kw) * Cin + u_ci;                                                 \
        _Pragma("unroll") for (int p = 0; p < PA; p++) {                                                   \
            /* out-of-image taps read a zero page: no select after the load, so it stays in flight */     \
            const float* src;                                                                              \
            if (FAST) {                                                                                    \
                const bool ok = (rowmask[p] >> u_tap) & 1u;                                                \
                src = ok ? (in_v + (rowoff[p] + u_off)) : zpage;                                           \
            } else {                                                                                       \
                const int iy = iy0[p] + kh, ix = ix0[p] + kw;                                              \
                const bool ok = rvalid[p] && kh < KH && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;          \
                src = ok ? (in_v + ((long long)(iy * Wi + ix) * Cin + ci)) : zpage;                        \
            }                                                                                              \
            const float4 t = *reinterpret_cast<const float4*>(src);                                        \
            rax[p] = t.x; ray[p] = t.y; raz[p] = t.z; raw[p] = t.w;                                        \
        }                                                                                                  \
        _Pragma("unroll") for (int p = 0; p < PB; p++) {                                                   \
            const int f = tid + 256 * p;                                                                   \
            if (B_F4 >= 256 * (p + 1) || f < B_F4) {                                                       \
                const float4 t = *reinterpret_cast<const float4*>(wgt + (long long)((KTI) * BK + f / (BN / 4)) * CoutPad + n0 + 4 * (f % (BN / 4))); \
                rbx[p] = t.x; rby[p] = t.y; rbz[p] = t.z; rbw[p] = t.w;                                    \
            }                                                                                              \
        }                                                                                                  \
        if (FAST) {                                                                                        \
            u_kw++; if (u_kw == KW) { u_kw = 0; u_kh++; if (u_kh == KH) { u_kh = 0; u_ci += BK; } }        \
        } else {                                                                                           \
            ci += BK;                                                                                      \
            while (ci >= Cin) { ci -= Cin; kw++; if (kw == KW) { kw = 0; kh++; } }                         \
        }                                                                                                  \
    }
#define STORE_TILE(BUF)                                                                                    \
    {                                                                                                      \
        float* dA = sA + (BUF) * BK * SA + (4 * g) * SA + arow;                                            \
        _Pragma("unroll") for (int p = 0; p < PA; p++) {                                                   \
            if (in_relu) {                                                                                 \
                rax[p] = rax[p] < 0.0f ? 0.0f : rax[p]; ray[p] = ray[p] < 0.0f ? 0.0f : ray[p];            \
                raz[p] = raz[p] < 0.0f ? 0.0f : raz[p]; raw[p] = raw[p] < 0.0f ? 0.0f : raw[p];            \
            }                                                                                              \
            dA[RP * p] = rax[p]; dA[SA + RP * p] = ray[p]; dA[2 * SA + RP * p] = raz[p]; dA[3 * SA + RP * p] = raw[p]; \
        }                                                                                                  \
        float* dB = sB + (BUF) * BK * SB;                                                                  \
        _Pragma("unroll") for (int p = 0; p < PB; p++) {                                                   \
            const int f = tid + 256 * p;                                                                   \
            if (B_F4 >= 256 * (p + 1) || f < B_F4)                                                         \
                *reinterpret_cast<float4*>(dB + (f / (BN / 4)) * SB + 4 * (f % (BN / 4))) = make_float4(rbx[p], rby[p], rbz[p], rbw[p]); \
        }                                                                                                  \
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

static int conv_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
template <int WM, int WN, int TM, int TN, int BK>
static void launch_cfg(const ConvArgs& a, int bn, hipStream_t stream) {
    dim3 grid((unsigned)(a.total_mtiles * (a.CoutPad / bn))), block(256);
    static const int dl = conv_env("CALD_CONV_DYNLDS", 0);   // extra dynamic LDS (experiments): caps workgroups per CU
    const bool fast = (a.Cin % BK == 0) && (a.KH * a.KW <= 32);
    if (!fast) { hipLaunchKernelGGL((conv_mfma_f32_kernel<WM, WN, TM, TN, 0, BK, false>), grid, block, dl, stream, a); return; }
    if (a.residual) hipLaunchKernelGGL((conv_mfma_f32_kernel<WM, WN, TM, TN, 1, BK, true>), grid, block, dl, stream, a);
    else if (a.up) hipLaunchKernelGGL((conv_mfma_f32_kernel<WM, WN, TM, TN, 2, BK, true>), grid, block, dl, stream, a);
    else hipLaunchKernelGGL((conv_mfma_f32_kernel<WM, WN, TM, TN, 0, BK, true>), grid, block, dl, stream, a);
}

bool launch_conv_p4(const ConvArgs& a, hipStream_t stream);   // conv_p4.hip
bool launch_conv_stem(const ConvArgs& a, hipStream_t stream); // conv_stem.hip (only when a.wstem is set)
bool launch_conv_h3(const ConvArgs& a, hipStream_t stream);   // conv_h3.hip (opt-in split-fp16 mode: only when a.w16 is set)

bool launch_conv_p4_group(const ConvArgs* p, int n, hipStream_t stream);
bool launch_conv_h3_group(const ConvArgs* p, int n, hipStream_t stream);
bool launch_conv_h4(const ConvArgs& a, hipStream_t stream);   // conv_h4.hip (the 256 x 256 LDS-DMA kernel of the same mode, where it fills the chip)
bool launch_conv_h4_group(const ConvArgs* p, int n, hipStream_t stream);
void launch_conv(const ConvArgs& a, hipStream_t stream);
int launch_conv_group(const ConvArgs* probs, int n, hipStream_t stream) {   // returns the number of kernel launches issued
    static const int grp = conv_env("CALD_CONV_GROUP", 1);
    static const int p4 = conv_env("CALD_CONV_P4", 1);
    if (grp && n > 1) {
        if (probs[0].w16 && launch_conv_h4_group(probs, n, stream)) return 1;
        if (probs[0].w16 && launch_conv_h3_group(probs, n, stream)) return 1;
        if (!probs[0].w16 && p4 && launch_conv_p4_group(probs, n, stream)) return 1;
    }
    for (int i = 0; i < n; i++) launch_conv(probs[i], stream);
    return n;
}

void launch_conv(const ConvArgs& a, hipStream_t stream) {
    static const int p4 = conv_env("CALD_CONV_P4", 1);   // conv_p4.hip: 3-buffer pipelined schedule, 128-bit LDS fragment reads; 0 = this file only
    if (a.w16 && launch_conv_h4(a, stream)) return;
    if (a.w16 && launch_conv_h3(a, stream)) return;
    static const int stem = conv_env("CALD_CONV_STEM", 1);   // 0: the stem runs on the generic kernels (same bits)
    if (stem && a.wstem && launch_conv_stem(a, stream)) return;
    if (p4 && launch_conv_p4(a, stream)) return;
    if (a.row_map) {                   // gathered rows exist in conv_p4.hip only (api.hip switches the pruning off when CALD_CONV_P4=0):
                                         // never compute the wrong rows silently -- the sweep's bound check then sends it to the dense head
        fprintf(stderr, "cald: a gathered conv launch (Cin=%d Cout=%d) found no kernel with row_map support; launch skipped\n", a.Cin, a.Cout);
        return;
    }
    if (a.CoutPad % 128 == 0) launch_cfg<2, 2, 2, 2, 16>(a, 128, stream);
    else if (a.CoutPad % 64 == 0) launch_cfg<2, 2, 2, 1, 16>(a, 64, stream);
    else launch_cfg<4, 1, 1, 1, 16>(a, 32, stream);
}
