// conv_h3.hip -- the "fp16 MFMA path" of BASELINE.json configs[4], built to stay fp32-grade (a plain fp16 / bf16 pass
// is ~1e-3): every fp32 operand is split into two fp16 values,
//     x = hi + lo,  hi = fp16(x),  lo = fp16(x - hi)                      (22 significant bits)
// and a product a*b is evaluated as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  with three v_mfma_f32_32x32x16_f16 into
// the same fp32 accumulator (the a_lo*b_lo term, 2^-22 relative, is dropped).  The fp16 matrix pipe is 16x the
// fp32-input one (32 cycles per 32x32x16 vs 64 cycles per 32x32x2), so three passes are ~5x faster than
// conv_p4.hip's exact v_mfma_f32_32x32x2_f32 chain.  NOT bit-identical to the oracle (different rounding, ~1e-6
// end to end like any other fp32 implementation, so ~1 % of images change through a flipped borderline detection;
// DESIGN.md section 6) -- this mode is opt-in (cald_model_cfg.precision = 1); the default mode stays the exact one.
// fp16 has a narrow exponent range: the lo part of a small operand would be subnormal (absolute step 2^-24, e.g.
// only ~1e-6 relative for a weight of 0.03).  Both operands are therefore scaled by exact powers of two before the
// split -- weights by 2^S per layer at finalize (max |w| * 2^S <= 2^14), activations by 2^4 while staging -- and the
// accumulator is multiplied by 2^-(S+4) first thing in the epilogue (exact).  Range: |activation| < 4094.
//
// k order: the same (16-channel chunk, kh, kw) k-tile walk as the exact kernels (api.hip conv_k_index): all nine taps of a
// chunk are consecutive k-tiles, the three kw taps of a row read the same 64-byte pixel segments shifted by one pixel (L1
// hits) and the three rows stay in L2 -- the kernel is L2-bandwidth-bound otherwise (at 128 x 128 tiles the 3x3 256->256
// layer on P2 moves 74 GB per launch from L2, 10.6 TB/s).
//
// Same implicit-GEMM structure as conv_p4.hip: 128 x 128 x 16 tiles, 4 waves (64 x 64 each), buffer loads with
// hardware zero fill for out-of-image taps, XCD-aware tile map, fused fp32 epilogue (h16.h).  Activations arrive in split form
// (h16.h, written so by their producer: the loader's 16 bytes go to LDS with one ds_write_b128) or as fp32 (split while staging:
// the stem's input, tensors that also have fp32 readers); weights are split and packed at model finalize (ConvArgs::w16).
// Large launches with long chains run on conv_h4.hip instead (same arithmetic, bit for bit).
// LDS tile (16 KB): planes A_hi, A_lo, B_hi, B_lo of [128][16] fp16; a lane's MFMA operand (8 consecutive k of one
// row) is one ds_read_b128, 16-byte halves XOR-swizzled with bit 3 of the row (conflict-free).  Two tile buffers,
// one barrier per k-tile: the fragments of tile t+1 are read right after the barrier of tile t, under the
// a_hi*b_hi MFMAs of tile t.
#include "h16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// TN = 2: 128 x 128 tiles (64 x 64 per wave); TN = 1: 128 x 64 tiles (64 x 32 per wave) for 64-wide layers.
// C4: Cin == 4 (the stem conv on the NHWC4 input): a thread's four consecutive k are one filter tap, so the tap
// (kh, kw) and its validity are per-thread quantities instead of wave-uniform ones.
template <int EPI, int TN, bool C4>
__device__ __forceinline__ void conv_h3_body(const ConvArgs& a, const int blk) {
    constexpr int BM = 128, BN = 64 * TN, BK = 16, TM = 2, WN = 2;
    // the A lo plane starts 64 bytes (16 banks) past a multiple of the bank row: with the pre-split input a quad of lanes writes
    // {hi half 0, hi half 1, lo half 0, lo half 1} of ONE row with one ds_write_b128 each, and planes exactly 4 KB apart put the hi and
    // the lo pieces of a row on the same banks.  Measured +1.5 % on the 3 x 3 256 -> 256 layer (345-349 -> 353 TF-eq); the kernel's
    // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE still reads 0.20 -- it does for every kernel here that stores 16 bytes per lane to
    // LDS (conv_h4, which has no LDS stores, reads 0), so that figure is how the wide stores are accounted, not this pattern
    constexpr int PLANE = BM * 32 + 64, PLANE_B = BN * 32, TILE_B = 2 * PLANE + 2 * PLANE_B;       // bytes
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_B];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int NT = a.CoutPad / BN;
    int mt, nt;
    {
        const int b = blk, MT = a.total_mtiles, MT8 = MT & ~7;
        if (b < MT8 * NT) { const int xcd = b & 7, idx = b >> 3; mt = (idx / NT) * 8 + xcd; nt = idx % NT; }
        else { const int r = b - MT8 * NT; mt = MT8 + r / NT; nt = r % NT; }
    }
    const int n0 = nt * BN;
    const int v = seg_find_view(a.seg_out, a.V, mt);
    const LevelSeg so = a.seg_out[v];
    const LevelSeg si = a.seg_in[v];
    const int Ho = so.H, Wo = so.W, Hi = si.H, Wi = si.W;
    int Mv = Ho * Wo;
    if (a.dyn_rows) { const int d = a.dyn_rows[v]; Mv = d < Mv ? d : Mv; }
    const int m0 = (mt - so.tile_start) * BM;
    if (m0 >= Mv) return;
    // pre-split input (h16.h: [16 hi | 16 lo] per 16-channel chunk, the fp32 tensor's byte offsets): the loader fetches the same offsets;
    // its 16 bytes are then one 16-byte half of the hi or the lo plane of the LDS tile -- no arithmetic, one ds_write_b128
    const bool in16 = !C4 && a.in16 != nullptr;
    const float* __restrict__ in_v = (in16 ? reinterpret_cast<const float*>(a.in16) : a.in) + si.pix_off * (long long)a.Cin;
    const int Cin = a.Cin, KW = a.KW, KH = a.KH, CoutPad = a.CoutPad;
    const bool in_relu = a.in_relu != 0;

    // ---- A gather: rows arow, arow + 64; k group g = 4 consecutive k ----
    const int g = tid & 3, arow = tid >> 2;
    unsigned rowmask[2];
    int rowvoff[2], riy0[2], rix0[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int m = m0 + arow + 64 * p;
        const int oy = m / Wo, ox = m - oy * Wo;
        const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
        unsigned msk = 0;
        if (!C4 && m < Mv)
            for (int t = 0; t < KH * KW; t++) {
                const int th = t / KW, tw = t - th * KW;
                const int iy = iy0 + th, ix = ix0 + tw;
                if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) msk |= 1u << t;
            }
        rowmask[p] = msk;
        rowvoff[p] = (((oy * a.stride) * Wi + ox * a.stride) * Cin + (C4 ? 0 : 4 * g)) * 4;
        riy0[p] = m < Mv ? iy0 : -0x40000000; rix0[p] = ix0;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(in_v - (long long)a.pad * (Wi + 1) * Cin), 0, 0x7FFE0000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(a.w16) + (long long)n0 * 32), 0, 0x7FFE0000, 0x00020000);
    int u_kh = 0, u_kw = 0, u_ci = 0, u_kt = 0;
    // A LDS write offsets (bytes, hi plane): k = 4g .. 4g+3 -> 16-byte half g >> 1, 8-byte slot g & 1
    int aw_off[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int r = arow + 64 * p;
        aw_off[p] = r * 32 + (((g >> 1) ^ ((r >> 3) & 1)) * 16) + (g & 1) * 8;
    }
    int aw16_off[2];   // pre-split input: piece g -> plane g >> 1, 16-byte half g & 1 (same swizzle)
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int r = arow + 64 * p;
        aw16_off[p] = (g >> 1) * PLANE + r * 32 + (((g & 1) ^ ((r >> 3) & 1)) * 16);
    }
    // B: plane p (hi, lo) x n_l x 16-byte half.  TN = 2: two 16-byte pieces per thread (plane 0 and 1);
    // TN = 1: one piece per thread (plane = tid >> 7)
    const int b_t = TN == 2 ? tid : (tid & 127), b_p = TN == 2 ? 0 : (tid >> 7);
    const int b_nl = b_t >> 1, b_h = b_t & 1;
    const int bvoff0 = b_p * CoutPad * 32 + b_nl * 32 + b_h * 16, bvoff1 = bvoff0 + CoutPad * 32;
    const int bw_off0 = 2 * PLANE + b_p * PLANE_B + b_nl * 32 + ((b_h ^ ((b_nl >> 3) & 1)) * 16), bw_off1 = bw_off0 + PLANE_B;

    f32x4 ra0, ra1;
    i32x4 rb0, rb1 = {0, 0, 0, 0};

#define H3_LOAD()                                                                                          \
    {                                                                                                      \
        const int soffB = u_kt * 2 * CoutPad * 32;                                                         \
        int soffA, v0, v1;                                                                                 \
        if (C4) {                                                                                          \
            const int tap = u_kt * 4 + g, th = tap / KW, tw = tap - th * KW;                               \
            const int toff = (th * Wi + tw) * 16;                                                          \
            const bool ok0 = th < KH && (unsigned)(riy0[0] + th) < (unsigned)Hi && (unsigned)(rix0[0] + tw) < (unsigned)Wi; \
            const bool ok1 = th < KH && (unsigned)(riy0[1] + th) < (unsigned)Hi && (unsigned)(rix0[1] + tw) < (unsigned)Wi; \
            soffA = 0;                                                                                     \
            v0 = ok0 ? rowvoff[0] + toff : 0x7FFF0000;                                                     \
            v1 = ok1 ? rowvoff[1] + toff : 0x7FFF0000;                                                     \
        } else {                                                                                           \
            const unsigned u_bit = 1u << (u_kh * KW + u_kw);                                               \
            soffA = ((u_kh * Wi + u_kw) * Cin + u_ci) * 4;                                                 \
            v0 = (rowmask[0] & u_bit) ? rowvoff[0] : 0x7FFF0000;                                           \
            v1 = (rowmask[1] & u_bit) ? rowvoff[1] : 0x7FFF0000;                                           \
        }                                                                                                  \
        ra0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, v0, soffA, 0));         \
        ra1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, v1, soffA, 0));         \
        rb0 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff0, soffB, 0));     \
        if (TN == 2) rb1 = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff1, soffB, 0)); \
        u_kt++;                                                                                            \
        if (C4) { u_ci += BK; if (u_ci >= Cin) { u_ci = 0; u_kw++; if (u_kw == KW) { u_kw = 0; u_kh++; } } } \
        else { u_kw++; if (u_kw == KW) { u_kw = 0; u_kh++; if (u_kh == KH) { u_kh = 0; u_ci += BK; } } }  \
    }
#define H3_STORE(BUF)                                                                                      \
    {                                                                                                      \
        if (in_relu) {                                                                                     \
            _Pragma("unroll") for (int q = 0; q < 4; q++) { ra0[q] = ra0[q] < 0.f ? 0.f : ra0[q]; ra1[q] = ra1[q] < 0.f ? 0.f : ra1[q]; } \
        }                                                                                                  \
        unsigned char* tb = smem + (BUF) * TILE_B;                                                         \
        if (in16) {   /* split form (h16.h): this thread's 16 bytes are piece g of the row's 64-byte chunk = half g & 1 of plane g >> 1 */ \
            *reinterpret_cast<f32x4*>(tb + aw16_off[0]) = ra0;                                             \
            *reinterpret_cast<f32x4*>(tb + aw16_off[1]) = ra1;                                             \
        } else {                                                                                           \
        ra0 = ra0 * 16.0f; ra1 = ra1 * 16.0f;                                                              \
        const h4 hi0 = __builtin_convertvector(ra0, h4), hi1 = __builtin_convertvector(ra1, h4);           \
        const h4 lo0 = __builtin_convertvector(ra0 - __builtin_convertvector(hi0, f32x4), h4);             \
        const h4 lo1 = __builtin_convertvector(ra1 - __builtin_convertvector(hi1, f32x4), h4);             \
        *reinterpret_cast<h4*>(tb + aw_off[0]) = hi0;                                                      \
        *reinterpret_cast<h4*>(tb + PLANE + aw_off[0]) = lo0;                                              \
        *reinterpret_cast<h4*>(tb + aw_off[1]) = hi1;                                                      \
        *reinterpret_cast<h4*>(tb + PLANE + aw_off[1]) = lo1;                                              \
        }                                                                                                  \
        *reinterpret_cast<i32x4*>(tb + bw_off0) = rb0;                                                     \
        if (TN == 2) *reinterpret_cast<i32x4*>(tb + bw_off1) = rb1;                                        \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int KT = a.Kpad / BK;
    H3_LOAD();
    H3_STORE(0);
    if (KT > 1) H3_LOAD();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragment read offsets (bytes) inside a tile buffer: hi plane; lo plane = + PLANE
    const int kh_lane = lane >> 5, l31 = lane & 31;
    int fo_a[TM], fo_b[TN];
#pragma unroll
    for (int t = 0; t < TM; t++) { const int m = wm * 64 + t * 32 + l31; fo_a[t] = m * 32 + ((kh_lane ^ ((m >> 3) & 1)) * 16); }
#pragma unroll
    for (int t = 0; t < TN; t++) { const int n = wn * 32 * TN + t * 32 + l31; fo_b[t] = 2 * PLANE + n * 32 + ((kh_lane ^ ((n >> 3) & 1)) * 16); }
    h8 ah0[TM], al0[TM], bh0[TN], bl0[TN], ah1[TM], al1[TM], bh1[TN], bl1[TN];

#define H3_READ(AH, AL, BH, BL, TB)                                                                        \
    {                                                                                                      \
        _Pragma("unroll") for (int t = 0; t < TM; t++) {                                                   \
            AH[t] = *reinterpret_cast<const h8*>((TB) + fo_a[t]);                                          \
            AL[t] = *reinterpret_cast<const h8*>((TB) + PLANE + fo_a[t]);                                  \
        }                                                                                                  \
        _Pragma("unroll") for (int t = 0; t < TN; t++) {                                                   \
            BH[t] = *reinterpret_cast<const h8*>((TB) + fo_b[t]);                                          \
            BL[t] = *reinterpret_cast<const h8*>((TB) + PLANE_B + fo_b[t]);                                \
        }                                                                                                  \
    }
#define H3_MFMA(FA, FB)                                                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; i++)                                                         \
        _Pragma("unroll") for (int j = 0; j < TN; j++)                                                     \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[i], FB[j], acc[i][j], 0, 0, 0);
    // one k-tile: operands in (AH, AL, BH, BL); the next tile's operands are read into (NAH, ...)
#define H3_TILE(AH, AL, BH, BL, NAH, NAL, NBH, NBL)                                                        \
    {                                                                                                      \
        const int nxt = cur ^ 1;                                                                           \
        const bool has1 = kt + 1 < KT, has2 = kt + 2 < KT;                                                 \
        H3_MFMA(AL, BH)                                                                                    \
        if (has1) H3_STORE(nxt)                                                                            \
        H3_MFMA(AH, BL)                                                                                    \
        if (has2) H3_LOAD()                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                      \
        if (has1) H3_READ(NAH, NAL, NBH, NBL, smem + nxt * TILE_B)                                         \
        H3_MFMA(AH, BH)                                                                                    \
        cur = nxt; kt++;                                                                                   \
    }

    H3_READ(ah0, al0, bh0, bl0, smem)
    int cur = 0, kt = 0;
    while (kt + 1 < KT) {
        H3_TILE(ah0, al0, bh0, bl0, ah1, al1, bh1, bl1)
        H3_TILE(ah1, al1, bh1, bl1, ah0, al0, bh0, bl0)
    }
    if (kt < KT) H3_TILE(ah0, al0, bh0, bl0, ah1, al1, bh1, bl1)
#undef H3_LOAD
#undef H3_STORE
#undef H3_READ
#undef H3_MFMA
#undef H3_TILE

    // ---- fused fp32 epilogue (h16.h): the same operations as conv_p4.hip, fp32 and / or split-form stores ----
    h16_epilogue<EPI, TM, TN>(a, acc, so, v, m0 + wm * TM * 32, n0 + wn * TN * 32, Mv, lane);
}

template <int EPI, int TN, bool C4>
__global__ __launch_bounds__(256, 3) void conv_h3_kernel(const ConvArgs a) { conv_h3_body<EPI, TN, C4>(a, blockIdx.x); }
template <int EPI, int TN, bool C4>
__global__ __launch_bounds__(256, 3) void conv_h3_group_kernel(const ConvGroup g) {
    int i = 0;
    while (i + 1 < g.n && g.blk0[i + 1] <= (int)blockIdx.x) i++;
    conv_h3_body<EPI, TN, C4>(g.p[i], (int)blockIdx.x - g.blk0[i]);
}

bool launch_conv_h3_group(const ConvArgs* p, int n, hipStream_t stream) {
    if (n < 1 || n > CALD_MAX_GROUP) return false;
    const bool wide = p[0].CoutPad % 128 == 0;
    ConvGroup g; g.n = n; int blk = 0;
    for (int i = 0; i < n; i++) {
        const ConvArgs& a = p[i];
        if (!a.w16 || a.CoutPad % 64 != 0 || (a.CoutPad % 128 == 0) != wide || a.Cin % 16 != 0 || a.KH * a.KW > 32 || a.residual || a.up) return false;
        g.blk0[i] = blk; blk += a.total_mtiles * (a.CoutPad / (wide ? 128 : 64)); g.p[i] = a;
    }
    g.blk0[n] = blk;
    if (wide) hipLaunchKernelGGL((conv_h3_group_kernel<0, 2, false>), dim3((unsigned)blk), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((conv_h3_group_kernel<0, 1, false>), dim3((unsigned)blk), dim3(256), 0, stream, g);
    return true;
}

// returns true if this variant handled the launch
bool launch_conv_h3(const ConvArgs& a, hipStream_t stream) {
    if (!a.w16 || a.CoutPad % 64 != 0 || (a.in16 && a.in_relu)) return false;
    dim3 block(256);
    if (a.Cin == 4) {      // stem conv (7 x 7, NHWC4 input): per-thread taps
        if (a.residual || a.up || a.in_relu) return false;
        if (a.CoutPad % 128 == 0) hipLaunchKernelGGL((conv_h3_kernel<0, 2, true>), dim3((unsigned)(a.total_mtiles * (a.CoutPad / 128))), block, 0, stream, a);
        else hipLaunchKernelGGL((conv_h3_kernel<0, 1, true>), dim3((unsigned)(a.total_mtiles * (a.CoutPad / 64))), block, 0, stream, a);
        return true;
    }
    if (a.Cin % 16 != 0 || a.KH * a.KW > 32) return false;
    if (a.CoutPad % 128 == 0) {
        dim3 grid((unsigned)(a.total_mtiles * (a.CoutPad / 128)));
        if (a.residual) hipLaunchKernelGGL((conv_h3_kernel<1, 2, false>), grid, block, 0, stream, a);
        else if (a.up) hipLaunchKernelGGL((conv_h3_kernel<2, 2, false>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((conv_h3_kernel<0, 2, false>), grid, block, 0, stream, a);
    } else {
        dim3 grid((unsigned)(a.total_mtiles * (a.CoutPad / 64)));
        if (a.residual) hipLaunchKernelGGL((conv_h3_kernel<1, 1, false>), grid, block, 0, stream, a);
        else if (a.up) hipLaunchKernelGGL((conv_h3_kernel<2, 1, false>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((conv_h3_kernel<0, 1, false>), grid, block, 0, stream, a);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// cald_op_mfma_f16: the instruction itself on caller-supplied operands (parity hook for oracle/mfma_f16_model.h).  One wave evaluates 32
// dot products per instruction: case i of a block of 32 is row i of A and column i of B, D[i][i] its result; the tile is spilled to LDS
// and lanes 0..31 read the diagonal.
__global__ __launch_bounds__(256) void mfma_f16_probe_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                             const unsigned* __restrict__ C, unsigned* __restrict__ D, const long long n) {
    __shared__ float tile[4][32 * 33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long base = ((long long)blockIdx.x * 4 + wave) * 32;
    const int l31 = lane & 31, kh = lane >> 5;
    const long long ia = base + l31 < n ? base + l31 : n - 1;
    const h8 a = __builtin_bit_cast(h8, *reinterpret_cast<const i32x4*>(A + ia * 16 + 8 * kh));
    const h8 b = __builtin_bit_cast(h8, *reinterpret_cast<const i32x4*>(B + ia * 16 + 8 * kh));
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const long long ic = base + 4 * kh + (r & 3) + 8 * (r >> 2);
        c[r] = __builtin_bit_cast(float, C[ic < n ? ic : n - 1]);
    }
    const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) tile[wave][(4 * kh + (r & 3) + 8 * (r >> 2)) * 33 + l31] = d[r];
    __syncthreads();
    if (lane < 32 && base + lane < n) D[base + lane] = __builtin_bit_cast(unsigned, tile[wave][lane * 33 + lane]);
}
void launch_mfma_f16_probe(const unsigned short* A, const unsigned short* B, const unsigned* C, unsigned* D, long long n, hipStream_t stream) {
    const long long blocks = (n + 127) / 128;
    hipLaunchKernelGGL(mfma_f16_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, A, B, C, D, n);
}
