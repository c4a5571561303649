// conv_h4.hip -- the large-tile kernel of the "fp16 MFMA path" (CALD_PRECISION_F16X3, BASELINE.json configs[4]).
//
// Same arithmetic as conv_h3.hip (every fp32 operand split into two fp16 values, a * b = a_lo b_hi + a_hi b_lo + a_hi b_hi on
// v_mfma_f32_32x32x16_f16 into one fp32 accumulator), different data movement.  conv_h3's 128 x 128 tiles
// at three workgroups per CU pull 42 B / clk / CU through L2 -> VGPR -> LDS (16 KB per workgroup and k-step, each MFMA triple
// paying for its own operand bytes) and its matrix pipe idles 40 % of the time waiting for them.  Here:
//   * 256 x 256 x 16 tiles, ONE 512-thread workgroup per CU, 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave = 24 MFMAs
//     per wave and k-step between barriers (conv_h3: 12), half the operand bytes per MFMA;
//   * both operands arrive in split form (h16.h: [16 hi | 16 lo] per 16-channel chunk -- activations written so by their producer,
//     weights packed so at finalize) and go HBM / L2 -> LDS by buffer_load_dwordx4 ... lds: no VGPR, no VALU, no ds_write in the
//     k-loop.  An LDS row is the 64 bytes of one (row, k-step): four 16-byte pieces, piece p stored at slot p ^ ((row >> 2) & 3)
//     (the swizzle is applied on the SOURCE address -- the DMA's LDS image is lane-linear) so that every ds_read_b128 fragment
//     read is bank-conflict-free;
//   * a ring of four 32 KB stages (128 KB LDS), three k-steps of DMA in flight across the raw s_barrier of each step, counted
//     s_waitcnt vmcnt -- the ring is what hides the L2 latency, not occupancy; the four DMA pieces a wave issues per step sit
//     BETWEEN its MFMAs (a piece costs the wave 60-180 cycles of issue, which only the SIMD's other wave can cover);
//   * out-of-image taps: the buffer descriptor's range check returns zeros to the LDS for lanes whose offset is out of range.
// A workgroup covers two consecutive 128-row M tiles of the ragged batch (each lies inside one view; the halves may belong to
// different views), so the batch plan is the one every other kernel uses.  Bit-identical to conv_h3.hip (same MFMAs, same order).
// Layers it does not cover or does not pay for (Cout % 256, fp32-only input, in_relu, K < 1024, fewer than two rounds of
// workgroups) run on conv_h3.hip.
#include "h16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

#define H4_STAGE 32768
#define H4_NSTAGE 4

template <int EPI>
__device__ __forceinline__ void conv_h4_body(const ConvArgs& a, const int blk) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar branches, SGPR descriptors
    const int NT = a.CoutPad >> 8;
    const int MT2 = (a.total_mtiles + 1) >> 1;
    // XCD-contiguous map: the workgroups of one XCD (b % 8) walk a contiguous range of (M pair, N tile) -- neighbouring M tiles
    // share the halo rows of a 3 x 3 filter and the N tiles of an M pair share the whole A operand in that XCD's L2
    int mt2, nt;
    {
        const int total = MT2 * NT, b = blk, xcd = b & 7, idx = b >> 3, Q = total >> 3, R = total & 7;
        const int L = (xcd < R ? xcd * (Q + 1) : R * (Q + 1) + (xcd - R) * Q) + idx;
        mt2 = L / NT; nt = L - mt2 * NT;
    }
    const int n0 = nt << 8;
    const int Cin = a.Cin, KW = a.KW, KH = a.KH, CoutPad = a.CoutPad;

    // ---------------- loader role: waves 0-3 stage A rows [64 w, 64 w + 64), waves 4-7 stage B columns [64 (w - 4), ...) ----------------
    const bool ldA = wave < 4;
    const int lw = wave & 3;
    const int l_row = lane >> 2;                                  // row inside a 16-row DMA piece
    const int l_piece = (lane & 3) ^ ((lane >> 4) & 3);           // which 16-byte piece of the 64-byte row this lane fetches ((row >> 2) & 3 == (lane >> 4) & 3)
    int voff[4];                                                  // per DMA piece i: byte offset of this lane's source (A: the row's pixel, tap (0, 0))
    unsigned rowmask[4];                                          // A: taps of the row that lie inside the image
    __amdgpu_buffer_rsrc_t rs;
    int Wi_l = 1;
    if (ldA) {
        const int mt = 2 * mt2 + (lw >> 1);
        const int v = seg_find_view(a.seg_out, a.V, mt);
        const LevelSeg so = a.seg_out[v], si = a.seg_in[v];
        const int Wo = so.W, Hi = si.H, Wi = si.W;
        Wi_l = Wi;
        int Mv = so.H * Wo;
        if (a.dyn_rows) { const int d = a.dyn_rows[v]; Mv = d < Mv ? d : Mv; }
        if (mt >= a.total_mtiles) Mv = 0;
        const int mrow0 = (mt - so.tile_start) * 128 + (lw & 1) * 64;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = mrow0 + 16 * i + l_row;
            const int oy = m / Wo, ox = m - oy * Wo;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            unsigned msk = 0;
            if (m < Mv)
                for (int t = 0; t < KH * KW; t++) {
                    const int th = t / KW, tw = t - th * KW;
                    const int iy = iy0 + th, ix = ix0 + tw;
                    if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) msk |= 1u << t;
                }
            rowmask[i] = msk;
            voff[i] = ((oy * a.stride) * Wi + ox * a.stride) * Cin * 4 + l_piece * 16;
        }
        const unsigned char* in_v = reinterpret_cast<const unsigned char*>(a.in16) + (si.pix_off * (long long)Cin - (long long)a.pad * (Wi + 1) * Cin) * 4;
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)in_v, 0, 0x7FFE0000, 0x00020000);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int n = 64 * lw + 16 * i + l_row;
            rowmask[i] = 0xffffffffu;
            voff[i] = ((l_piece >> 1) * CoutPad + n) * 32 + (l_piece & 1) * 16;       // w16: [k-step][2 (hi, lo)][CoutPad][16]
        }
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const unsigned char*>(a.w16) + (long long)n0 * 32), 0, 0x7FFE0000, 0x00020000);
    }
    const int lds_l = (ldA ? 0 : 16384) + lw * 4096;             // this wave's 4 KB of a stage
    int u_kh = 0, u_kw = 0, u_ci = 0, u_kt = 0;                   // wave-uniform k cursor: (16-channel chunk, kh, kw), kw fastest

    // (branch-free on purpose: the steady-state k-step must be ONE basic block, or the scheduler cannot interleave the DMA issues and
    // the next step's fragment reads with the MFMAs -- all operands of the selects are wave-uniform, so they are s_cselect)
#define H4_ISSUE(SLOT)                                                                                                       \
    {                                                                                                                        \
        const int soffA = ((u_kh * Wi_l + u_kw) * Cin + u_ci) * 4, soffB = u_kt * (2 * CoutPad * 32);                        \
        const int soff = ldA ? soffA : soffB;                                                                                \
        const unsigned bit = ldA ? (1u << (u_kh * KW + u_kw)) : 1u;                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                                      \
            const int vo = (rowmask[i] & bit) ? voff[i] : 0x7FFF0000;                                                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (SLOT) * H4_STAGE + lds_l + i * 1024), 16, vo, soff, 0, 0); \
        }                                                                                                                    \
        u_kt++; u_kw++;                                                                                                      \
        const bool ww = u_kw == KW; u_kw = ww ? 0 : u_kw; u_kh += ww ? 1 : 0;                                                \
        const bool wh = u_kh == KH; u_kh = wh ? 0 : u_kh; u_ci += wh ? 16 : 0;                                               \
    }

    // ---------------- compute role: wave (wm, wn) owns rows [128 wm, +128) x columns [64 wn, +64) of the tile ----------------
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, kh_lane = lane >> 5, sw = (l31 >> 2) & 3;
    const int fa = (wm * 128 + l31) * 64, fb = 16384 + (wn * 64 + l31) * 64;
    const int s_hi = ((kh_lane) ^ sw) * 16, s_lo = ((2 + kh_lane) ^ sw) * 16;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int KT = a.Kpad >> 4;
    H4_ISSUE(0)
    if (KT > 1) H4_ISSUE(1)
    if (KT > 2) H4_ISSUE(2)
    if (KT > 3) H4_ISSUE(3)

#define H4_READ(AH, AL, BH, BL, SLOT)                                                                                        \
    {                                                                                                                        \
        const unsigned char* tb = smem + (SLOT) * H4_STAGE;                                                                  \
        _Pragma("unroll") for (int t = 0; t < 2; t++) {                                                                      \
            BH[t] = *reinterpret_cast<const h8*>(tb + fb + t * 2048 + s_hi);                                                 \
            BL[t] = *reinterpret_cast<const h8*>(tb + fb + t * 2048 + s_lo);                                                 \
        }                                                                                                                    \
        _Pragma("unroll") for (int t = 0; t < 4; t++) {                                                                      \
            AL[t] = *reinterpret_cast<const h8*>(tb + fa + t * 2048 + s_lo);                                                 \
            AH[t] = *reinterpret_cast<const h8*>(tb + fa + t * 2048 + s_hi);                                                 \
        }                                                                                                                    \
    }
    // One k-step.  Operands of stage kt are in (AH, AL, BH, BL) -- read from LDS during the previous step; this step reads stage
    // kt + 1 into (NAH, ...) under its own MFMAs and refills the slot stage kt just left with stage kt + 4.
#define H4_STEP(SLOT, AH, AL, BH, BL, NAH, NAL, NBH, NBL)                                                                    \
    {                                                                                                                        \
        /* this wave's pieces of stage kt + 1 have landed (two younger stages stay in flight), then everybody's have; every */ \
        /* wave has the fragments of stage kt in registers (lgkmcnt(0)), so its slot may be overwritten                      */ \
        const int later = KT - 2 - kt;                                                                                       \
        if (later >= 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                          \
        else if (later == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                                     \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                     \
        __builtin_amdgcn_s_barrier();                                                                                        \
        if (kt + 4 < KT) H4_ISSUE(SLOT)                                                                                      \
        if (kt + 1 < KT) H4_READ(NAH, NAL, NBH, NBL, ((SLOT) + 1) & 3)                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                        \
            _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                    \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BH[j], acc[i][j], 0, 0, 0);                        \
        _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                        \
            _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                    \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BL[j], acc[i][j], 0, 0, 0);                        \
        _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                        \
            _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                    \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BH[j], acc[i][j], 0, 0, 0);                        \
        kt++;                                                                                                                \
    }

    // Steady-state k-step (kt + 4 < KT): no conditionals, one basic block.  The 12 fragment reads of the NEXT step go out right after the
    // barrier; the four DMA pieces of stage kt + 4 are placed in program order after the 4th, 8th, 12th and 16th of the step's 24 MFMAs
    // (pinned with sched_barrier).  A piece costs the issuing wave ~60-180 cycles (MI355X_MICROARCH.md), which only the SIMD's other wave
    // can cover -- and with all four pieces right after the barrier both waves of a SIMD, which leave the barrier together, sit in
    // their DMA issues at the same time: SQ_VALU_MFMA_BUSY_CYCLES 73 of 128 per XCD-cycle then, 88 now (group launches).
// H4_VARIANT (compile time, tools/h4_dev.hip): 5 = shipped; 1 = the four pieces up front; 2 / 3 / 4 = TIMING ablations without the DMA /
// the barrier / the fragment reads -- their results are wrong by construction (profiles/r4_conv_h4_ablations.txt)
#ifndef H4_VARIANT
#define H4_VARIANT 5
#endif
#define H4_MFMA4(AX, BX, I0)                                                                                                 \
        _Pragma("unroll") for (int i = (I0); i < (I0) + 2; i++)                                                              \
            _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                    \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AX[i], BX[j], acc[i][j], 0, 0, 0);
#define H4_PIECE(SLOT, I, SOFF, BIT)                                                                                         \
        {                                                                                                                    \
            const int vo = (rowmask[I] & (BIT)) ? voff[I] : 0x7FFF0000;                                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (SLOT) * H4_STAGE + lds_l + (I) * 1024), 16, vo, (SOFF), 0, 0); \
        }
#define H4_STEPF(SLOT, AH, AL, BH, BL, NAH, NAL, NBH, NBL)                                                                   \
    {                                                                                                                        \
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                                          \
        if (H4_VARIANT != 3) __builtin_amdgcn_s_barrier();                                                                   \
        const int soffA = ((u_kh * Wi_l + u_kw) * Cin + u_ci) * 4, soffB = u_kt * (2 * CoutPad * 32);                        \
        const int soff = ldA ? soffA : soffB;                                                                                \
        const unsigned bit = ldA ? (1u << (u_kh * KW + u_kw)) : 1u;                                                          \
        if (H4_VARIANT != 4) H4_READ(NAH, NAL, NBH, NBL, ((SLOT) + 1) & 3)                                                   \
        if (H4_VARIANT == 5) {                                                                                               \
            /* pieces in program order BETWEEN the MFMAs: the two waves of a SIMD leave the barrier together, and with the  */ \
            /* four pieces up front both sit in their (60-180 cycle) DMA issues at once while the matrix pipe idles          */ \
            H4_MFMA4(AL, BH, 0) __builtin_amdgcn_sched_barrier(0); H4_PIECE(SLOT, 0, soff, bit) __builtin_amdgcn_sched_barrier(0);   \
            H4_MFMA4(AL, BH, 2) __builtin_amdgcn_sched_barrier(0); H4_PIECE(SLOT, 1, soff, bit) __builtin_amdgcn_sched_barrier(0);   \
            H4_MFMA4(AH, BL, 0) __builtin_amdgcn_sched_barrier(0); H4_PIECE(SLOT, 2, soff, bit) __builtin_amdgcn_sched_barrier(0);   \
            H4_MFMA4(AH, BL, 2) __builtin_amdgcn_sched_barrier(0); H4_PIECE(SLOT, 3, soff, bit) __builtin_amdgcn_sched_barrier(0);   \
            H4_MFMA4(AH, BH, 0) H4_MFMA4(AH, BH, 2)                                                                          \
        } else {                                                                                                             \
            if (H4_VARIANT != 2) { H4_PIECE(SLOT, 0, soff, bit) H4_PIECE(SLOT, 1, soff, bit) H4_PIECE(SLOT, 2, soff, bit) H4_PIECE(SLOT, 3, soff, bit) } \
            _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                    \
                _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BH[j], acc[i][j], 0, 0, 0);                    \
            _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                    \
                _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BL[j], acc[i][j], 0, 0, 0);                    \
            _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                    \
                _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BH[j], acc[i][j], 0, 0, 0);                    \
        }                                                                                                                    \
        u_kt++; u_kw++;                                                                                                      \
        const bool ww = u_kw == KW; u_kw = ww ? 0 : u_kw; u_kh += ww ? 1 : 0;                                                \
        const bool wh = u_kh == KH; u_kh = wh ? 0 : u_kh; u_ci += wh ? 16 : 0;                                               \
        kt++;                                                                                                                \
    }

    h8 ah0[4], al0[4], bh0[2], bl0[2], ah1[4], al1[4], bh1[2], bl1[2];
    if (KT > 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (KT > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (KT > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    H4_READ(ah0, al0, bh0, bl0, 0)
    int kt = 0;
    while (kt + 8 <= KT) {                      // every step of the body has kt + 4 < KT
        H4_STEPF(0, ah0, al0, bh0, bl0, ah1, al1, bh1, bl1)
        H4_STEPF(1, ah1, al1, bh1, bl1, ah0, al0, bh0, bl0)
        H4_STEPF(2, ah0, al0, bh0, bl0, ah1, al1, bh1, bl1)
        H4_STEPF(3, ah1, al1, bh1, bl1, ah0, al0, bh0, bl0)
    }
    for (int u = 0; u < 2; u++) {               // the last (up to seven) steps: the general form
        if (kt < KT) H4_STEP(0, ah0, al0, bh0, bl0, ah1, al1, bh1, bl1)
        if (kt < KT) H4_STEP(1, ah1, al1, bh1, bl1, ah0, al0, bh0, bl0)
        if (kt < KT) H4_STEP(2, ah0, al0, bh0, bl0, ah1, al1, bh1, bl1)
        if (kt < KT) H4_STEP(3, ah1, al1, bh1, bl1, ah0, al0, bh0, bl0)
    }
#undef H4_STEPF
#undef H4_PIECE
#undef H4_MFMA4
#undef H4_READ
#undef H4_ISSUE
#undef H4_STEP

    // ---------------- epilogue (h16.h) ----------------
    const int mt = 2 * mt2 + wm;
    if (mt >= a.total_mtiles) return;
    const int v = seg_find_view(a.seg_out, a.V, mt);
    const LevelSeg so = a.seg_out[v];
    int Mv = so.H * so.W;
    if (a.dyn_rows) { const int d = a.dyn_rows[v]; Mv = d < Mv ? d : Mv; }
    const int m0 = (mt - so.tile_start) * 128;
    if (m0 >= Mv) return;
    h16_epilogue<EPI, 4, 2>(a, acc, so, v, m0, n0 + wn * 64, Mv, lane);
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void conv_h4_kernel(const ConvArgs a) { conv_h4_body<EPI>(a, (int)blockIdx.x); }
// several independent problems of one shape class in one launch (ConvGroup): the small pyramid levels fill the tail of the large ones.
// Workgroups [blk0[i], blk0[i + 1]) belong to problem i; inside a problem the XCD map above applies to the problem-local index, which
// keeps its residue mod 8 when blk0[i] is a multiple of 8 (the launcher pads every problem's block range to one)
__global__ __launch_bounds__(512, 2) void conv_h4_group_kernel(const ConvGroup g) {
    int i = 0;
    while (i + 1 < g.n && g.blk0[i + 1] <= (int)blockIdx.x) i++;
    const int local = (int)blockIdx.x - g.blk0[i];
    const ConvArgs& a = g.p[i];
    if (local >= ((a.total_mtiles + 1) >> 1) * (a.CoutPad >> 8)) return;
    conv_h4_body<0>(a, local);
}

static bool h4_covers(const ConvArgs& a) {
    return a.w16 && a.in16 && !a.in_relu && a.CoutPad % 256 == 0 && a.Cin % 16 == 0 && a.KH * a.KW <= 32 && !a.mask && !(a.residual && a.up);
}
static int h4_mode() { static const int mode = getenv("CALD_H4") ? atoi(getenv("CALD_H4")) : 1; return mode; }   // 0: off, 1: where it fills the chip, 2: wherever it fits
static PerDeviceOnce h4_once[4];
// true if this kernel took the launch
bool launch_conv_h4(const ConvArgs& a, hipStream_t stream) {
    if (!h4_mode() || !h4_covers(a)) return false;
    const int wgs = ((a.total_mtiles + 1) >> 1) * (a.CoutPad >> 8);
    // where it pays, measured per layer class on BASELINE configs[4] with CALD_H4=2 against CALD_H4=0 (profiles/r4_h4_everywhere_vs_default.txt):
    // every class with K >= 1024 and at least two rounds of one workgroup per CU gains 4-9 % (3 x 3 layers of >= 256 channels, the
    // 1024 -> 256 / 2048 -> 512 reduce layers, fc6, the predictor); short chains (K <= 512: the expand layers with their residual
    // epilogue, the laterals) lose 10-30 % -- they want several residents per CU to hide prologue / epilogue phases (conv_h3)
    if (h4_mode() == 1 && (a.Kpad < 1024 || wgs < 512)) return false;
    const dim3 grid((unsigned)wgs), block(512);
    const size_t lds = (size_t)H4_NSTAGE * H4_STAGE;
    if (a.residual) { allow_big_lds(h4_once[1], conv_h4_kernel<1>); hipLaunchKernelGGL((conv_h4_kernel<1>), grid, block, lds, stream, a); }
    else if (a.up) { allow_big_lds(h4_once[2], conv_h4_kernel<2>); hipLaunchKernelGGL((conv_h4_kernel<2>), grid, block, lds, stream, a); }
    else { allow_big_lds(h4_once[0], conv_h4_kernel<0>); hipLaunchKernelGGL((conv_h4_kernel<0>), grid, block, lds, stream, a); }
    return true;
}
bool launch_conv_h4_group(const ConvArgs* p, int n, hipStream_t stream) {
    if (!h4_mode() || n < 1 || n > CALD_MAX_GROUP) return false;
    ConvGroup g; g.n = n; int blk = 0;
    for (int i = 0; i < n; i++) {
        if (!h4_covers(p[i]) || p[i].residual || p[i].up) return false;
        g.blk0[i] = blk; g.p[i] = p[i];
        blk += (((p[i].total_mtiles + 1) >> 1) * (p[i].CoutPad >> 8) + 7) & ~7;
    }
    g.blk0[n] = blk;
    if (h4_mode() == 1 && (blk < 512 || p[0].Kpad < 1024)) return false;
    allow_big_lds(h4_once[3], conv_h4_group_kernel);
    hipLaunchKernelGGL(conv_h4_group_kernel, dim3((unsigned)blk), dim3(512), (size_t)H4_NSTAGE * H4_STAGE, stream, g);
    return true;
}
