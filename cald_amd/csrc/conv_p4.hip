// conv_p4.hip -- the default implicit-GEMM conv / linear kernel: three LDS buffers, one raw s_barrier per k-tile placed
// inside the MFMA burst, LDS writes / global loads / next-tile fragment prefetch interleaved with the MFMAs, SRSRC buffer
// loads with hardware zero-fill for out-of-image taps, 128-bit LDS fragment reads.
//
// Same arithmetic as conv_mfma.hip and bit-identical results (one k-ordered fma chain per output):
//   * LDS tile layout [kq][row][h][j] (k = 8*kq + 2*j + h): the four A (or B) values a lane feeds into four
//     consecutive v_mfma_f32_32x32x2_f32 are one aligned 16-byte word -> one ds_read_b128 per (sub-tile, kq)
//     instead of four ds_read_b32; h is XOR-swizzled with bit 3 of the row so the 16-lane b128 groups are
//     conflict-free.  Weights are pre-packed in that order at model finalize (ConvArgs::w4), so B is staged with
//     coalesced 16-byte loads + ds_write_b128; A is staged with two ds_write_b64 per gathered float4.
//   Per k-tile and wave: 32 MFMA, 8 ds_read_b128, 6 LDS writes, 4 buffer loads, ~10 VALU.
#include "h16.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- fused epilogue (shared by the plain kernels and by the bottleneck-fused kernel's second GEMM) ----
template <int EPI, bool MASK, int TM, int TN>
__device__ __forceinline__ void p4_epilogue(const ConvArgs& a, const f32x16 (&acc)[TM][TN], float* smem, const int v, const LevelSeg& so,
                                            const int m0, const int Mv, const int n0) {
    constexpr int BM = 128, WN = 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int kh_lane = lane >> 5, l31 = lane & 31;
    const int Ho = so.H, Wo = so.W;
    // (same arithmetic as conv_mfma.hip): (+bias) -> (*bn_scale, +bn_shift) -> (+residual | +upsampled) -> ReLU ----
    // VALU instructions issued here take matrix-pipe time away from the two other workgroups of the CU, and for the short 1 x 1
    // chains (K = 64 ... 256) the epilogue is a large part of a workgroup's life: addresses are therefore kept off the VALU --
    // one byte offset per lane and accumulator tile, the 16 rows of a tile reached through the scalar offset of the buffer
    // instruction -- and absent bias / BN terms are skipped by wave-uniform branches instead of neutral operands.
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
    constexpr bool has_mask = MASK;
    const float* __restrict__ mask_v = has_mask ? a.mask + so.pix_off * (long long)out_ld : out_v;
    const int row_b = out_ld * 4;
    // The descriptors of everything addressed by output row END at the view's last valid row: a row past it loads zeros and its store is
    // dropped by the hardware, so a view's last, partial tile runs the same branch-free code as a full one (no per-element row tests, no
    // 64-bit addresses).  A lane of a padded output channel uses an offset beyond any descriptor.  (exp_flags bit 0, a tuning experiment:
    // a zero-sized output descriptor drops every store.)
    const unsigned long long vb64 = (unsigned long long)Mv * (unsigned)row_b;
    const unsigned valid_b = vb64 < 0x7FFE0000ull ? (unsigned)vb64 : 0x7FFE0000u;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)out_v, 0, (a.exp_flags & 1) ? 0 : valid_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI != 0 ? ex_v : out_v), 0, EPI == 1 ? valid_b : 0x7FFE0000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)mask_v, 0, valid_b, 0x00020000);
    // FPN top-down (EPI 2): the nearest-neighbour source pixel of each of the tile's 128 rows is computed ONCE (one thread per
    // row; the LDS tile buffers are free after the k-loop's last barrier) instead of by every lane for each of its 32 rows
    const int Mlast = Mv - 1;
    int* const s_src = reinterpret_cast<int*>(smem);
    if (EPI == 2) {
        if (tid < BM) {
            int m = m0 + tid;
            m = m < Mlast ? m : Mlast;
            const int oy = m / Wo, ox = m - oy * Wo;
            int sy = (int)floorf((float)oy * uph_s); sy = sy > upH - 1 ? upH - 1 : sy;
            int sx = (int)floorf((float)ox * upw_s); sx = sx > upW - 1 ? upW - 1 : sx;
            s_src[tid] = (sy * upW + sx) * out_ld * 4;
        }
        __syncthreads();
    }
    // The FPN output convs of P2 / P3 under the certified RPN pruning also leave (i) the split-fp16 copy of their output (h16.h) for the
    // look-ahead conv and (ii) per pixel the sum of squares over this wave's 64 channels for the bound -- what prune_energy_kernel did in a
    // second pass over the tensor.  Compiled into the plain (EPI 0, no mask) kernels only; wave-uniform branches elsewhere.
    constexpr bool EXTRA = EPI == 0 && !MASK;
    const bool want16 = EXTRA && a.out16 != nullptr, wantE = EXTRA && TN == 2 && a.energy4 != nullptr;      // energy4 has four slots: 2 n-tiles of 128 x 2 wave columns (Cout = 256)
    unsigned char* const out16_v = want16 ? reinterpret_cast<unsigned char*>(a.out16) + so.pix_off * (long long)out_ld * 4 : nullptr;
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(want16 ? (void*)out16_v : (void*)out_v), 0, valid_b, 0x00020000);
    const bool odd = (lane & 1) != 0;
    float sq[TM][16];
    if (EXTRA) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) sq[i][r] = 0.0f;
    }
    // Every load of the epilogue is issued BEFORE its first store: the per-channel bias / BN terms of both column tiles, then the residual
    // (or top-down) values of all TM x TN accumulator tiles.  Stores and loads go through different buffer resources the compiler cannot
    // tell apart, so a load written after a store stays after it -- tile by tile that was four dependent round trips to HBM per workgroup
    // (load 16, wait, store 16, load the next 16 ...), as long as the whole k-loop of a K = 128 layer.  64 + 64 live registers plus
    // temporaries fit the 168 of three workgroups per CU now that no path of the epilogue carries 64-bit addresses or row tests (PRE <
    // NTILE would request tile t + PRE between tile t's arithmetic and its stores).
    float bs_[TN], sc_[TN], sh_[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        const int nc = n < a.Cout ? n : 0;
        bs_[j] = has_bias ? a.bias[nc] : 0.0f;
        sc_[j] = has_bn ? a.scale[nc] : 1.0f;
        sh_[j] = has_bn ? a.shift[nc] : 0.0f;
    }
    constexpr int NTILE = TM * TN, PRE = NTILE;
    float extra_[EPI != 0 ? NTILE : 1][16];
    auto load_extra = [&](const int t, float (&extra)[16]) {      // tile t = j * TM + i
        const int j = t / TM, i = t - j * TM;
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        const bool nok = n < a.Cout;
        const int nc = nok ? n : 0;
        const int mbase = m0 + wm * TM * 32 + i * 32 + 4 * kh_lane;
        if (EPI == 1) {
            const int vo = nok ? (mbase * out_ld + n) * 4 : 0x7FFF0000;
#pragma unroll
            for (int r = 0; r < 16; r++)
                extra[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, vo, ((r & 3) + 8 * (r >> 2)) * row_b, 0));
        } else if (EPI == 2) {
            const int rl = wm * TM * 32 + i * 32 + 4 * kh_lane;          // tile-local row of accumulator register 0
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const i32x4 so4 = *reinterpret_cast<const i32x4*>(s_src + rl + 8 * q);
#pragma unroll
                for (int jj = 0; jj < 4; jj++)
                    extra[4 * q + jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, so4[jj] + nc * 4, 0, 0));
            }
        }
    };
    if (EPI != 0) {
#pragma unroll
        for (int t = 0; t < PRE; t++) load_extra(t, extra_[EPI != 0 ? t : 0]);
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        const bool nok = n < a.Cout;
        const int nc = nok ? n : 0;
        const float bs = bs_[j], sc = sc_[j], sh = sh_[j];
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int mbase = m0 + wm * TM * 32 + i * 32 + 4 * kh_lane;
            const float (&extra)[16] = extra_[EPI != 0 ? j * TM + i : 0];
            float val[16];
#pragma unroll
            for (int r = 0; r < 16; r++) val[r] = acc[i][j][r];
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
            if (EPI != 0 && j * TM + i + PRE < NTILE) load_extra(j * TM + i + PRE, extra_[EPI != 0 ? j * TM + i + PRE : 0]);
            const int vo = nok ? (mbase * out_ld + n) * 4 : 0x7FFF0000;
            if (has_mask) {      // training backward: ReLU backward of the layer this data gradient flows into (wave-uniform branch)
                float mk[16];
#pragma unroll
                for (int r = 0; r < 16; r++)
                    mk[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsM, vo, ((r & 3) + 8 * (r >> 2)) * row_b, 0));
#pragma unroll
                for (int r = 0; r < 16; r++) val[r] = mk[r] > 0.0f ? val[r] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; r++)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val[r]), rsO, vo, ((r & 3) + 8 * (r >> 2)) * row_b, 0);
            if (EXTRA) {
                if (wantE) {
#pragma unroll
                    for (int r = 0; r < 16; r++) sq[i][r] = sq[i][r] + (nok ? val[r] * val[r] : 0.0f);
                }
                if (want16) {       // lanes l and l ^ 1 hold neighbouring channels: the even lane stores {hi, hi}, the odd one {lo, lo} (h16.h)
                    const int poff = h16_pair_off(nc);
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const unsigned w = h16_split(val[r]);
                        const unsigned pw = h16_pair_word(w, h16_partner(w), odd);
                        __builtin_amdgcn_raw_buffer_store_b32(pw, rsS, nok ? mbase * row_b + poff : 0x7FFF0000, ((r & 3) + 8 * (r >> 2)) * row_b, 0);
                    }
                }
            }
        }
    }
    if (EXTRA) {
        if (wantE) {
            // sum over the wave's 32 column lanes, in lane order (a fixed order: the energy, and with it the pruning's selection, is the same
            // from run to run): each lane parks its 32 row partials in LDS (free after the k-loop), then lane L adds up row L of the wave's 64
            __syncthreads();
            float* const red = smem + wave * (64 * 33);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) red[(i * 32 + 4 * kh_lane + (r & 3) + 8 * (r >> 2)) * 33 + l31] = sq[i][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the wave's own LDS writes have landed (one wave = one reduction, no barrier)
            float e = 0.0f;
#pragma unroll
            for (int q = 0; q < 32; q++) e = e + red[lane * 33 + q];
            const int m = m0 + wm * TM * 32 + lane;
            if (m < Mv) a.energy4[(so.pix_off + m) * 4 + (n0 / (64 * TN)) * 2 + wn] = e;
        }
    }
}

// C4: Cin == 4 (the stem conv on the NHWC4 input).  A thread's four consecutive k are one filter tap, so the tap and its
// validity are per-thread quantities; the k order (tap, channel) is the contract's (kh, kw, cin) order unchanged.
// TN = 2: 128 x 128 tiles; TN = 1: 128 x 64 tiles (64 x 32 per wave) for the 64-wide layers.
// TAPS: 9 = 3 x 3 filter, 1 = 1 x 1 filter / linear layer: the k-loop is unrolled over the LDS buffer rotation (and the nine taps),
// so LDS addresses are register + immediate and the per-tap load offsets are precomputed registers -- the loop body holds
// no address / mask VALU at all (every VALU instruction beside v_mfma_f32_32x32x2_f32 costs matrix-pipe time: they share the
// fp32 datapath).  13 = the 7 x 7 stem on the 4-channel input (13 k-tiles, fully unrolled).  0 = generic rolled loop.
template <int EPIX, bool C4, int TN, int TAPS>
__device__ __forceinline__ void conv_p4_body(const ConvArgs& a, const int blk, const ConvArgs* a3 = nullptr) {
    constexpr int EPI = EPIX & 3;               // 0 none, 1 residual, 2 nearest-upsampled top-down
    constexpr bool MASK = (EPIX & 4) != 0;      // training backward only: the ReLU-backward mask step is compiled in
    constexpr bool FUSE = (EPIX & 8) != 0;      // bottleneck conv2 (3 x 3, 64 -> 64) + conv3 (1 x 1, 64 -> 256, + residual) in one workgroup
    constexpr int BM = 128, BN = 64 * TN, BK = 16, TM = 2, WN = 2;
    constexpr int TILE_A = 2 * BM * 8, TILE_F = TILE_A + 2 * BN * 8;      // floats: 2048 + 2048 (TN = 2)
    constexpr int FUSE_T = 4 * TILE_A;          // floats: conv2's 128 x 64 output tile as the A operand of four k-tiles (32 KB)
    __shared__ __attribute__((aligned(16))) float smem[FUSE ? FUSE_T + 4096 : 3 * TILE_F];      // fused: T + a 64-column weight chunk (48 KB)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int NT = a.CoutPad / BN;
    unsigned long long* const trace = (a.trace && blockIdx.x < (1u << 17)) ? a.trace + (size_t)blockIdx.x * 8 : nullptr;
    if (trace && tid == 0) {
        trace[0] = __builtin_amdgcn_s_memtime();
        trace[5] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
        trace[6] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
        trace[7] = blockIdx.x;
    }
    int mt, nt;
    if (a.KH * a.KW > 1) {
        // filters with a spatial extent: XCD-contiguous map: block b runs on XCD b % 8; XCD x owns the contiguous M-tile range [x * CH, (x + 1) * CH), so
        // tiles that share input rows (neighbours along a row, and the rows above / below ~W/128 tiles away) meet in ONE L2
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
    const int v = seg_find_view(a.seg_out, a.V, mt);
    const LevelSeg so = a.seg_out[v];
    const LevelSeg si = a.seg_in[v];
    const int Ho = so.H, Wo = so.W, Hi = si.H, Wi = si.W;
    int Mv = Ho * Wo;
    if (a.dyn_rows) { const int d = a.dyn_rows[v]; Mv = d < Mv ? d : Mv; }
    const int m0 = (mt - so.tile_start) * BM;
    if (m0 >= Mv) return;
    const float* __restrict__ in_v = a.in + si.pix_off * (long long)a.Cin;
    const int Cin = a.Cin, KW = a.KW, KH = a.KH, CoutPad = a.CoutPad;
    const bool in_relu = a.in_relu != 0;

    // ---- A gather (buffer resource, tap masks) : rows arow, arow + 64; k group g = 4 consecutive k ----
    const int g = tid & 3, arow = tid >> 2;
    unsigned rowmask[2];
    int rowvoff[2], riy0[2], rix0[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int m = m0 + arow + 64 * p;
        // gathered rows (ConvArgs::row_map): row m is the output pixel row_map[m] of the view's grid; everything below addresses the INPUT
        // from (oy, ox), the epilogue stores at row m
        const int mp = (a.row_map && m < Mv) ? a.row_map[so.pix_off + m] : m;
        const int oy = mp / Wo, ox = mp - oy * Wo;
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
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(a.w4 + (long long)n0 * 8), 0, 0x7FFE0000, 0x00020000);
    int u_kh = 0, u_kw = 0, u_ci = 0, u_kt = 0;
    // A LDS write offsets (floats): k = 4g + t -> kq = g >> 1, j = 2 (g & 1) + (t >> 1), h = t & 1
    const int a_kq = g >> 1, a_j = 2 * (g & 1);
    int aw_off[2][2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int r = arow + 64 * p;
#pragma unroll
        for (int h = 0; h < 2; h++) aw_off[p][h] = ((a_kq * BM + r) * 2 + (h ^ ((r >> 3) & 1))) * 4 + a_j;
    }
    // B: TN = 2: float4 f = tid + 256 p -> kq = p, n_l = tid >> 1, h = tid & 1 (two pieces per thread);
    //    TN = 1: one piece per thread, kq = tid >> 7
    const int b_t = TN == 2 ? tid : (tid & 127), b_kq = TN == 2 ? 0 : (tid >> 7);
    const int b_nl = b_t >> 1, b_h = b_t & 1;
    const int bvoff0 = (b_kq * CoutPad * 8 + b_nl * 8 + b_h * 4) * 4, bvoff1 = bvoff0 + CoutPad * 8 * 4;
    const int bw_off0 = TILE_A + ((b_kq * BN + b_nl) * 2 + (b_h ^ ((b_nl >> 3) & 1))) * 4, bw_off1 = bw_off0 + BN * 8;

    f32x4 ra0, ra1, rb0, rb1 = {0.f, 0.f, 0.f, 0.f};

#define P4_LOAD()                                                                                          \
    {                                                                                                      \
        const int soffB = u_kt * 2 * CoutPad * 8 * 4;                                                      \
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
        rb0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff0, soffB, 0));     \
        if (TN == 2) rb1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff1, soffB, 0)); \
        u_kt++;                                                                                            \
        /* k-tile order (16-channel chunk, kh, kw): api.hip conv_k_index */                                \
        u_kw++; if (u_kw == KW) { u_kw = 0; u_kh++; if (u_kh == KH) { u_kh = 0; u_ci += BK; } }            \
    }
#define P4_STORE(BUF)                                                                                      \
    {                                                                                                      \
        if (in_relu) {                                                                                     \
            _Pragma("unroll") for (int q = 0; q < 4; q++) { ra0[q] = ra0[q] < 0.f ? 0.f : ra0[q]; ra1[q] = ra1[q] < 0.f ? 0.f : ra1[q]; } \
        }                                                                                                  \
        float* tb = smem + (BUF) * TILE_F;                                                                 \
        *reinterpret_cast<float2*>(tb + aw_off[0][0]) = make_float2(ra0[0], ra0[2]);                       \
        *reinterpret_cast<float2*>(tb + aw_off[0][1]) = make_float2(ra0[1], ra0[3]);                       \
        *reinterpret_cast<float2*>(tb + aw_off[1][0]) = make_float2(ra1[0], ra1[2]);                       \
        *reinterpret_cast<float2*>(tb + aw_off[1][1]) = make_float2(ra1[1], ra1[3]);                       \
        *reinterpret_cast<f32x4*>(tb + bw_off0) = rb0;                                                     \
        if (TN == 2) *reinterpret_cast<f32x4*>(tb + bw_off1) = rb1;                                        \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int KT = a.Kpad / BK;
    // unrolled variants: per-thread load offset of every tap (out-of-image taps -> an out-of-range offset: the buffer load
    // returns zeros), and the loader's channel cursor
    int voffA[2][TAPS > 0 ? TAPS : 1];
    int ci_ld = 0;
    if (TAPS == 13) {
        // the 7 x 7 stem on the 4-channel input: 13 k-tiles of four taps; thread g owns tap 4 kt + g of k-tile kt -- its offset (or the
        // out-of-range offset for taps outside the image / beyond tap 48) is computed once here instead of in every k-tile
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int t = 0; t < 13; t++) {
                const int tap = 4 * t + g, th = tap / 7, tw = tap - th * 7;
                const bool ok = th < 7 && (unsigned)(riy0[p] + th) < (unsigned)Hi && (unsigned)(rix0[p] + tw) < (unsigned)Wi;
                voffA[p][t] = ok ? rowvoff[p] + (th * Wi + tw) * 16 : 0x7FFF0000;
            }
    } else if (TAPS > 0) {
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int t = 0; t < (TAPS > 0 ? TAPS : 1); t++) voffA[p][t] = ((rowmask[p] >> t) & 1u) ? rowvoff[p] : 0x7FFF0000;
    }
    if (TAPS == 0) {
        P4_LOAD();
        P4_STORE(0);
        if (KT > 1) P4_LOAD();
    } else {
#define U_LOAD0(T)                                                                                         \
    {                                                                                                      \
        const int soffB = u_kt * 2 * CoutPad * 8 * 4;                                                      \
        const int soffA = (TAPS == 9 ? ((((T) / 3) * Wi + ((T) % 3)) * Cin + ci_ld) : ci_ld) * 4;          \
        ra0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA[0][T], soffA, 0)); \
        ra1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA[1][T], soffA, 0)); \
        rb0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff0, soffB, 0));     \
        if (TN == 2) rb1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff1, soffB, 0)); \
        u_kt++;                                                                                            \
        if ((T) == TAPS - 1) ci_ld += BK;                                                                  \
    }
        U_LOAD0(0)
        P4_STORE(0);
        if (KT > 1) U_LOAD0((TAPS == 9 || TAPS == 13) ? 1 : 0)
#undef U_LOAD0
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (trace && tid == 0) trace[1] = __builtin_amdgcn_s_memtime();

    // fragment read offsets (floats) inside a tile buffer, kq = 0; kq = 1 adds BM*8 (A) / BN*8 (B)
    const int kh_lane = lane >> 5, l31 = lane & 31;
    int fo_a[TM], fo_b[TN];
#pragma unroll
    for (int t = 0; t < TM; t++) { const int m = wm * 64 + t * 32 + l31; fo_a[t] = (m * 2 + (kh_lane ^ ((m >> 3) & 1))) * 4; }
#pragma unroll
    for (int t = 0; t < TN; t++) { const int n = wn * 32 * TN + t * 32 + l31; fo_b[t] = TILE_A + (n * 2 + (kh_lane ^ ((n >> 3) & 1))) * 4; }
    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
#pragma unroll
    for (int t = 0; t < TM; t++) fa0[t] = *reinterpret_cast<const f32x4*>(smem + fo_a[t]);
#pragma unroll
    for (int t = 0; t < TN; t++) fb0[t] = *reinterpret_cast<const f32x4*>(smem + fo_b[t]);

#define P4_MFMA(FA, FB, Q)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; i++)                                                         \
        _Pragma("unroll") for (int j = 0; j < TN; j++)                                                     \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i][Q], FB[j][Q], acc[i][j], 0, 0, 0);
#define P4_TILE(CUR, NXT)                                                                                  \
    {                                                                                                      \
        const float* tc = smem + (CUR) * TILE_F;                                                           \
        const float* tn = smem + (NXT) * TILE_F;                                                           \
        const bool has1 = kt + 1 < KT, has2 = kt + 2 < KT;                                                 \
        _Pragma("unroll") for (int t = 0; t < TM; t++) fa1[t] = *reinterpret_cast<const f32x4*>(tc + fo_a[t] + BM * 8); \
        _Pragma("unroll") for (int t = 0; t < TN; t++) fb1[t] = *reinterpret_cast<const f32x4*>(tc + fo_b[t] + BN * 8); \
        P4_MFMA(fa0, fb0, 0)                                                                               \
        P4_MFMA(fa0, fb0, 1)                                                                               \
        if (has1) P4_STORE(NXT)                                                                            \
        P4_MFMA(fa0, fb0, 2)                                                                               \
        if (has2) P4_LOAD()                                                                                \
        P4_MFMA(fa0, fb0, 3)                                                                               \
        P4_MFMA(fa1, fb1, 0)                                                                               \
        P4_MFMA(fa1, fb1, 1)                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                      \
        if (has1) {                                                                                        \
            _Pragma("unroll") for (int t = 0; t < TM; t++) fa0[t] = *reinterpret_cast<const f32x4*>(tn + fo_a[t]); \
            _Pragma("unroll") for (int t = 0; t < TN; t++) fb0[t] = *reinterpret_cast<const f32x4*>(tn + fo_b[t]); \
        }                                                                                                  \
        P4_MFMA(fa1, fb1, 2)                                                                               \
        P4_MFMA(fa1, fb1, 3)                                                                               \
    }

    if (TAPS == 0) {
        int cur = 0;
        for (int kt = 0; kt < KT; kt++) {
            const int nxtb = cur == 2 ? 0 : cur + 1;
            P4_TILE(cur, nxtb);
            cur = nxtb;
        }
    } else {
        // LDS addresses as pointers: buffer / kq offsets below are compile-time constants folded into the DS offset field
        // A stores go out as ds_write2_b32 (8-bit dword offsets: no room for the buffer offset) -> one address register per (buffer, row, h)
        unsigned awa[3][4];
#pragma unroll
        for (int b = 0; b < 3; b++) {
            awa[b][0] = (unsigned)(uintptr_t)(smem + b * TILE_F + aw_off[0][0]); awa[b][1] = (unsigned)(uintptr_t)(smem + b * TILE_F + aw_off[0][1]);
            awa[b][2] = (unsigned)(uintptr_t)(smem + b * TILE_F + aw_off[1][0]); awa[b][3] = (unsigned)(uintptr_t)(smem + b * TILE_F + aw_off[1][1]);
        }
        float* const bwp0 = smem + bw_off0; float* const bwp1 = smem + bw_off1;
        const float* fpa[TM]; const float* fpb[TN];
#pragma unroll
        for (int t = 0; t < TM; t++) fpa[t] = smem + fo_a[t];
#pragma unroll
        for (int t = 0; t < TN; t++) fpb[t] = smem + fo_b[t];
#define U_LOAD(T)                                                                                          \
    {                                                                                                      \
        const int soffB = u_kt * 2 * CoutPad * 8 * 4;                                                      \
        const int soffA = (TAPS == 9 ? ((((T) / 3) * Wi + ((T) % 3)) * Cin + ci_ld) : ci_ld) * 4;          \
        ra0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA[0][T], soffA, 0)); \
        ra1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA[1][T], soffA, 0)); \
        rb0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff0, soffB, 0));     \
        if (TN == 2) rb1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bvoff1, soffB, 0)); \
        u_kt++;                                                                                            \
        if ((T) == TAPS - 1) ci_ld += BK;                                                                  \
    }
/* two dwords from two independent registers to consecutive LDS words (hipcc would merge plain stores into a ds_write_b64  \
   plus two v_mov; LDS completes in order, so the compiler's own lgkmcnt bookkeeping stays conservative-correct) */
#define U_W2(ADDR, D0, D1)                                                                                 \
    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(ADDR), "v"(D0), "v"(D1) : "memory");
#define U_STORE(BUF)                                                                                       \
    {                                                                                                      \
        if (in_relu) {                                                                                     \
            _Pragma("unroll") for (int q = 0; q < 4; q++) { ra0[q] = ra0[q] < 0.f ? 0.f : ra0[q]; ra1[q] = ra1[q] < 0.f ? 0.f : ra1[q]; } \
        }                                                                                                  \
        /* scalar stores: the pairs (k, k + 2) sit in non-adjacent registers -- ds_write2_b32 takes two independent data   \
           registers, a ds_write_b64 would need two v_mov per pair (VALU beside the MFMAs) */                           \
        U_W2(awa[BUF][0], ra0[0], ra0[2]) U_W2(awa[BUF][1], ra0[1], ra0[3])                                \
        U_W2(awa[BUF][2], ra1[0], ra1[2]) U_W2(awa[BUF][3], ra1[1], ra1[3])                                \
        *reinterpret_cast<f32x4*>(bwp0 + (BUF) * TILE_F) = rb0;                                            \
        if (TN == 2) *reinterpret_cast<f32x4*>(bwp1 + (BUF) * TILE_F) = rb1;                               \
    }
#define U_TILE(CUR, NXT, TLD, KTX)                                                                         \
    {                                                                                                      \
        const bool has1 = (KTX) + 1 < KT, has2 = (KTX) + 2 < KT;                                           \
        _Pragma("unroll") for (int t = 0; t < TM; t++) fa1[t] = *reinterpret_cast<const f32x4*>(fpa[t] + (CUR) * TILE_F + BM * 8); \
        _Pragma("unroll") for (int t = 0; t < TN; t++) fb1[t] = *reinterpret_cast<const f32x4*>(fpb[t] + (CUR) * TILE_F + BN * 8); \
        P4_MFMA(fa0, fb0, 0)                                                                               \
        P4_MFMA(fa0, fb0, 1)                                                                               \
        if (has1) U_STORE(NXT)                                                                             \
        P4_MFMA(fa0, fb0, 2)                                                                               \
        if (has2) U_LOAD(TLD)                                                                              \
        P4_MFMA(fa0, fb0, 3)                                                                               \
        P4_MFMA(fa1, fb1, 0)                                                                               \
        P4_MFMA(fa1, fb1, 1)                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                      \
        if (has1) {                                                                                        \
            _Pragma("unroll") for (int t = 0; t < TM; t++) fa0[t] = *reinterpret_cast<const f32x4*>(fpa[t] + (NXT) * TILE_F); \
            _Pragma("unroll") for (int t = 0; t < TN; t++) fb0[t] = *reinterpret_cast<const f32x4*>(fpb[t] + (NXT) * TILE_F); \
        }                                                                                                  \
        P4_MFMA(fa1, fb1, 2)                                                                               \
        P4_MFMA(fa1, fb1, 3)                                                                               \
    }
        if (TAPS == 13) {                              // the stem: all 13 k-tiles, k-tile index = load-offset index
            U_TILE(0, 1, 2, 0)  U_TILE(1, 2, 3, 1)  U_TILE(2, 0, 4, 2)  U_TILE(0, 1, 5, 3)  U_TILE(1, 2, 6, 4)
            U_TILE(2, 0, 7, 5)  U_TILE(0, 1, 8, 6)  U_TILE(1, 2, 9, 7)  U_TILE(2, 0, 10, 8) U_TILE(0, 1, 11, 9)
            U_TILE(1, 2, 12, 10) U_TILE(2, 0, 12, 11) U_TILE(0, 1, 12, 12)
        } else if (TAPS == 9) {
            for (int kt = 0; kt < KT; kt += 9) {      // one 16-channel chunk: nine taps, three turns of the buffer ring
                U_TILE(0, 1, 2, kt)     U_TILE(1, 2, 3, kt + 1) U_TILE(2, 0, 4, kt + 2)
                U_TILE(0, 1, 5, kt + 3) U_TILE(1, 2, 6, kt + 4) U_TILE(2, 0, 7, kt + 5)
                U_TILE(0, 1, 8, kt + 6) U_TILE(1, 2, 0, kt + 7) U_TILE(2, 0, 1, kt + 8)
            }
        } else {
            for (int kt = 0; kt < KT; kt += 3) {
                U_TILE(0, 1, 0, kt)
                if (kt + 1 < KT) U_TILE(1, 2, 0, kt + 1)
                if (kt + 2 < KT) U_TILE(2, 0, 0, kt + 2)
            }
        }
#undef U_LOAD
#undef U_STORE
#undef U_W2
#undef U_TILE
    }
#undef P4_LOAD
#undef P4_STORE
#undef P4_MFMA
#undef P4_TILE

    if (trace && tid == 0) trace[2] = __builtin_amdgcn_s_memtime();
    if constexpr (FUSE) {
        // ---- bottleneck tail: T = relu(bn2(acc)) (conv2's own epilogue arithmetic, bit for bit what the unfused kernel stores) never
        // leaves the CU -- it is laid down in LDS as the A operand of conv3 (plain channel order = conv3's chain order), conv3's weights
        // come in 128-column chunks, and each chunk runs the ordinary residual epilogue.  The 64-channel tensor is neither written nor
        // read back, and conv3's HBM-bound epilogue (residual in, 256 channels out) overlaps the other workgroup's 3 x 3 k-loop. ----
        const ConvArgs& b = *a3;
        float* const Tl = smem;
        float* const Wl = smem + FUSE_T;
        {
            const int c = wn * 32 + l31;                        // conv2 output channel of this lane (TN == 1, BN == 64 == Cout)
            const float sc = a.scale ? a.scale[c] : 1.0f, sh = a.scale ? a.shift[c] : 0.0f, bs = a.bias ? a.bias[c] : 0.0f;
            const bool relu2 = a.relu != 0, has_bias2 = a.bias != nullptr, has_bn2 = a.scale != nullptr;
            // element (row, c) of T sits at k-tile c >> 4, kq (c >> 3) & 1, j (c & 7) >> 1, h c & 1 (XOR-swizzled with bit 3 of the row).
            // row = R0 + 32 i + (r & 3) + 8 (r >> 2) with R0 = 64 wm + 4 kh_lane, so bit 3 of the row is bit 2 of r: two lane bases, the
            // rest is an immediate
            float* const tc = Tl + (c >> 4) * TILE_A + ((c >> 3) & 1) * BM * 8 + ((c & 7) >> 1) + (wm * 64 + 4 * kh_lane) * 8;
            float* const tcb[2] = {tc + (c & 1) * 4, tc + ((c & 1) ^ 1) * 4};
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float val = acc[i][0][r];
                    if (has_bias2) val = val + bs;
                    if (has_bn2) { val = val * sc; val = val + sh; }
                    if (relu2) val = val > 0.0f ? val : 0.0f;
                    tcb[(r >> 2) & 1][(i * 32 + (r & 3) + 8 * (r >> 2)) * 8] = val;
                }
        }
        const int CoutPad3 = b.CoutPad;
        const LevelSeg so3 = b.seg_out[v];
        const int ld3 = b.out_ld;
        const bool relu3 = b.relu != 0;
        const __amdgpu_buffer_rsrc_t rsO3 = __builtin_amdgcn_make_buffer_rsrc((void*)(b.out + so3.pix_off * (long long)ld3), 0, Mv * ld3 * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsX3 = __builtin_amdgcn_make_buffer_rsrc((void*)(b.residual + so3.pix_off * (long long)ld3), 0, Mv * ld3 * 4, 0x00020000);
        // conv3 in 64-column chunks (T 32 KB + a chunk's weights 16 KB = 48 KB of LDS: three workgroups per CU, like the plain
        // kernels); waves 2 x 2 over the 128 x 64 chunk, 64 x 32 each.  The next chunk's weights are fetched into registers while
        // this chunk multiplies and stores.
        int foa2[2], fob2;
#pragma unroll
        for (int t = 0; t < 2; t++) { const int m = wm * 64 + t * 32 + l31; foa2[t] = (m * 2 + (kh_lane ^ ((m >> 3) & 1))) * 4; }
        { const int n = wn * 32 + l31; fob2 = (n * 2 + (kh_lane ^ ((n >> 3) & 1))) * 4; }
        const int w_t = tid & 127, w_kq = tid >> 7, w_nl = w_t >> 1, w_h = w_t & 1;
        const int wvoff = (w_kq * CoutPad3 * 8 + w_nl * 8 + w_h * 4) * 4;
        const int ww_off = ((w_kq * 64 + w_nl) * 2 + (w_h ^ ((w_nl >> 3) & 1))) * 4;
        f32x4 wr[4];
        auto fetch_w = [&](int n0c) {
            const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(b.w4 + (long long)n0c * 8), 0, 0x7FFE0000, 0x00020000);
#pragma unroll
            for (int kt = 0; kt < 4; kt++)
                wr[kt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, wvoff, kt * 2 * CoutPad3 * 8 * 4, 0));
        };
        fetch_w(0);
        for (int n0c = 0; n0c < CoutPad3; n0c += 64) {
            if (n0c > 0) __syncthreads();                       // the previous chunk's fragment reads are done
#pragma unroll
            for (int kt = 0; kt < 4; kt++) *reinterpret_cast<f32x4*>(Wl + kt * 1024 + ww_off) = wr[kt];
            __syncthreads();                                    // T (first chunk) and this chunk's weights are in LDS
            if (n0c + 64 < CoutPad3) fetch_w(n0c + 64);
            // the chunk's residual values are requested BEFORE its 64 MFMAs (their HBM latency hides under ~4 000 matrix cycles)
            const int n3 = n0c + wn * 32 + l31;
            int vo3[2];
            float ex[2][16];
#pragma unroll
            for (int i = 0; i < 2; i++) vo3[i] = ((m0 + wm * 64 + i * 32 + 4 * kh_lane) * ld3 + n3) * 4;
            const bool late = (b.exp_flags & 2) != 0;           // tuning experiment (CALD_P4_FUSE_LATE=1): load them after the MFMAs instead
            if (!late) {
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        ex[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX3, vo3[i], ((r & 3) + 8 * (r >> 2)) * ld3 * 4, 0));
            }
            __builtin_amdgcn_sched_barrier(0);                  // keep the scheduler from sinking them to their first use
            f32x16 acc2[2][1];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc2[i][0][r] = 0.0f;
#pragma unroll
            for (int kt = 0; kt < 4; kt++) {
                f32x4 ga0[2], ga1[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    ga0[t] = *reinterpret_cast<const f32x4*>(Tl + kt * TILE_A + foa2[t]);
                    ga1[t] = *reinterpret_cast<const f32x4*>(Tl + kt * TILE_A + foa2[t] + BM * 8);
                }
                const f32x4 gb0 = *reinterpret_cast<const f32x4*>(Wl + kt * 1024 + fob2);
                const f32x4 gb1 = *reinterpret_cast<const f32x4*>(Wl + kt * 1024 + fob2 + 64 * 8);
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int i = 0; i < 2; i++) acc2[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga0[i][q], gb0[q], acc2[i][0], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int i = 0; i < 2; i++) acc2[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga1[i][q], gb1[q], acc2[i][0], 0, 0, 0);
            }
            // conv3's epilogue, BN -> + residual -> ReLU as in p4_epilogue<1>: the buffer resources end at the view's last valid row, so
            // the rows of a partial tile need no per-element test (loads beyond the end return 0, stores are dropped)
            {
                const float sc3 = b.scale[n3], sh3 = b.shift[n3];
                if (late) {
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int r = 0; r < 16; r++)
                            ex[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX3, vo3[i], ((r & 3) + 8 * (r >> 2)) * ld3 * 4, 0));
                }
#pragma unroll
                for (int i = 0; i < 2; i++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        float val = acc2[i][0][r] * sc3;
                        val = val + sh3;
                        val = val + ex[i][r];
                        if (relu3) val = val > 0.0f ? val : 0.0f;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsO3, vo3[i], ((r & 3) + 8 * (r >> 2)) * ld3 * 4, 0);
                    }
                }
            }
        }
    } else {
        p4_epilogue<EPI, MASK, TM, TN>(a, acc, smem, v, so, m0, Mv, n0);
    }
    if (trace && tid == 0) {
        trace[3] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        trace[4] = __builtin_amdgcn_s_memtime();
    }
}

template <int EPI, bool C4, int TN, int TAPS>
__global__ __launch_bounds__(256, 3) void conv_p4_kernel(const ConvArgs a) { conv_p4_body<EPI, C4, TN, TAPS>(a, blockIdx.x); }
template <int EPI, bool C4, int TN, int TAPS>
__global__ __launch_bounds__(256, 3) void conv_p4_group_kernel(const ConvGroup g) {
    int i = 0;
    while (i + 1 < g.n && g.blk0[i + 1] <= (int)blockIdx.x) i++;
    conv_p4_body<EPI, C4, TN, TAPS>(g.p[i], (int)blockIdx.x - g.blk0[i]);
}
// bottleneck conv2 + conv3 in one launch (the two problems of a ConvGroup: p[0] = conv2, p[1] = conv3); 48 KB of LDS -> 3 workgroups / CU
__global__ __launch_bounds__(256, 3) void conv_p4_fused_kernel(const ConvGroup g) { conv_p4_body<8, false, 1, 9>(g.p[0], blockIdx.x, &g.p[1]); }

// CALD_P4_EXP: kernel-tuning experiments (tools/bench_conv.py): bit 0 = drop the output stores
static inline int p4_exp_env() { static const int e = getenv("CALD_P4_EXP") ? atoi(getenv("CALD_P4_EXP")) : 0; return e; }

// filter shape -> k-loop variant
static inline int p4_taps(const ConvArgs& a) {
    static const int unroll_env = getenv("CALD_P4_UNROLL") ? atoi(getenv("CALD_P4_UNROLL")) : 1;      // 0: generic rolled loop everywhere
    if (!unroll_env || a.Cin % 16 != 0) return 0;
    if (a.KH == 3 && a.KW == 3) return 9;
    if (a.KH == 1 && a.KW == 1) return 1;
    return 0;
}

// M-tile slots of the launch grid: the XCD-contiguous map of spatial filters rounds the tile count up to a multiple of 8
static inline int p4_grid_mtiles(const ConvArgs& a) { return a.KH * a.KW > 1 ? 8 * ((a.total_mtiles + 7) / 8) : a.total_mtiles; }

// grouped launch (EPI 0 only): returns true if every problem qualifies for the same variant
bool launch_conv_p4_group(const ConvArgs* p, int n, hipStream_t stream) {
    if (n < 1 || n > CALD_MAX_GROUP) return false;
    const bool wide = p[0].CoutPad % 128 == 0;
    ConvGroup g; g.n = n; int blk = 0;
    for (int i = 0; i < n; i++) {
        const ConvArgs& a = p[i];
        if (!a.w4 || a.w16 || a.CoutPad % 64 != 0 || (a.CoutPad % 128 == 0) != wide || a.Cin % 16 != 0 || a.KH * a.KW > 32 || a.residual || a.up) return false;
        if ((a.mask != nullptr) != (p[0].mask != nullptr)) return false;
        g.blk0[i] = blk; blk += p4_grid_mtiles(a) * (a.CoutPad / (wide ? 128 : 64)); g.p[i] = a; g.p[i].exp_flags = 0;
    }
    g.blk0[n] = blk;
    const int taps = p4_taps(p[0]);
    for (int i = 1; i < n; i++) if (p4_taps(p[i]) != taps) return false;
    const dim3 grid((unsigned)blk), block(256);
#define P4_GROUP(EV, TNV, TAPSV) hipLaunchKernelGGL((conv_p4_group_kernel<EV, false, TNV, TAPSV>), grid, block, 0, stream, g)
#define P4_GROUP_T(EV) { if (wide) { if (taps == 9) P4_GROUP(EV, 2, 9); else if (taps == 1) P4_GROUP(EV, 2, 1); else P4_GROUP(EV, 2, 0); } \
                         else { if (taps == 9) P4_GROUP(EV, 1, 9); else if (taps == 1) P4_GROUP(EV, 1, 1); else P4_GROUP(EV, 1, 0); } }
    if (p[0].mask) P4_GROUP_T(4) else P4_GROUP_T(0)
#undef P4_GROUP_T
#undef P4_GROUP
    return true;
}

// conv2 (3 x 3, stride 1, 64 -> 64, BN + ReLU) followed by conv3 (1 x 1, 64 -> 128 k channels, BN + residual + ReLU) on the same pixels:
// returns true if the fused kernel took both (same bits as the two separate launches)
bool launch_conv_p4_fused(const ConvArgs& c2, const ConvArgs& c3, hipStream_t stream) {
    static const int on = getenv("CALD_P4_FUSE") ? atoi(getenv("CALD_P4_FUSE")) : 1;
    if (!on || !c2.w4 || !c3.w4 || c2.w16 || c3.w16) return false;
    if (c2.KH != 3 || c2.KW != 3 || c2.stride != 1 || c2.pad != 1 || c2.Cin % 16 || c2.Cout != 64 || c2.CoutPad != 64 || c2.out_ld != 64) return false;
    if (c2.residual || c2.up || c2.mask || c2.dyn_rows || c2.in_relu || p4_taps(c2) != 9) return false;
    if (c3.KH != 1 || c3.KW != 1 || c3.stride != 1 || c3.pad != 0 || c3.Cin != 64 || c3.Kpad != 64 || c3.CoutPad % 64 || c3.Cout != c3.CoutPad) return false;
    if (!c3.residual || !c3.scale || c3.bias || c3.out_ld != c3.Cout || c3.up || c3.mask || c3.dyn_rows || c3.in_relu || c3.in != c2.out || c3.total_mtiles != c2.total_mtiles || c3.V != c2.V) return false;
    ConvGroup g; g.n = 2; g.blk0[0] = 0; g.p[0] = c2; g.p[1] = c3; g.p[0].exp_flags = 0;
    static const int late_env = getenv("CALD_P4_FUSE_LATE") ? atoi(getenv("CALD_P4_FUSE_LATE")) : 0;
    g.p[1].exp_flags = late_env ? 2 : 0;
    const unsigned grid = (unsigned)p4_grid_mtiles(c2);          // one workgroup per 128-row tile (conv2's only N tile)
    hipLaunchKernelGGL(conv_p4_fused_kernel, dim3(grid), dim3(256), 0, stream, g);
    return true;
}

// returns true if this variant handled the launch
bool launch_conv_p4(const ConvArgs& a_in, hipStream_t stream) {
    if (!a_in.w4 || a_in.CoutPad % 64 != 0) return false;
    const int exp_env = p4_exp_env();
    static const int pad_lds = getenv("CALD_P4_PADLDS") ? atoi(getenv("CALD_P4_PADLDS")) : 0;    // extra dynamic LDS: caps workgroups per CU
    ConvArgs a = a_in; a.exp_flags = exp_env;
    bool wide = a.CoutPad % 128 == 0;
    if (wide) {
        // tail quantisation: a launch of B equal workgroups on 768 slots (256 CUs x 3) runs at B / (ceil(B / 768) * 768); the
        // 128 x 64 tile doubles B at ~0.88 of the 128 x 128 tile's per-workgroup efficiency -- use it where that wins
        static const int narrow_env = getenv("CALD_CONV_NARROW") ? atoi(getenv("CALD_CONV_NARROW")) : 1;
        const long long b2 = (long long)a.total_mtiles * (a.CoutPad / 128), b1 = 2 * b2;
        const double e2 = (double)b2 / (double)(((b2 + 767) / 768) * 768), e1 = 0.88 * (double)b1 / (double)(((b1 + 767) / 768) * 768);
        if (narrow_env && e1 > e2 && !a.dyn_rows) wide = false;
    }
    dim3 grid((unsigned)(p4_grid_mtiles(a) * (a.CoutPad / (wide ? 128 : 64)))), block(256);
    if (a.Cin == 4) {
        if (a.residual || a.up || a.in_relu) return false;
        static const int stem_env = getenv("CALD_P4_UNROLL") ? atoi(getenv("CALD_P4_UNROLL")) : 1;
        if (stem_env && a.KH == 7 && a.KW == 7 && a.Kpad == 208) {      // unrolled stem: per-k-tile tap offsets precomputed
            if (wide) hipLaunchKernelGGL((conv_p4_kernel<0, true, 2, 13>), grid, block, pad_lds, stream, a);
            else hipLaunchKernelGGL((conv_p4_kernel<0, true, 1, 13>), grid, block, pad_lds, stream, a);
            return true;
        }
        if (wide) hipLaunchKernelGGL((conv_p4_kernel<0, true, 2, 0>), grid, block, pad_lds, stream, a);
        else hipLaunchKernelGGL((conv_p4_kernel<0, true, 1, 0>), grid, block, pad_lds, stream, a);
        return true;
    }
    if (a.Cin % 16 != 0 || a.KH * a.KW > 32) return false;
    const int taps = p4_taps(a);
#define P4_ONE(EPIV, TNV, TAPSV) hipLaunchKernelGGL((conv_p4_kernel<EPIV, false, TNV, TAPSV>), grid, block, pad_lds, stream, a)
#define P4_TAPS(EPIV, TNV) { if (taps == 9) P4_ONE(EPIV, TNV, 9); else if (taps == 1) P4_ONE(EPIV, TNV, 1); else P4_ONE(EPIV, TNV, 0); }
    if (a.mask) {                      // training backward (never set by the inference engine)
        if (a.up) return false;
        if (wide) { if (a.residual) P4_TAPS(5, 2) else P4_TAPS(4, 2) } else { if (a.residual) P4_TAPS(5, 1) else P4_TAPS(4, 1) }
    } else if (wide) {
        if (a.residual) P4_TAPS(1, 2) else if (a.up) P4_TAPS(2, 2) else P4_TAPS(0, 2)
    } else {
        if (a.residual) P4_TAPS(1, 1) else if (a.up) P4_TAPS(2, 1) else P4_TAPS(0, 1)
    }
#undef P4_TAPS
#undef P4_ONE
    return true;
}
