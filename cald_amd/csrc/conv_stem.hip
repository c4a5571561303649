// conv_stem.hip -- the ResNet stem (7 x 7, stride 2, pad 3, 3 -> 64 channels, FrozenBN + ReLU; torchvision resnet conv1 / bn1 /
// relu under detection/frcnn_la.py:283 resnet_fpn_backbone) as an LDS-resident implicit GEMM on v_mfma_f32_32x32x2_f32.
//
// Same arithmetic as the generic kernels (conv_p4.hip <.., C4 = true, .., 13>) and bit-identical results: one k-ordered fma chain
// per output in (kh, kw, channel) order starting from +0.  The generic path walks the 4-channel NHWC4 input, i.e. 7 * 7 * 4 = 196
// chain slots padded to 208 (13 k-tiles), a quarter of them multiplying the zero fourth channel.  Here the chain has
// 7 * 22 = 154 slots: per filter row the 21 real (kw, c) taps plus ONE zero-weight slot that makes the row even (an fma with a zero
// weight leaves the accumulator unchanged bit for bit) -> 77 MFMA k-pairs instead of 104.
//
//   * A workgroup owns an 8 x 16 block of output pixels (the padded sizes are multiples of 32, so every view splits exactly).  Its
//     21 x 37 input patch is staged ONCE in LDS, three channels packed (12 bytes per pixel): for an output pixel the 21 taps of a
//     filter row are then 21 CONSECUTIVE floats, and the A fragment of k-pair jj is one ds_read_b32 at base + immediate
//     (base = patch + (row(2 ty) + 2 tx * 3 + h) * 4: no address arithmetic in the loop, no tap validity logic at all -- pixels
//     outside the image are zeros in the patch).  Patch rows are laid out in pairs of 225 floats (STEM_PAIR): the two tile rows a
//     wave's 32 lanes read are then one bank apart and the stride-6 column reads are conflict-free.
//   * The weights [20 quads][2 h][64 n][4] (40 KB) are LDS-resident for the whole life of the workgroup, which walks several
//     tiles (persistent grid of 3 workgroups per CU); a B fragment read is one ds_read_b128 per four k-pairs.
//   * No k-loop over global memory: per tile one patch load (9.3 KB), 2 x 77 MFMAs per wave, the epilogue.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define STEM_TH 8
#define STEM_TW 16
#define STEM_PR (2 * STEM_TH + 5)      // 21 patch rows
#define STEM_PC (2 * STEM_TW + 5)      // 37 patch columns
#define STEM_PITCH 112                 // floats per patch row: 37 * 3 = 111 + the slot the zero-weight tap of the last pixel reads
#define STEM_PAIR 225                  // floats per PAIR of patch rows (2 * 112 + 1): the two tile rows a wave's 32 lanes cover are two patch
                                       // rows apart = 225 floats = 1 (mod 32 banks), so their stride-6 column reads use the odd banks
#define STEM_ROW(r) (((r) >> 1) * STEM_PAIR + ((r) & 1) * STEM_PITCH)
#define STEM_WFLOATS (20 * 2 * 64 * 4)

__global__ __launch_bounds__(256, 3) void conv_stem_kernel(const ConvArgs a, const float* __restrict__ wstem, const int tiles_per_xcd) {
    __shared__ __attribute__((aligned(16))) float s_w[STEM_WFLOATS];
    __shared__ __attribute__((aligned(16))) float s_patch[((STEM_PR + 1) / 2) * STEM_PAIR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;

    // weights -> LDS (once per workgroup)
#pragma unroll
    for (int i = 0; i < STEM_WFLOATS / 4 / 256; i++)
        reinterpret_cast<f32x4*>(s_w)[tid + 256 * i] = reinterpret_cast<const f32x4*>(wstem)[tid + 256 * i];
    for (int i = tid; i < ((STEM_PR + 1) / 2) * STEM_PAIR; i += 256) s_patch[i] = 0.0f;   // incl. float 111 of every row, never written again

    // patch pixels this thread stages: p = tid + 256 i  ->  (row, column)
    int p_row[4], p_col[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = tid + 256 * i;
        p_row[i] = p / STEM_PC; p_col[i] = p - p_row[i] * STEM_PC;
        if (p >= STEM_PR * STEM_PC) p_row[i] = -1;
    }
    // fragment addresses
    const int m_l = wave * 32 + l31, ty = m_l >> 4, tx = m_l & 15;
    const float* const fa = s_patch + ty * STEM_PAIR + 6 * tx + h;            // patch row 2 ty
    const float* const fb0 = s_w + (h * 64 + l31) * 4;
    const float* const fb1 = fb0 + 32 * 4;
    const float sc0 = a.scale[l31], sh0 = a.shift[l31], sc1 = a.scale[32 + l31], sh1 = a.shift[32 + l31];
    const bool relu = a.relu != 0;

    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int t_end = min(a.total_mtiles, (xcd + 1) * tiles_per_xcd);
    int v = 0;
    // the patch of tile t + 1 is fetched into registers before the MFMAs of tile t (global latency under the matrix work)
    f32x4 px[4];
    int vn = 0;
    auto fetch = [&](int tile) {
        while (vn + 1 < a.V && a.seg_out[vn + 1].tile_start <= tile) vn++;
        const LevelSeg so = a.seg_out[vn];
        const LevelSeg si = a.seg_in[vn];
        const int tcols = so.W / STEM_TW, t_in = tile - so.tile_start;
        const int tr = t_in / tcols, tc = t_in - tr * tcols;
        const int iy0 = 2 * tr * STEM_TH - 3, ix0 = 2 * tc * STEM_TW - 3;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + si.pix_off * 4), 0, 0x7FFE0000, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int iy = iy0 + p_row[i], ix = ix0 + p_col[i];
            const bool ok = p_row[i] >= 0 && (unsigned)iy < (unsigned)si.H && (unsigned)ix < (unsigned)si.W;
            px[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? (iy * si.W + ix) * 16 : 0x7FFF0000, 0, 0));
        }
    };
    const int t_first = xcd * tiles_per_xcd + idx;
    if (t_first < t_end) fetch(t_first);
    for (int tile = t_first; tile < t_end; tile += per_xcd) {
        while (v + 1 < a.V && a.seg_out[v + 1].tile_start <= tile) v++;
        const LevelSeg so = a.seg_out[v];
        const int Wo = so.W;
        const int tcols = Wo / STEM_TW;
        const int t_in = tile - so.tile_start;
        const int tr = t_in / tcols, tc = t_in - tr * tcols;
        const int oy0 = tr * STEM_TH, ox0 = tc * STEM_TW;
        __syncthreads();                       // the previous tile's fragment reads are done (first pass: weights / zero fill are in)
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (p_row[i] >= 0) {
                float* d = s_patch + STEM_ROW(p_row[i]) + 3 * p_col[i];
                d[0] = px[i][0]; d[1] = px[i][1]; d[2] = px[i][2];
            }
        __syncthreads();
        if (tile + per_xcd < t_end) fetch(tile + per_xcd);

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        // 77 k-pairs: k-pair j = 11 kh + jj  (slots 2 jj, 2 jj + 1 of filter row kh); weights in quads of four k-pairs
#pragma unroll
        for (int q = 0; q < 20; q++) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(fb0 + q * 512);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(fb1 + q * 512);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int j = 4 * q + e;
                if (j < 77) {
                    const int kh = j / 11, jj = j - kh * 11;
                    const float av = fa[STEM_ROW(kh) + 2 * jj];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1[e], acc1, 0, 0, 0);
                }
            }
        }

        // epilogue: FrozenBN (x * scale, + shift: two roundings) -> ReLU; accumulator register r of lane (l31, h) is tile-local row
        // 32 wave + 4 h + (r & 3) + 8 (r >> 2), column l31 (+ 32)
        float* const out_v = a.out + so.pix_off * 64;
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)out_v, 0, 0x7FFE0000, 0x00020000);
        const int vo = (((oy0 + 2 * wave) * Wo + ox0 + 4 * h) * 64 + l31) * 4;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rr = r >> 2;                                  // rows +0, +8 | +16, +24  ->  tile row +0 | +1, column +0 | +8
            const int soff = (((rr >> 1) * Wo + 8 * (rr & 1) + (r & 3)) * 64) * 4;
            float v0 = acc0[r] * sc0; v0 = v0 + sh0;
            float v1 = acc1[r] * sc1; v1 = v1 + sh1;
            if (relu) { v0 = v0 > 0.0f ? v0 : 0.0f; v1 = v1 > 0.0f ? v1 : 0.0f; }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rsO, vo, soff, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rsO, vo + 128, soff, 0);
        }
    }
}

// returns true if this kernel handled the launch (7 x 7 / 2 stem with packed stem weights, every view an exact grid of 8 x 16 blocks)
bool launch_conv_stem(const ConvArgs& a, hipStream_t stream) {
    if (!a.wstem || a.KH != 7 || a.KW != 7 || a.stride != 2 || a.pad != 3 || a.Cin != 4 || a.Cout != 64 || a.out_ld != 64) return false;
    if (a.residual || a.up || a.in_relu || a.bias || !a.scale || a.mask || a.dyn_rows) return false;
    int grid = 768;
    if (a.total_mtiles < grid) grid = (a.total_mtiles + 7) / 8 * 8;
    const int per_xcd = (a.total_mtiles + 7) / 8;
    hipLaunchKernelGGL(conv_stem_kernel, dim3((unsigned)grid), dim3(256), 0, stream, a, a.wstem, per_xcd);
    return true;
}
