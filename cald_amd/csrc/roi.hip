// roi.hip -- box head front/back ends of the Faster R-CNN forward:
//   * MultiScaleRoIAlign (7x7, sampling_ratio 2, aligned=False) over P2..P5, written directly in the
//     A-operand layout of the fc6 GEMM: [roi][bin(49)][channel]  (reference detection/frcnn_la.py:112,
//     :205-209; torchvision 0.8.2 roi_align semantics)
//   * RoIHeads.postprocess_detections (detection/frcnn_la.py:32-87) + transform.postprocess
//     (:292-315): softmax, per-class decode, clip, score threshold, class-batched NMS, top-100,
//     gather of scores_cls / prob_max / props, rescale to the original image.
#include "common.h"
#include "kernels.h"
#include "h16.h"
#include "sortnms.h"

// Processing order of a view's RoIs.  Proposals arrive in score order, i.e. scattered over the image and the pyramid: with one
// workgroup per RoI in that order every XCD's L2 (4 MB) kept re-fetching the whole 40 MB pyramid of the view (8.3 GB raw L2->fabric
// fetch per 64-view forward for 2.6 GB of features).  One workgroup per view sorts (level, 16-pixel row band, column) keys in LDS;
// roi_align_kernel then walks a view's RoIs in that order ON ONE XCD, so RoIs in flight together overlap in the feature map.
// Results do not depend on the order (every RoI is computed alone and written to its own slot r).
__global__ __launch_bounds__(1024) void roi_order_kernel(RoiArgs a) {
    __shared__ unsigned long long keys[1024];
    const int v = blockIdx.x, tid = threadIdx.x;
    const int n = a.prop_count[v];
    unsigned long long key = 0ull;
    if (tid < n && tid < CALD_ROI_CAP) {
        const float4 box = reinterpret_cast<const float4*>(a.proposals)[(long long)v * CALD_ROI_CAP + tid];
        const int l = roi_level(box);
        const float scale = 1.0f / (float)(4 << l);
        const float cy = (box.y + box.w) * 0.5f * scale, cx = (box.x + box.z) * 0.5f * scale;
        int yb = (int)(cy * (1.0f / 16.0f)), xb = (int)cx;
        yb = yb < 0 ? 0 : (yb > 1023 ? 1023 : yb); xb = xb < 0 ? 0 : (xb > 4095 ? 4095 : xb);
        // descending sort: complement so that level 0 / top rows come first; + 1 keeps valid keys above the padding zeros
        const unsigned pos = ((unsigned)l << 22) | ((unsigned)yb << 12) | (unsigned)xb;
        key = ((unsigned long long)(0x0FFFFFFFu - pos + 1u) << 32) | (unsigned long long)tid;
    }
    keys[tid] = key;
    __syncthreads();
    block_bitonic_sort_desc(keys, 1024);
    a.order[v * 1024 + tid] = (int)(keys[tid] & 0xFFFFFFFFull);
}

// One workgroup per RoI, block = 256; workgroup b runs on XCD b % 8 and XCD x owns views x, x + 8, ... (all RoIs of a view on one
// L2, in roi_order_kernel's order).  The 14 sample rows and 14 sample columns of the RoI (7 bins x 2 samples,
// separable) are set up once per workgroup in LDS; then a thread owns (bin, 4 consecutive channels): 16 float4
// gathers in flight per bin and one float4 store, 1 KB contiguous per wavefront.  Arithmetic order per channel is
// the oracle's: acc += ((w1*v1 + w2*v2) + w3*v3) + w4*v4 over samples (iy, ix), then / 4.
__global__ __launch_bounds__(256) void roi_align_kernel(RoiArgs a) {
    __shared__ RoiSample sy[14], sx[14];
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3, tid = threadIdx.x;
    const int v = xcd + 8 * (seq / CALD_ROI_CAP), slot = seq % CALD_ROI_CAP;
    if (v >= a.V || slot >= a.prop_count[v]) return;
    const int r = a.order[v * 1024 + slot];
    const float4 box = reinterpret_cast<const float4*>(a.proposals)[(long long)v * CALD_ROI_CAP + r];
    const int l = roi_level(box);
    const LevelSeg sg = a.seg[l][v];
    const int Hf = sg.H, Wf = sg.W, C = a.C, Cq = C >> 2;
    const float4* f = reinterpret_cast<const float4*>(a.feat[l] + sg.pix_off * (long long)C);
    if (tid < 28) {
        const float scale = 1.0f / (float)(4 << l);
        const float x1 = box.x * scale, y1 = box.y * scale, x2 = box.z * scale, y2 = box.w * scale;
        float rw = x2 - x1; if (!(rw >= 1.0f)) rw = 1.0f;
        float rh = y2 - y1; if (!(rh >= 1.0f)) rh = 1.0f;
        const float bw = rw / 7.0f, bh = rh / 7.0f;
        if (tid < 14) sy[tid] = roi_sample(y1, bh, tid >> 1, tid & 1, Hf);
        else sx[tid - 14] = roi_sample(x1, bw, (tid - 14) >> 1, (tid - 14) & 1, Wf);
    }
    __syncthreads();
    float4* out = reinterpret_cast<float4*>(a.out + ((long long)v * CALD_ROI_CAP + r) * 49 * C);
    for (int idx = tid; idx < 49 * Cq; idx += 256) {
        const int bin = idx / Cq, q = idx - bin * Cq;
        const int ph = bin / 7, pw = bin - ph * 7;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int iy = 0; iy < 2; iy++) {
            const RoiSample Y = sy[ph * 2 + iy];
#pragma unroll
            for (int ix = 0; ix < 2; ix++) {
                const RoiSample X = sx[pw * 2 + ix];
                const bool ok = Y.valid && X.valid;
                const int yl = ok ? Y.lo : 0, yh = ok ? Y.hi : 0, xl = ok ? X.lo : 0, xh = ok ? X.hi : 0;
                const float w1 = ok ? Y.h * X.h : 0.0f, w2 = ok ? Y.h * X.l : 0.0f;
                const float w3 = ok ? Y.l * X.h : 0.0f, w4 = ok ? Y.l * X.l : 0.0f;
                const float4 v1 = f[(long long)(yl * Wf + xl) * Cq + q], v2 = f[(long long)(yl * Wf + xh) * Cq + q];
                const float4 v3 = f[(long long)(yh * Wf + xl) * Cq + q], v4 = f[(long long)(yh * Wf + xh) * Cq + q];
                acc.x = acc.x + (((w1 * v1.x + w2 * v2.x) + w3 * v3.x) + w4 * v4.x);
                acc.y = acc.y + (((w1 * v1.y + w2 * v2.y) + w3 * v3.y) + w4 * v4.y);
                acc.z = acc.z + (((w1 * v1.z + w2 * v2.z) + w3 * v3.z) + w4 * v4.z);
                acc.w = acc.w + (((w1 * v1.w + w2 * v2.w) + w3 * v3.w) + w4 * v4.w);
            }
        }
        const float4 res = make_float4(acc.x / 4.0f, acc.y / 4.0f, acc.z / 4.0f, acc.w / 4.0f);
        if (a.out16) h16_store4(reinterpret_cast<unsigned char*>(out + (long long)bin * Cq), 4 * q, res);
        else out[idx] = res;
    }
}

// C = 256 (every FPN box head here): one workgroup per RoI, SEVEN waves -- wave = bin row ph, lane = channel quad.  The kernel above
// is bound by the texture-address path: 16 float4 gathers per (bin, quad), 784 KB of L1 requests for a 50 KB RoI, although a
// 7 x 7 / 2 x 2 sample grid over a ~14-pixel RoI touches each feature pixel several times.  Here every sample index of a wave is
// wave-uniform (readfirstlane -> scalar branches): the wave walks its 14 sample columns left to right keeping the pixel columns
// (lo, hi) of the previous sample for its (up to) four pixel rows in registers, and loads a column only when the sample moves on
// to a new one (hi of one sample is usually lo of the next; the two sample rows of a bin usually share a pixel row).  Loads per
// bin row: (unique rows <= 4) x (unique columns ~ RoI width + 2) instead of 7 x 16.  The arithmetic per output is the kernel
// above's, term for term; an out-of-range sample has zero weights in both kernels (which finite pixel it multiplies is immaterial:
// acc starts at +0 and x + (+-0) = x).
__device__ __forceinline__ float4 roi_bilinear(const float4 acc, const float w1, const float w2, const float w3, const float w4,
                                               const float4 v1, const float4 v2, const float4 v3, const float4 v4) {
    float4 r;
    r.x = acc.x + (((w1 * v1.x + w2 * v2.x) + w3 * v3.x) + w4 * v4.x);
    r.y = acc.y + (((w1 * v1.y + w2 * v2.y) + w3 * v3.y) + w4 * v4.y);
    r.z = acc.z + (((w1 * v1.z + w2 * v2.z) + w3 * v3.z) + w4 * v4.z);
    r.w = acc.w + (((w1 * v1.w + w2 * v2.w) + w3 * v3.w) + w4 * v4.w);
    return r;
}
// NR = number of distinct pixel rows the two sample rows of this bin row touch, PAT the way they share them:
//   PAT 0: (Y0.lo, Y0.hi, Y1.lo, Y1.hi) loaded as four rows (no sharing assumed)
//   PAT 1: Y1.lo == Y0.hi            -> rows (Y0.lo, Y0.hi, Y1.hi), sample row 1 uses rows (1, 2)
//   PAT 2: Y1 on the same pixel rows -> rows (Y0.lo, Y0.hi), both sample rows use (0, 1)
template <int PAT>
__device__ __forceinline__ void roi_rows_walk(const float4* const f, const int Wf, const RoiSample* const sx, const float4* const wt, const RoiSample Y0,
                                              const RoiSample Y1, const int ph, const int q, float4* const out, const bool out16) {
    constexpr int NR = PAT == 0 ? 4 : (PAT == 1 ? 3 : 2);
    const int r0 = __builtin_amdgcn_readfirstlane(Y0.lo), r1 = __builtin_amdgcn_readfirstlane(Y0.hi);
    const int r2 = __builtin_amdgcn_readfirstlane(Y1.lo), r3 = __builtin_amdgcn_readfirstlane(Y1.hi);
    const float4* const p0 = f + (long long)r0 * Wf * 64 + q;
    const float4* const p1 = f + (long long)r1 * Wf * 64 + q;
    const float4* const p2 = f + (long long)(PAT == 0 ? r2 : r3) * Wf * 64 + q;
    const float4* const p3 = f + (long long)r3 * Wf * 64 + q;
    // four pixel-column slots per bin: (L0, H0) = sample 2 pw, (L1, H1) = sample 2 pw + 1, each NR rows.  Every load a bin needs is
    // issued before its arithmetic starts (up to 4 NR float4 in flight per lane; a walk that loaded sample by sample was bound by
    // the latency of its 14 dependent steps).
    // (the empty asm keeps each copy inside its wave-uniform branch: the compiler otherwise flattens the branches into per-lane selects,
    // 45 v_cndmask per bin against 76 packed multiplies / adds of bilinear arithmetic)
#define ROI_COPY(D, S) { asm volatile(""); D##0 = S##0; D##1 = S##1; if (NR > 2) D##2 = S##2; if (NR > 3) D##3 = S##3; }
#define ROI_LOAD(D, X) { D##0 = p0[(X) * 64]; D##1 = p1[(X) * 64]; if (NR > 2) D##2 = p2[(X) * 64]; if (NR > 3) D##3 = p3[(X) * 64]; }
    // the four bilinear weight products of every (sample row, sample column) pair are set up once per workgroup in LDS (wt): a
    // wave-uniform ds_read_b128 per sample instead of the products and validity selects on every lane
#define ROI_SAMPLE(SX, A, B, T0, T1)                                                                                             \
    {                                                                                                                            \
        const float4 u = wt[(ph * 2) * 14 + (SX)], w = wt[(ph * 2 + 1) * 14 + (SX)];                                             \
        T0 = roi_bilinear(zero, u.x, u.y, u.z, u.w, A##0, B##0, A##1, B##1);                                                     \
        if (PAT == 0) T1 = roi_bilinear(zero, w.x, w.y, w.z, w.w, A##2, B##2, A##3, B##3);                                       \
        else if (PAT == 1) T1 = roi_bilinear(zero, w.x, w.y, w.z, w.w, A##1, B##1, A##2, B##2);                                  \
        else T1 = roi_bilinear(zero, w.x, w.y, w.z, w.w, A##0, B##0, A##1, B##1);                                                \
    }
    float4 L00, L01, L02, L03, H00, H01, H02, H03, L10, L11, L12, L13, H10, H11, H12, H13;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    L00 = L01 = L02 = L03 = H00 = H01 = H02 = H03 = L10 = L11 = L12 = L13 = H10 = H11 = H12 = H13 = zero;
    int plo = -1, phi = -1;              // columns held by (L1, H1) = the previous bin's second sample
#pragma unroll
    for (int pw = 0; pw < 7; pw++) {
        const RoiSample X0 = sx[pw * 2], X1 = sx[pw * 2 + 1];
        const int c0l = __builtin_amdgcn_readfirstlane(X0.lo), c0h = __builtin_amdgcn_readfirstlane(X0.hi);
        const int c1l = __builtin_amdgcn_readfirstlane(X1.lo), c1h = __builtin_amdgcn_readfirstlane(X1.hi);
        // 1. what the previous bin already holds
        const bool h0_alias = c0h == c0l;
        const bool l0_prev = c0l == plo || c0l == phi, h0_prev = !h0_alias && (c0h == phi || c0h == plo);
        if (l0_prev) { if (c0l == plo) ROI_COPY(L0, L1) else ROI_COPY(L0, H1) }
        if (h0_prev) { if (c0h == phi) ROI_COPY(H0, H1) else ROI_COPY(H0, L1) }
        // 2. every column that has to come from memory
        const bool l1_alias = c1l == c0l || c1l == c0h;
        const bool h1_alias = c1h == c1l || c1h == c0h || c1h == c0l;
        if (!l0_prev) ROI_LOAD(L0, c0l)
        if (!h0_prev && !h0_alias) ROI_LOAD(H0, c0h)
        if (!l1_alias) ROI_LOAD(L1, c1l)
        if (!h1_alias) ROI_LOAD(H1, c1h)
        // 3. columns shared inside the bin
        if (h0_alias) ROI_COPY(H0, L0)
        if (l1_alias) { if (c1l == c0l) ROI_COPY(L1, L0) else ROI_COPY(L1, H0) }
        if (h1_alias) { if (c1h == c1l) ROI_COPY(H1, L1) else if (c1h == c0h) ROI_COPY(H1, H0) else ROI_COPY(H1, L0) }
        plo = c1l; phi = c1h;
        float4 s00, s01, s10, s11;
        ROI_SAMPLE(pw * 2, L0, H0, s00, s10)
        ROI_SAMPLE(pw * 2 + 1, L1, H1, s01, s11)
        // the kernel above adds the four samples in (iy, ix) order: (0,0), (0,1), (1,0), (1,1)
        float4 acc = s00;
        acc.x = acc.x + s01.x; acc.y = acc.y + s01.y; acc.z = acc.z + s01.z; acc.w = acc.w + s01.w;
        acc.x = acc.x + s10.x; acc.y = acc.y + s10.y; acc.z = acc.z + s10.z; acc.w = acc.w + s10.w;
        acc.x = acc.x + s11.x; acc.y = acc.y + s11.y; acc.z = acc.z + s11.z; acc.w = acc.w + s11.w;
        const float4 res = make_float4(acc.x / 4.0f, acc.y / 4.0f, acc.z / 4.0f, acc.w / 4.0f);
        const int idx = (ph * 7 + pw) * 64 + q;
        if (out16) h16_store4(reinterpret_cast<unsigned char*>(out + (ph * 7 + pw) * 64), 4 * q, res);
        else out[idx] = res;
    }
#undef ROI_COPY
#undef ROI_LOAD
#undef ROI_SAMPLE
}
__global__ __launch_bounds__(448, 4) void roi_align_rows_kernel(RoiArgs a) {
    __shared__ RoiSample sy[14], sx[14];
    __shared__ float4 s_wt[196];             // [sample row][sample column] -> (w1, w2, w3, w4)
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3, tid = threadIdx.x;
    const int v = xcd + 8 * (seq / CALD_ROI_CAP), slot = seq % CALD_ROI_CAP;
    if (v >= a.V || slot >= a.prop_count[v]) return;
    const int r = a.order[v * 1024 + slot];
    const float4 box = reinterpret_cast<const float4*>(a.proposals)[(long long)v * CALD_ROI_CAP + r];
    const int l = roi_level(box);
    const LevelSeg sg = a.seg[l][v];
    const int Hf = sg.H, Wf = sg.W;
    const float4* f = reinterpret_cast<const float4*>(a.feat[l] + sg.pix_off * 256ll);
    if (tid < 28) {
        const float scale = 1.0f / (float)(4 << l);
        const float x1 = box.x * scale, y1 = box.y * scale, x2 = box.z * scale, y2 = box.w * scale;
        float rw = x2 - x1; if (!(rw >= 1.0f)) rw = 1.0f;
        float rh = y2 - y1; if (!(rh >= 1.0f)) rh = 1.0f;
        const float bw = rw / 7.0f, bh = rh / 7.0f;
        if (tid < 14) sy[tid] = roi_sample(y1, bh, tid >> 1, tid & 1, Hf);
        else sx[tid - 14] = roi_sample(x1, bw, (tid - 14) >> 1, (tid - 14) & 1, Wf);
    }
    __syncthreads();
    if (tid < 196) {
        const int iy = tid / 14, ix = tid - iy * 14;
        const RoiSample Y = sy[iy], X = sx[ix];
        const bool ok = Y.valid && X.valid;
        s_wt[tid] = make_float4(ok ? Y.h * X.h : 0.0f, ok ? Y.h * X.l : 0.0f, ok ? Y.l * X.h : 0.0f, ok ? Y.l * X.l : 0.0f);
    }
    __syncthreads();
    const int ph = __builtin_amdgcn_readfirstlane(tid >> 6), q = tid & 63;
    const RoiSample Y0 = sy[ph * 2], Y1 = sy[ph * 2 + 1];
    float4* out = reinterpret_cast<float4*>(a.out + ((long long)v * CALD_ROI_CAP + r) * 49 * 256);
    const int r0 = __builtin_amdgcn_readfirstlane(Y0.lo), r1 = __builtin_amdgcn_readfirstlane(Y0.hi);
    const int r2 = __builtin_amdgcn_readfirstlane(Y1.lo), r3 = __builtin_amdgcn_readfirstlane(Y1.hi);
    if (r2 == r0 && r3 == r1) roi_rows_walk<2>(f, Wf, sx, s_wt, Y0, Y1, ph, q, out, a.out16 != 0);
    else if (r2 == r1) roi_rows_walk<1>(f, Wf, sx, s_wt, Y0, Y1, ph, q, out, a.out16 != 0);
    else roi_rows_walk<0>(f, Wf, sx, s_wt, Y0, Y1, ph, q, out, a.out16 != 0);
}
void launch_roi_align(const RoiArgs& a, hipStream_t st) {
    static const bool rows = !(getenv("CALD_ROI_ROWS") && atoi(getenv("CALD_ROI_ROWS")) == 0);
    hipLaunchKernelGGL(roi_order_kernel, dim3(a.V), dim3(1024), 0, st, a);
    if (rows && a.C == 256) hipLaunchKernelGGL(roi_align_rows_kernel, dim3(8 * ((a.V + 7) / 8) * CALD_ROI_CAP), dim3(448), 0, st, a);
    else hipLaunchKernelGGL(roi_align_kernel, dim3(8 * ((a.V + 7) / 8) * CALD_ROI_CAP), dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------------
// Post-processing kernel 1: softmax per proposal; candidate (proposal, fg class) keys above the
// score threshold are appended to the per-view key list.  grid = (ceil(ROI_CAP/256), V)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void post_softmax_kernel(PostArgs a) {
    int* key_count = a.key_count;
    const int v = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.prop_count[v]) return;
    const int C = a.C;
    const float* lg = a.pred + ((long long)v * CALD_ROI_CAP + r) * a.pred_ld;
    float* pr = a.prob + ((long long)v * CALD_ROI_CAP + r) * C;
    float m = lg[0];
    for (int c = 1; c < C; c++) if (lg[c] > m) m = lg[c];
    float s = 0.0f;
    for (int c = 0; c < C; c++) { const float e = det_expf(lg[c] - m); pr[c] = e; s = s + e; }
    float pm = 0.0f;
    for (int c = 0; c < C; c++) {
        const float p = pr[c] / s;
        pr[c] = p;
        if (c == 1 || (c > 1 && p > pm)) pm = p;
        if (c >= 1 && p > a.score_thr) {
            const int slot = atomicAdd(&key_count[v], 1);
            if (slot < a.key_cap)
                a.keys[(long long)v * a.key_cap + slot] =
                    ((unsigned long long)det_orderable(p) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(r * (C - 1) + (c - 1)));
        }
    }
    a.pmax[(long long)v * CALD_ROI_CAP + r] = pm;
}

// ---------------------------------------------------------------------------------------------
// Post-processing kernel 2: per view: sort candidates, decode + clip their boxes, class-batched
// NMS (coordinate offset label*(max_coord+1) in fp32), keep the first det_cap, write the outputs.
// grid = V, block = 1024, dynamic LDS (sort buffer of up to 16384 keys; larger lists sort in place
// in global memory).
// ---------------------------------------------------------------------------------------------
#define POST_LDS_KEYS 8192
__global__ __launch_bounds__(1024) void post_nms_kernel(PostArgs a) {
    const int* key_count = a.key_count;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    unsigned long long* lkeys = reinterpret_cast<unsigned long long*>(dyn);
    const int cap = a.det.cap;
    float4* kept_box = reinterpret_cast<float4*>(dyn + (size_t)POST_LDS_KEYS * 8);
    float* kept_area = reinterpret_cast<float*>(kept_box + cap);
    int* keep_idx = reinterpret_cast<int*>(kept_area + cap);
    int* dead_or = keep_idx + cap;
    float* red = reinterpret_cast<float*>(dead_or + 256);
    __shared__ int s_nk;
    const int v = blockIdx.x, tid = threadIdx.x, C = a.C;
    int n = key_count[v];
    if (n > a.key_cap) n = a.key_cap;
    int NP = 1024;
    while (NP < n) NP <<= 1;
    unsigned long long* gkeys = a.keys + (long long)v * a.key_cap;
    unsigned long long* keys = (NP <= POST_LDS_KEYS) ? lkeys : gkeys;
    for (int i = tid; i < NP; i += 1024) keys[i] = (i < n) ? gkeys[i] : 0ull;
    __syncthreads();
    block_bitonic_sort_desc(keys, NP);
    // decode + clip in sorted order
    const ViewDesc vd = a.views[v];
    const float Wr = (float)vd.Wr, Hr = (float)vd.Hr;
    float4* cbox = reinterpret_cast<float4*>(a.cbox) + (long long)v * 2 * a.key_cap;
    const float4* props = reinterpret_cast<const float4*>(a.proposals) + (long long)v * CALD_ROI_CAP;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 1024) {
        const int pos = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
        const int r = pos / (C - 1), c = pos - r * (C - 1) + 1;
        const float4 p = props[r];
        const float pb[4] = {p.x, p.y, p.z, p.w};
        const float* dl = a.pred + ((long long)v * CALD_ROI_CAP + r) * a.pred_ld + C + 4 * c;
        const float d[4] = {dl[0], dl[1], dl[2], dl[3]};
        float o[4];
        det_box_decode(pb, d, 10.0f, 10.0f, 5.0f, 5.0f, o);
        float4 b;
        b.x = det_clamp(o[0], 0.0f, Wr); b.z = det_clamp(o[2], 0.0f, Wr);
        b.y = det_clamp(o[1], 0.0f, Hr); b.w = det_clamp(o[3], 0.0f, Hr);
        cbox[i] = b;
        mx = fmaxf(mx, fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
    }
    red[tid] = mx;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
    const float maxc = red[0];
    // offset boxes for class-batched NMS, stored after the raw ones
    float4* obox = cbox + n;   // cbox has 2*key_cap rows per view
    for (int i = tid; i < n; i += 1024) {
        const int pos = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
        const int c = pos % (C - 1) + 1;
        const float off = (float)c * (maxc + 1.0f);
        const float4 b = cbox[i];
        obox[i] = make_float4(b.x + off, b.y + off, b.z + off, b.w + off);
    }
    __syncthreads();
    block_nms_sorted(obox, n, a.nms_thr, cap, kept_box, kept_area, dead_or, keep_idx, &s_nk);
    const int nk = s_nk;
    const float rh = (float)((double)vd.Ho / (double)vd.Hr), rw = (float)((double)vd.Wo / (double)vd.Wr);
    for (int i = tid; i < nk; i += 1024) {
        const int k = keep_idx[i];
        const unsigned long long key = keys[k];
        const int pos = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        const int r = pos / (C - 1), c = pos - r * (C - 1) + 1;
        const float4 b = cbox[k];
        const float4 p = props[r];
        const long long o = (long long)v * cap + i;
        reinterpret_cast<float4*>(a.det.boxes)[o] = make_float4(b.x * rw, b.y * rh, b.z * rw, b.w * rh);
        reinterpret_cast<float4*>(a.det.props)[o] = make_float4(p.x * rw, p.y * rh, p.z * rw, p.w * rh);
        const float* pr = a.prob + ((long long)v * CALD_ROI_CAP + r) * C;
        a.det.scores[o] = pr[c];
        a.det.labels[o] = c;
        a.det.prob_max[o] = a.pmax[(long long)v * CALD_ROI_CAP + r];
        for (int q = 0; q < C; q++) a.det.scores_cls[o * C + q] = pr[q];
        if (a.kept_key) a.kept_key[o] = key;
    }
    if (tid == 0) { a.det.count[v] = nk; if (a.post_maxc) a.post_maxc[v] = maxc; }
}

void launch_frcnn_postprocess(const PostArgs& a, hipStream_t st) {
    hipMemsetAsync(a.key_count, 0, sizeof(int) * a.V, st);
    hipLaunchKernelGGL(post_softmax_kernel, dim3((CALD_ROI_CAP + 255) / 256, a.V), dim3(256), 0, st, a);
    size_t lds = (size_t)POST_LDS_KEYS * 8 + (size_t)a.det.cap * (16 + 4 + 4) + 256 * 4 + 1024 * 4;
    static PerDeviceOnce once;
    allow_big_lds(once, post_nms_kernel);
    hipLaunchKernelGGL(post_nms_kernel, dim3(a.V), dim3(1024), lds, st, a);
}
