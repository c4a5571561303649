// retina.hip -- RetinaNet.postprocess_detections (reference detection/retinanet_cal.py:402-490) and the
// stock transform.postprocess: sigmoid over all anchor x class logits, per-class score threshold,
// anchor decode (weights 1,1,1,1) + clip, remove_small_boxes(1e-2), per-class NMS(0.5), first 300
// per class, detections concatenated in class order (labels 0..K-1; not globally sorted).
// The reference launches 21 NMS kernels + a host sync per class and view; here one workgroup handles
// one (class, view) and all of them run concurrently.
#include "common.h"
#include "kernels.h"
#include "sortnms.h"

__device__ inline int retina_level_of(const RetinaArgs& a, int v, int i, int* local) {
    int l = 0;
    int base = 0;
#pragma unroll
    for (int t = 0; t < 5; t++) {
        const int n = a.seg[t][v].H * a.seg[t][v].W * a.A;
        if (i >= base + n && t < 4) { base += n; l = t + 1; }
        else break;
    }
    *local = i - base;
    return l;
}

// Kernel 1: one thread per anchor: K sigmoids, candidate keys per (view, class).
__global__ __launch_bounds__(256) void retina_cand_kernel(RetinaArgs a) {
    const int v = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int total = 0;
#pragma unroll
    for (int t = 0; t < 5; t++) total += a.seg[t][v].H * a.seg[t][v].W * a.A;
    if (i >= total) return;
    int li;
    const int l = retina_level_of(a, v, i, &li);
    const LevelSeg sg = a.seg[l][v];
    const int pix = li / a.A, an = li - pix * a.A;
    const float* lg = a.cls[l] + (sg.pix_off + pix) * (long long)a.cls_ld + an * a.K;
    for (int k = 0; k < a.K; k++) {
        const float s = det_sigmoidf(lg[k]);
        if (s > a.score_thr) {
            const int slot = atomicAdd(&a.cand_count[v * a.K + k], 1);
            a.cand_key[((long long)v * a.K + k) * a.cand_cap + slot] =
                ((unsigned long long)det_orderable(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
    }
}

__device__ inline float4 retina_decode(const RetinaArgs& a, int v, int i, float Wr, float Hr) {
    int li;
    const int l = retina_level_of(a, v, i, &li);
    const LevelSeg sg = a.seg[l][v];
    const LevelSeg s0 = a.seg0[v];
    const int pix = li / a.A, an = li - pix * a.A;
    const int y = pix / sg.W, x = pix - y * sg.W;
    const int sth = s0.H / sg.H, stw = s0.W / sg.W;
    const float* ba = a.base_anchors + (l * a.A + an) * 4;
    const float anchor[4] = {(float)(x * stw) + ba[0], (float)(y * sth) + ba[1], (float)(x * stw) + ba[2], (float)(y * sth) + ba[3]};
    const float* rp = a.reg[l] + (sg.pix_off + pix) * (long long)a.reg_ld + an * 4;
    const float d[4] = {rp[0], rp[1], rp[2], rp[3]};
    float o[4];
    det_box_decode(anchor, d, 1.0f, 1.0f, 1.0f, 1.0f, o);
    float4 b;
    b.x = det_clamp(o[0], 0.0f, Wr); b.z = det_clamp(o[2], 0.0f, Wr);
    b.y = det_clamp(o[1], 0.0f, Hr); b.w = det_clamp(o[3], 0.0f, Hr);
    return b;
}

// Kernel 2: per (class, view): sort candidates, decode, small-box filter, NMS, keep <= per_class.
#define RET_LDS_KEYS 8192
__global__ __launch_bounds__(1024) void retina_class_nms_kernel(RetinaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    unsigned long long* lkeys = reinterpret_cast<unsigned long long*>(dyn);
    const int per = a.per_class;
    float4* kept_box = reinterpret_cast<float4*>(dyn + (size_t)RET_LDS_KEYS * 8);
    float* kept_area = reinterpret_cast<float*>(kept_box + per);
    int* keep_idx = reinterpret_cast<int*>(kept_area + per);
    int* dead_or = keep_idx + per;
    __shared__ int s_nk;
    const int k = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
    const long long slot = (long long)v * a.K + k;
    int n = a.cand_count[slot];
    if (n > a.cand_cap) n = a.cand_cap;
    int NP = 1024;
    while (NP < n) NP <<= 1;
    unsigned long long* gkeys = a.cand_key + slot * a.cand_cap;
    unsigned long long* keys = (NP <= RET_LDS_KEYS) ? lkeys : gkeys;
    for (int i = tid; i < NP; i += 1024) keys[i] = (i < n) ? gkeys[i] : 0ull;
    __syncthreads();
    block_bitonic_sort_desc(keys, NP);
    const ViewDesc vd = a.views[v];
    const float Wr = (float)vd.Wr, Hr = (float)vd.Hr;
    float4* cbox = reinterpret_cast<float4*>(a.cand_box) + slot * a.cand_cap;
    unsigned char* cskip = a.cand_skip + slot * a.cand_cap;
    for (int i = tid; i < n; i += 1024) {
        const int ai = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
        float4 b = retina_decode(a, v, ai, Wr, Hr);
        cskip[i] = ((b.z - b.x) >= a.min_box && (b.w - b.y) >= a.min_box) ? 0 : 1;   // remove_small_boxes
        cbox[i] = b;
    }
    __syncthreads();
    block_nms_sorted(cbox, n, a.nms_thr, per, kept_box, kept_area, dead_or, keep_idx, &s_nk, cskip);
    const int nk = s_nk;
    for (int i = tid; i < nk; i += 1024) {
        const int ci = keep_idx[i];
        a.kept_anchor[slot * per + i] = (int)(0xFFFFFFFFu - (unsigned)(keys[ci] & 0xFFFFFFFFull));
        reinterpret_cast<float4*>(a.kept_box)[slot * per + i] = cbox[ci];
    }
    if (tid == 0) a.kept_count[slot] = nk;
}

// Kernel 3: per (class, view): emit rows at the class's offset in the concatenated output.
__global__ __launch_bounds__(256) void retina_emit_kernel(RetinaArgs a) {
    const int k = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
    const int K = a.K, per = a.per_class, cap = a.det.cap;
    int off = 0, total = 0;
    for (int q = 0; q < K; q++) { const int c = a.kept_count[v * K + q]; if (q < k) off += c; total += c; }
    if (k == 0 && tid == 0) a.det.count[v] = total;
    const long long slot = (long long)v * K + k;
    const int nk = a.kept_count[slot];
    const ViewDesc vd = a.views[v];
    const float rh = (float)vd.Ho / (float)vd.Hr, rw = (float)vd.Wo / (float)vd.Wr;   // stock resize_boxes: fp32 ratio
    for (int i = tid; i < nk; i += 256) {
        const int ai = a.kept_anchor[slot * per + i];
        const float4 b = reinterpret_cast<const float4*>(a.kept_box)[slot * per + i];
        int li;
        const int l = retina_level_of(a, v, ai, &li);
        const LevelSeg sg = a.seg[l][v];
        const int pix = li / a.A, an = li - pix * a.A;
        const float* lg = a.cls[l] + (sg.pix_off + pix) * (long long)a.cls_ld + an * K;
        const long long o = (long long)v * cap + off + i;
        float pm = 0.0f;
        for (int q = 0; q < K; q++) {
            const float s = det_sigmoidf(lg[q]);
            a.det.scores_cls[o * K + q] = s;
            if (q == 0 || s > pm) pm = s;
            if (q == k) a.det.scores[o] = s;
        }
        a.det.prob_max[o] = pm;
        a.det.labels[o] = k;
        reinterpret_cast<float4*>(a.det.boxes)[o] = make_float4(b.x * rw, b.y * rh, b.z * rw, b.w * rh);
        reinterpret_cast<float4*>(a.det.props)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

void launch_retina_postprocess(const RetinaArgs& a, int max_anchors, hipStream_t st) {
    hipMemsetAsync(a.cand_count, 0, sizeof(int) * a.V * a.K, st);
    hipLaunchKernelGGL(retina_cand_kernel, dim3((max_anchors + 255) / 256, a.V), dim3(256), 0, st, a);
    size_t lds = (size_t)RET_LDS_KEYS * 8 + (size_t)a.per_class * (16 + 4 + 4) + 256 * 4;
    static PerDeviceOnce once;
    allow_big_lds(once, retina_class_nms_kernel);
    hipLaunchKernelGGL(retina_class_nms_kernel, dim3(a.K, a.V), dim3(1024), lds, st, a);
    hipLaunchKernelGGL(retina_emit_kernel, dim3(a.K, a.V), dim3(256), 0, st, a);
}
