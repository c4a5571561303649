// audit.hip -- decision margins of one Faster R-CNN forward (cald_sweep_audit; the cascade of DESIGN.md section 6b).
//
// The sweep's result is a continuous function of the GEMM outputs EXCEPT at its discrete decisions: the per-level top-k cut and the
// NMS of the RPN (detection/frcnn_ll.py:284-321), the post-NMS top-n, the RoI level mapper and RoIAlign's border rule
// (torchvision MultiScaleRoIAlign, detection/frcnn_la.py:205-209), the score threshold / class-batched NMS / top-100 of
// postprocess_detections (detection/frcnn_la.py:72-80), and -- in the scoring loop -- argmax over the IoU row and the linspace
// sub-sample of a sorted list (cald_train.py:110-113, :214).  A forward computed with slightly different rounding (precision f16x3)
// reproduces the exact mode's result to ~1e-6 unless one of those decisions comes out differently.  These kernels record, per view
// and per kind of decision, the smallest distance of any RELEVANT decision to its flip point; a view whose margins all exceed the
// rounding noise took the same decisions in both modes.
//
// Relevance: the merged post-NMS proposal list is cut at post_n entries by score.  A decision that only concerns boxes whose score
// lies more than `delta` below that cut cannot change the output (a box is affected only by higher-scored boxes, and everything
// below the cut is dropped), so it is ignored -- otherwise the top-k cut of the finest level alone (60 800 x 3 logits, gaps of 1e-4
// at rank 1000) would flag every other view.  Likewise a near-threshold (proposal, class) score matters only if the box would
// survive NMS, and candidates behind the top-100 cut never matter.
//
// Nothing here feeds the detections: the audit reads the forward's own scratch buffers after the fact.
#include "common.h"
#include "kernels.h"

namespace {
__device__ __forceinline__ float key_score(unsigned long long key) {       // inverse of det_orderable on the key's high word
    const unsigned k = (unsigned)(key >> 32);
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
__device__ __forceinline__ void upd(float* slot, float m) {
    if (!(m >= 0.0f)) m = 0.0f;                                              // NaN or negative: flag
    atomicMin(reinterpret_cast<unsigned*>(slot), __float_as_uint(m));       // non-negative floats order like their bit patterns
}
// the IoU torchvision's nms computes (sortnms.h nms_overlaps, same operation order)
__device__ __forceinline__ float nms_iou(const float4 bi, const float4 bj) {
    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
    const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    const float ai = (bi.z - bi.x) * (bi.w - bi.y), aj = (bj.z - bj.x) * (bj.w - bj.y);
    return inter / ((ai + aj) - inter);
}
__global__ void audit_fill_kernel(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = INFINITY;
}

// ---------------------------------------------------------------------------------------------
// RPN: grid (5 levels, V), block 256.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void audit_rpn_kernel(AuditArgs a) {
    __shared__ float4 kb[1024];
    __shared__ float ks[1024];
    __shared__ int kidx[1024];
    __shared__ int s_nk;
    const int l = blockIdx.x, v = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int pre = a.pre_n;
    const long long base = (long long)v * 5 * pre + (long long)l * pre;
    const unsigned long long* ck = a.cand_key + base;
    const float4* cb = reinterpret_cast<const float4*>(a.cand_box) + base;
    const float4* sb = reinterpret_cast<const float4*>(a.sorted_box) + base;       // level offset applied: what NMS compared
    const unsigned char* skip = a.flags + (long long)v * 10 * pre + (long long)l * pre;
    const unsigned char* keep = skip + 5 * pre;
    float* out = a.out + (long long)v * CALD_VM;
    const unsigned long long t0 = a.trunc_key[2 * v], t1 = a.trunc_key[2 * v + 1];
    const float s_cut = t0 ? key_score(t0) - a.delta : -INFINITY;                  // scores below this cannot reach the output
    if (tid == 0) s_nk = 0;
    __syncthreads();
    if (tid < 64) {          // kept boxes of the level, in score order (one wave: ballots keep the order)
        int nk = 0;
        for (int c0 = 0; c0 < pre && c0 < 1024; c0 += 64) {
            const int j = c0 + lane;
            const bool kp = j < pre && keep[j];
            const unsigned long long m = __ballot(kp);
            if (kp) { const int p = nk + __popcll(m & ((1ull << lane) - 1ull)); kb[p] = sb[j]; ks[p] = key_score(ck[j]); kidx[p] = j; }
            nk += __popcll(m);
        }
        if (lane == 0) s_nk = nk;
    }
    __syncthreads();
    const int nk = s_nk;
    if (tid == 0) {
        const unsigned long long k1 = a.next_key[((long long)v * 5 + l) * 2], k2 = a.next_key[((long long)v * 5 + l) * 2 + 1];
        if (k1 && k2) { const float sk = key_score(k1); if (sk >= s_cut) upd(out + VM_RPN_TOPK, sk - key_score(k2)); }
        if (l == 0 && t1) upd(out + VM_RPN_TRUNC, key_score(t0) - key_score(t1));
    }
    float m_small = INFINITY, m_iou = INFINITY, m_ord = INFINITY;
    for (int j = tid; j < pre; j += 256) {
        const float4 b = cb[j];
        if (b.x != 0.0f || b.y != 0.0f || b.z != 0.0f || b.w != 0.0f)              // slots beyond the level's anchors hold zeros
            m_small = fminf(m_small, fminf(fabsf((b.z - b.x) - a.min_size), fabsf((b.w - b.y) - a.min_size)));
        if (skip[j] || ck[j] == 0ull) continue;
        const float sj = key_score(ck[j]);
        if (!(sj >= s_cut)) continue;
        const float4 bj = sb[j];
        float mx = 0.0f;
        for (int i = 0; i < nk && kidx[i] < j; i++) {
            const float iou = nms_iou(kb[i], bj);
            if (iou > mx) mx = iou;
            if (iou > a.rpn_nms_thr) m_ord = fminf(m_ord, ks[i] - sj);              // i suppresses j: had j scored higher, j would suppress i
        }
        m_iou = fminf(m_iou, fabsf(mx - a.rpn_nms_thr));
    }
    if (m_small < INFINITY) upd(out + VM_RPN_SMALL, m_small);
    if (m_iou < INFINITY) upd(out + VM_RPN_IOU, m_iou);
    if (m_ord < INFINITY) upd(out + VM_RPN_ORDER, m_ord);
}

// ---------------------------------------------------------------------------------------------
// RoI level mapper and RoIAlign border rule: grid V, block 256.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void audit_roi_kernel(AuditArgs a) {
    const int v = blockIdx.x, tid = threadIdx.x;
    const int n = a.prop_count[v];
    float m_lvl = INFINITY, m_edge = INFINITY;
    for (int r = tid; r < n; r += 256) {
        const float4 box = reinterpret_cast<const float4*>(a.proposals)[(long long)v * CALD_ROI_CAP + r];
        const float area = (box.z - box.x) * (box.w - box.y);
        const float val = (4.0f + det_log2f(sqrtf(area) / 224.0f)) + 1e-6f;         // roi_level(): floor(val) clamped to [2, 5]
        if (val == val) m_lvl = fminf(m_lvl, fminf(fabsf(val - 3.0f), fminf(fabsf(val - 4.0f), fabsf(val - 5.0f))));
        const int l = roi_level(box);
        const LevelSeg sg = a.seg[l][v];
        const float scale = 1.0f / (float)(4 << l);
        const float x1 = box.x * scale, y1 = box.y * scale, x2 = box.z * scale, y2 = box.w * scale;
        float rw = x2 - x1; if (!(rw >= 1.0f)) rw = 1.0f;
        float rh = y2 - y1; if (!(rh >= 1.0f)) rh = 1.0f;
        const float bw = rw / 7.0f, bh = rh / 7.0f;
        for (int s = 0; s < 14; s++) {                                               // roi_sample(): a sample outside [-1, size] reads zero
            const float ty = y1 + (float)(s >> 1) * bh + ((float)(s & 1) + 0.5f) * bh / 2.0f;
            const float tx = x1 + (float)(s >> 1) * bw + ((float)(s & 1) + 0.5f) * bw / 2.0f;
            m_edge = fminf(m_edge, fminf(fminf(fabsf(ty + 1.0f), fabsf(ty - (float)sg.H)), fminf(fabsf(tx + 1.0f), fabsf(tx - (float)sg.W))));
        }
    }
    if (m_lvl < INFINITY) upd(a.out + (long long)v * CALD_VM + VM_ROI_LEVEL, m_lvl);
    if (m_edge < INFINITY) upd(a.out + (long long)v * CALD_VM + VM_ROI_EDGE, m_edge);
}

// ---------------------------------------------------------------------------------------------
// postprocess_detections: grid V, block 256.
// ---------------------------------------------------------------------------------------------
#define AUDIT_MAX_DET 512
__device__ __forceinline__ float4 audit_cand_box(const AuditArgs& a, int v, int r, int c, float Wr, float Hr, float maxc) {
    const float4 p = reinterpret_cast<const float4*>(a.proposals)[(long long)v * CALD_ROI_CAP + r];
    const float pb[4] = {p.x, p.y, p.z, p.w};
    const float* dl = a.pred + ((long long)v * CALD_ROI_CAP + r) * a.pred_ld + a.C + 4 * c;
    const float d[4] = {dl[0], dl[1], dl[2], dl[3]};
    float o[4];
    det_box_decode(pb, d, 10.0f, 10.0f, 5.0f, 5.0f, o);
    const float off = (float)c * (maxc + 1.0f);
    return make_float4(det_clamp(o[0], 0.0f, Wr) + off, det_clamp(o[1], 0.0f, Hr) + off, det_clamp(o[2], 0.0f, Wr) + off, det_clamp(o[3], 0.0f, Hr) + off);
}
__global__ __launch_bounds__(256) void audit_post_kernel(AuditArgs a) {
    __shared__ float4 kb[AUDIT_MAX_DET];
    __shared__ float ks[AUDIT_MAX_DET];
    __shared__ int kc[AUDIT_MAX_DET];
    __shared__ unsigned long long kk[AUDIT_MAX_DET];
    __shared__ unsigned char picked[AUDIT_MAX_DET];
    const int v = blockIdx.x, tid = threadIdx.x, C = a.C;
    float* out = a.out + (long long)v * CALD_VM;
    int n = a.key_count[v]; if (n > a.key_cap) n = a.key_cap;
    int nk = a.det_count[v]; if (nk > AUDIT_MAX_DET) nk = AUDIT_MAX_DET;
    const int np = a.prop_count[v];
    const ViewDesc vd = a.views[v];
    const float Wr = (float)vd.Wr, Hr = (float)vd.Hr;
    const float maxc = a.post_maxc[v];
    const bool full = nk >= a.cap;
    for (int i = tid; i < nk; i += 256) {
        const unsigned long long key = a.kept_key[(long long)v * a.cap + i];
        const int pos = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        const int r = pos / (C - 1), c = pos - r * (C - 1) + 1;
        kk[i] = key; ks[i] = key_score(key); kc[i] = c; kb[i] = audit_cand_box(a, v, r, c, Wr, Hr, maxc);
        picked[i] = 0;
    }
    __syncthreads();
    float m_thr = INFINITY, m_iou = INFINITY, m_ord = INFINITY, m_cap = INFINITY;
    // (a) scores next to the threshold, on either side: relevant unless a kept detection of the class covers the box anyway
    if (!full) {
        for (int idx = tid; idx < np * (C - 1); idx += 256) {
            const int r = idx / (C - 1), c = idx - r * (C - 1) + 1;
            const float p = a.prob[((long long)v * CALD_ROI_CAP + r) * C + c];
            const float d = fabsf(p - a.score_thr);
            if (!(d < 1e-3f) || !(d < m_thr)) continue;
            const float4 b = audit_cand_box(a, v, r, c, Wr, Hr, maxc);
            bool covered = false;
            for (int i = 0; i < nk && !covered; i++) covered = kc[i] == c && nms_iou(kb[i], b) > a.post_nms_thr + 0.01f;
            if (!covered) m_thr = d;
        }
    }
    // (b) every candidate against the kept detections of its class that precede it
    const unsigned long long last = full && nk > 0 ? kk[nk - 1] : 0ull;
    for (int q = tid; q < n; q += 256) {
        const unsigned long long key = a.keys[(long long)v * a.key_cap + q];
        const float sj = key_score(key);
        if (full && key < last) { m_cap = fminf(m_cap, ks[nk - 1] - sj); continue; }      // behind the top-n cut
        const int pos = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        const int r = pos / (C - 1), c = pos - r * (C - 1) + 1;
        const float4 b = audit_cand_box(a, v, r, c, Wr, Hr, maxc);
        float mx = 0.0f;
        for (int i = 0; i < nk; i++) {
            if (kc[i] != c || !(kk[i] > key)) continue;
            const float iou = nms_iou(kb[i], b);
            if (iou > mx) mx = iou;
            if (iou > a.post_nms_thr) m_ord = fminf(m_ord, ks[i] - sj);
        }
        m_iou = fminf(m_iou, fabsf(mx - a.post_nms_thr));
    }
    if (m_thr < INFINITY) upd(out + VM_POST_THR, m_thr);
    if (m_iou < INFINITY) upd(out + VM_POST_IOU, m_iou);
    if (m_ord < INFINITY) upd(out + VM_POST_ORDER, m_ord);
    if (m_cap < INFINITY) upd(out + VM_POST_CAP, m_cap);
    // (c) output order: the gap between the first two detections (argmax over an all-zero IoU row picks detection 0), and -- were this a
    // reference view with more than 40 detections -- the gaps between neighbours of which np.round(np.linspace(0, n - 1, 50)) picks one
    if (tid == 0) {
        if (nk >= 2) upd(out + VM_POST_TOP2, ks[0] - ks[1]);
        if (nk > 40) {
            const double step = (double)(nk - 1) / 49.0;
            for (int i = 0; i < 50; i++) { const int k = i == 49 ? nk - 1 : (int)nearbyint((double)i * step); if (k >= 0 && k < nk) picked[k] = 1; }
            float m = INFINITY;
            for (int i = 0; i + 1 < nk; i++) if (picked[i] != picked[i + 1]) m = fminf(m, ks[i] - ks[i + 1]);
            if (m < INFINITY) upd(out + VM_POST_SUBORDER, m);
        }
    }
}
}   // namespace

void launch_audit(const AuditArgs& a, hipStream_t st) {      // callers guarantee pre_n <= 1024 and det cap <= AUDIT_MAX_DET
    const int n = a.V * CALD_VM;
    hipLaunchKernelGGL(audit_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a.out, n);
    hipLaunchKernelGGL(audit_rpn_kernel, dim3(5, a.V), dim3(256), 0, st, a);
    hipLaunchKernelGGL(audit_roi_kernel, dim3(a.V), dim3(256), 0, st, a);
    hipLaunchKernelGGL(audit_post_kernel, dim3(a.V), dim3(256), 0, st, a);
}
