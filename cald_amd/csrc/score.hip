// score.hip -- the CALD consistency score (reference cald_train.py:189-224, hot loops 3-4 of
// SURVEY.md section 3.2) and the per-view class-max vector (cald_train.py:114-117, :194-197).
//
// The reference runs ~25 tiny torch kernels + 6 host syncs per (reference box, augmentation);
// here one workgroup scores one (image, augmentation) pair: the augmented view's boxes are staged in LDS
// once (every one of the <= 50 reference boxes walks the whole list), a wavefront per reference box does the
// IoU row + first-index argmax across lanes, then the Jensen-Shannon divergence with one class per
// lane and 64-lane butterfly reductions (the same addition order the oracle's wave_sum uses).
#include "common.h"
#include "kernels.h"

__device__ inline float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

// scipy.special.rel_entr
__device__ inline float rel_entr_f(float x, float y) {
    if (x != x || y != y) return NAN;
    if (x > 0.0f && y > 0.0f) return x * det_logf(x / y);
    if (x == 0.0f && y >= 0.0f) return 0.0f;
    return INFINITY;
}

__device__ inline float cald_iou(const float4 ab, const float4 B) {
    float w = fminf(ab.z, B.z) - fmaxf(ab.x, B.x);
    float h = fminf(ab.w, B.w) - fmaxf(ab.y, B.y);
    float Aarea = (ab.z - ab.x) * (ab.w - ab.y);
    float Barea = (B.z - B.x) * (B.w - B.y);
    float inter = w * h;
    float iou = inter / ((Aarea + Barea) - inter);
    if (w < 0.0f) iou = 0.0f;
    if (h < 0.0f) iou = 0.0f;
    return iou;
}

// "is (v1, j1) a better argmax than (v2, j2)": torch.argmax = first maximum, NaN is maximal.
__device__ inline bool better(float v1, int j1, float v2, int j2) {
    const bool n1 = v1 != v1, n2 = v2 != v2;
    if (n1 != n2) return n1;
    if (n1) return j1 < j2;
    return v1 > v2 || (v1 == v2 && j1 < j2);
}

// the reference box as the augmented view sees it (cald_helper.HorizontalFlip :29, resize :53, rotate :160-222)
__device__ inline float4 aug_box(float4 ab, const int kind, const float* prmv) {
    const float prm = prmv[0];
    if (kind == 1) { float x0 = prm - ab.z, x2 = prm - ab.x; ab.x = x0; ab.z = x2; }   // cald_helper.py:29
    else if (kind == 2) { ab.x = ab.x * prm; ab.y = ab.y * prm; ab.z = ab.z * prm; ab.w = ab.w * prm; }  // :53
    else if (kind == 3) {   // cald_helper.rotate box transform, cald_helper.py:160-222
        const float bw = ab.z - ab.x, bh = ab.w - ab.y;
        const float xs[4] = {ab.x, ab.x + bw, ab.x, ab.z}, ys[4] = {ab.y, ab.y, ab.y + bh, ab.w};
        float xmin = 0.f, xmax = 0.f, ymin = 0.f, ymax = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float X = (prmv[0] * xs[q] + prmv[1] * ys[q]) + prmv[2] * 1.0f;
            const float Y = (prmv[3] * xs[q] + prmv[4] * ys[q]) + prmv[5] * 1.0f;
            if (q == 0 || X < xmin) xmin = X; if (q == 0 || X > xmax) xmax = X;
            if (q == 0 || Y < ymin) ymin = Y; if (q == 0 || Y > ymax) ymax = Y;
        }
        ab.x = det_clamp(xmin / prmv[6], 0.0f, prmv[8]); ab.y = det_clamp(ymin / prmv[7], 0.0f, prmv[9]);
        ab.z = det_clamp(xmax / prmv[6], 0.0f, prmv[8]); ab.w = det_clamp(ymax / prmv[7], 0.0f, prmv[9]);
    }
    return ab;
}

#define SCORE_LDS_BOXES 2048      /* 32 KB: Faster R-CNN lists (<= 100) fit whole; RetinaNet's (<= classes x 300) go through in chunks */
__global__ __launch_bounds__(256) void consistency_kernel(ScoreArgs a) {
    __shared__ float wmin[4];
    __shared__ float4 s_box[SCORE_LDS_BOXES];
    const int p = blockIdx.x;
    const int rv = a.ref_view[p], av = a.aug_view[p], img = a.pair_img[p];
    const int N = a.ref_n[img];
    const int M = a.det.count[av];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cap = a.det.cap, C = a.det.C;
    if (M == 0) { if (threadIdx.x == 0) a.cons[p] = 0.0f; return; }
    const float4* rboxes = reinterpret_cast<const float4*>(a.det.boxes) + (long long)rv * cap;
    const float4* aboxes = reinterpret_cast<const float4*>(a.det.boxes) + (long long)av * cap;
    const int kind = a.aug_kind[p];
    const float* prmv = a.aug_param + (long long)p * 12;
    float cur = 1.0f;
    // N <= 50 reference boxes over 4 waves: every wave runs the same number of rounds so that the staging barriers line up
    const int rounds = (N + 3) / 4;
    const int nchunk = (M + SCORE_LDS_BOXES - 1) / SCORE_LDS_BOXES;
    for (int rd = 0; rd < rounds; rd++) {
        const int i = rd * 4 + wave;
        const bool live = i < N;
        const int ri = live ? a.ref_sel[img * 50 + i] : 0;
        const float4 ab = aug_box(rboxes[ri], kind, prmv);
        float best = -INFINITY; int bj = 0x7fffffff;
        for (int ch = 0; ch < nchunk; ch++) {
            const int j0 = ch * SCORE_LDS_BOXES, nj = (M - j0 < SCORE_LDS_BOXES) ? M - j0 : SCORE_LDS_BOXES;
            if (rd == 0 || nchunk > 1) {      // one chunk (the usual case): staged once, reused by every round
                __syncthreads();
                for (int j = threadIdx.x; j < nj; j += 256) s_box[j] = aboxes[j0 + j];
                __syncthreads();
            }
            for (int j = lane; j < nj; j += 64) {
                float v = cald_iou(ab, s_box[j]);
                if (better(v, j0 + j, best, bj)) { best = v; bj = j0 + j; }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            float ov = __shfl_xor(best, off, 64); int oj = __shfl_xor(bj, off, 64);
            if (better(ov, oj, best, bj)) { best = ov; bj = oj; }
        }
        if (!live) continue;
        // Jensen-Shannon divergence through scipy.stats.entropy semantics (float32, renormalised)
        const float* pv = a.det.scores_cls + ((long long)rv * cap + ri) * C;
        const float* qv = a.det.scores_cls + ((long long)av * cap + bj) * C;
        float sp = 0.0f, sq = 0.0f, sm = 0.0f;
        for (int k = lane; k < C; k += 64) {
            float pk = pv[k], qk = qv[k];
            sp = sp + pk; sq = sq + qk; sm = sm + (pk + qk) / 2.0f;
        }
        sp = wave_sum64(sp); sq = wave_sum64(sq); sm = wave_sum64(sm);
        float t1 = 0.0f, t2 = 0.0f;
        for (int k = lane; k < C; k += 64) {
            float pk = pv[k], qk = qv[k];
            float mk = ((pk + qk) / 2.0f) / sm;
            t1 = t1 + rel_entr_f(pk / sp, mk);
            t2 = t2 + rel_entr_f(qk / sq, mk);
        }
        t1 = wave_sum64(t1); t2 = wave_sum64(t2);
        float js = 0.5f * t1 + 0.5f * t2;
        if (js < 0.0f) js = 0.0f;
        const float t = 0.5f * (1.0f - js);
        const float u = a.det.prob_max[(long long)rv * cap + ri] + a.det.prob_max[(long long)av * cap + bj];
        const float s = fabsf((best + t * u) - a.bp);
        if (s < cur) cur = s;
    }
    if (lane == 0) wmin[wave] = cur;
    __syncthreads();
    if (threadIdx.x == 0) {
        float c = 1.0f;
        for (int w = 0; w < 4; w++) if (wmin[w] < c) c = wmin[w];
        a.cons[p] = c;
    }
}

void launch_consistency(const ScoreArgs& a, hipStream_t st) {
    if (a.P <= 0) return;
    hipLaunchKernelGGL(consistency_kernel, dim3(a.P), dim3(256), 0, st, a);
}

// Decision-margin audit of the scoring loop (audit.hip explains the idea): the only discrete step is q = scores_cls[argmax(iou)].
// Per pair: the smallest (best IoU - best IoU among detections that stem from ANOTHER proposal) over the reference boxes -- two
// detections of one proposal carry the same scores_cls / prob_max row (frcnn_la.py:55-56, :64-65), so a flip between them changes
// nothing but the IoU itself, continuously -- and whether some row is all zero (argmax = 0: the FIRST detection is taken, whichever
// that is).  One workgroup per pair, one wavefront per reference box.
__global__ __launch_bounds__(256) void pair_audit_kernel(ScoreArgs a, float* out) {
    __shared__ float wgap[4];
    __shared__ int wzero[4];
    const int p = blockIdx.x;
    const int rv = a.ref_view[p], av = a.aug_view[p], img = a.pair_img[p];
    const int N = a.ref_n[img], M = a.det.count[av];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cap = a.det.cap;
    if (M == 0) { if (threadIdx.x == 0) { out[2 * p] = INFINITY; out[2 * p + 1] = 0.0f; } return; }
    const float4* rboxes = reinterpret_cast<const float4*>(a.det.boxes) + (long long)rv * cap;
    const float4* aboxes = reinterpret_cast<const float4*>(a.det.boxes) + (long long)av * cap;
    const float4* aprops = reinterpret_cast<const float4*>(a.det.props) + (long long)av * cap;
    const int kind = a.aug_kind[p];
    const float* prmv = a.aug_param + (long long)p * 12;
    float gap = INFINITY; int zero = 0;
    for (int i = wave; i < N; i += 4) {
        const float4 ab = aug_box(rboxes[a.ref_sel[img * 50 + i]], kind, prmv);
        float best = -INFINITY; int bj = 0x7fffffff;
        for (int j = lane; j < M; j += 64) { const float v = cald_iou(ab, aboxes[j]); if (better(v, j, best, bj)) { best = v; bj = j; } }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64); const int oj = __shfl_xor(bj, off, 64);
            if (better(ov, oj, best, bj)) { best = ov; bj = oj; }
        }
        if (!(best > 0.0f)) { zero = 1; continue; }        // all-zero (or NaN) row
        const float4 pb = aprops[bj];
        float second = 0.0f;
        for (int j = lane; j < M; j += 64) {
            const float4 pj = aprops[j];
            if (pj.x == pb.x && pj.y == pb.y && pj.z == pb.z && pj.w == pb.w) continue;
            const float v = cald_iou(ab, aboxes[j]);
            if (v > second) second = v;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) second = fmaxf(second, __shfl_xor(second, off, 64));
        gap = fminf(gap, best - second);
    }
    if (lane == 0) { wgap[wave] = gap; wzero[wave] = zero; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * p] = fminf(fminf(wgap[0], wgap[1]), fminf(wgap[2], wgap[3]));
        out[2 * p + 1] = (wzero[0] | wzero[1] | wzero[2] | wzero[3]) ? 1.0f : 0.0f;
    }
}
void launch_pair_audit(const ScoreArgs& a, float* out, hipStream_t st) {
    if (a.P <= 0) return;
    hipLaunchKernelGGL(pair_audit_kernel, dim3(a.P), dim3(256), 0, st, a, out);
}

// cls_corr[l-1] = max(cls_corr[l-1], score) with python negative indexing for label 0 (RetinaNet).
// Reference views use only their sub-sampled detections (cald_train.py:110-117).
__global__ __launch_bounds__(256) void cls_corr_kernel(DetBuffers det, const int* ref_sel, const int* ref_n,
                                                       const int* view_img, const int* view_is_ref, float* out) {
    __shared__ int smax[256];
    const int v = blockIdx.x;
    const int C = det.C, cap = det.cap;
    for (int k = threadIdx.x; k < C - 1; k += 256) smax[k] = 0;
    __syncthreads();
    const bool is_ref = view_is_ref[v] != 0;
    const int img = view_img[v];
    const int n = is_ref ? ref_n[img] : det.count[v];
    for (int d = threadIdx.x; d < n; d += 256) {
        const int di = is_ref ? ref_sel[img * 50 + d] : d;
        long long l = det.labels[(long long)v * cap + di] - 1;
        if (l < 0) l += C - 1;
        if (l < 0 || l >= C - 1) continue;
        const float s = det.scores[(long long)v * cap + di];
        if (s > 0.0f) atomicMax(&smax[l], __float_as_int(s));
    }
    __syncthreads();
    for (int k = threadIdx.x; k < C - 1; k += 256) out[(long long)v * (C - 1) + k] = __int_as_float(smax[k]);
}

void launch_cls_corr(const DetBuffers& det, const int* ref_sel, const int* ref_n, const int* view_img, const int* view_is_ref,
                     int V, float* out, hipStream_t st) {
    if (V <= 0) return;
    hipLaunchKernelGGL(cls_corr_kernel, dim3(V), dim3(256), 0, st, det, ref_sel, ref_n, view_img, view_is_ref, out);
}


// ---------------------------------------------------------------------------------------------
// scoring kernels of the baseline sweeps that share the detector forward.
// lt_c_train.py:92-121: min over detections of |calcu_iou(box, prop) + prob_max - 1| (init 1.0).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lt_uncertainty_kernel(DetBuffers det, float* out) {
    __shared__ float red[256];
    const int v = blockIdx.x, n = det.count[v], cap = det.cap;
    const float4* boxes = reinterpret_cast<const float4*>(det.boxes) + (long long)v * cap;
    const float4* props = reinterpret_cast<const float4*>(det.props) + (long long)v * cap;
    float unc = 1.0f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float4 A = boxes[i], B = props[i];
        const float width = (fminf(A.z, B.z) - fmaxf(A.x, B.x)) + 1.0f;
        const float height = (fminf(A.w, B.w) - fmaxf(A.y, B.y)) + 1.0f;
        float iou = 0.0f;
        if (!(width <= 0.0f || height <= 0.0f)) {
            const float Aarea = (A.z - A.x) * ((A.w - A.y) + 1.0f);
            const float Barea = (B.z - B.x) * ((B.w - B.y) + 1.0f);
            const float iner = width * height;
            iou = iner / ((Aarea + Barea) - iner);
        }
        const float u = fabsf((iou + det.prob_max[(long long)v * cap + i]) - 1.0f);
        if (u < unc) unc = u;
    }
    red[threadIdx.x] = unc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s && red[threadIdx.x + s] < red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) out[v] = red[0];
}
void launch_lt_uncertainty(const DetBuffers& det, int V, float* out, hipStream_t st) {
    if (V > 0) hipLaunchKernelGGL(lt_uncertainty_kernel, dim3(V), dim3(256), 0, st, det, out);
}

// ls_c_train.py:136-150: for every selected reference box the maximum IoU against one noisy view's detections.
__global__ __launch_bounds__(256) void max_iou_kernel(ScoreArgs a, float* out) {
    const int p = blockIdx.x;
    const int rv = a.ref_view[p], av = a.aug_view[p], img = a.pair_img[p];
    const int N = a.ref_n[img], M = a.det.count[av], cap = a.det.cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4* rboxes = reinterpret_cast<const float4*>(a.det.boxes) + (long long)rv * cap;
    const float4* aboxes = reinterpret_cast<const float4*>(a.det.boxes) + (long long)av * cap;
    for (int i = wave; i < N; i += 4) {
        const float4 ab = rboxes[a.ref_sel[img * 50 + i]];
        float best = -INFINITY; int bj = 0x7fffffff;
        for (int j = lane; j < M; j += 64) {
            const float v = cald_iou(ab, aboxes[j]);
            if (better(v, j, best, bj)) { best = v; bj = j; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64); const int oj = __shfl_xor(bj, off, 64);
            if (better(ov, oj, best, bj)) { best = ov; bj = oj; }
        }
        if (lane == 0) out[(long long)p * 50 + i] = M > 0 ? best : 0.0f;
    }
}
void launch_max_iou(const ScoreArgs& a, float* out, hipStream_t st) {
    if (a.P > 0) hipLaunchKernelGGL(max_iou_kernel, dim3(a.P), dim3(256), 0, st, a, out);
}
