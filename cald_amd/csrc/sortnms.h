// sortnms.h -- block-level primitives shared by the RPN and box-head post-processing kernels:
//   * bitonic sort of 64-bit keys, descending  (key = orderable(score) << 32 | ~position, so the
//     order is (score desc, position asc) -- the deterministic tie rule of the arithmetic contract)
//   * greedy NMS of score-sorted boxes against the running kept list (torchvision nms semantics:
//     keep i, drop later j with IoU(i, j) > thr; areas (x2-x1)*(y2-y1), no +1)
#pragma once
#include "common.h"

// keys may point to LDS or global memory (flat addressing).  n_pow2 is a power of two.
__device__ inline void block_bitonic_sort_desc(unsigned long long* keys, int n_pow2) {
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n_pow2 >> 1); t += blockDim.x) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // element with bit j clear
                const int hi = lo | j;
                const bool desc = ((lo & k) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if (desc ? (a < b) : (a > b)) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
}

__device__ inline bool nms_overlaps(const float4 bi, const float ai, const float4 bj, const float thr) {
    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
    const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    const float aj = (bj.z - bj.x) * (bj.w - bj.y);
    const float ovr = inter / ((ai + aj) - inter);
    return ovr > thr;
}

// Greedy NMS over n score-sorted boxes, executed by the first 256 threads (4 waves) of the block.
// kept_box / kept_area: LDS scratch of max_keep entries; dead_or: LDS scratch of 4*64 ints.
// keep_out (LDS or global): indices (into the sorted order) of kept boxes.  Returns count via *nk_out.
// skip (optional, global): 1 = the box takes no part in NMS (neither kept nor suppressing).
// Must be called by ALL threads of the block (contains __syncthreads).
__device__ inline void block_nms_sorted(const float4* boxes, int n, float thr, int max_keep, float4* kept_box,
                                        float* kept_area, int* dead_or, int* keep_out, int* nk_out,
                                        const unsigned char* skip = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int s_nk;
    if (tid == 0) s_nk = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int nk = s_nk;
        if (nk >= max_keep) break;
        const int j = c0 + lane;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        bool dead = true;
        if (wave < 4) {
            dead = !(j < n);
            if (!dead) { b = boxes[j]; if (skip && skip[j]) dead = true; }   // skip[]: boxes filtered out before NMS (remove_small_boxes)
            // phase A: this wave tests kept[t], t = wave, wave+4, ...
            if (!dead) {
                for (int t = wave; t < nk; t += 4)
                    if (nms_overlaps(kept_box[t], kept_area[t], b, thr)) { dead = true; break; }
            }
            dead_or[wave * 64 + lane] = dead ? 1 : 0;
        }
        __syncthreads();
        if (wave == 0) {
            dead = (dead_or[lane] | dead_or[64 + lane] | dead_or[128 + lane] | dead_or[192 + lane]) != 0;
            int cnt = nk;
            // phase B: resolve the chunk in order
            for (int s = 0; s < 64; s++) {
                const unsigned long long alive = __ballot(!dead);
                if (!((alive >> s) & 1ull)) continue;
                float4 bs;
                bs.x = __shfl(b.x, s, 64); bs.y = __shfl(b.y, s, 64); bs.z = __shfl(b.z, s, 64); bs.w = __shfl(b.w, s, 64);
                const float as = (bs.z - bs.x) * (bs.w - bs.y);
                if (lane == 0) { kept_box[cnt] = bs; kept_area[cnt] = as; keep_out[cnt] = c0 + s; }
                cnt++;
                if (cnt >= max_keep) break;
                if (lane > s && !dead && nms_overlaps(bs, as, b, thr)) dead = true;
            }
            if (lane == 0) s_nk = cnt;
        }
        __syncthreads();
    }
    if (tid == 0) *nk_out = s_nk;
    __syncthreads();
}
