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

// Greedy NMS over n score-sorted boxes, 64 candidates per round, all waves of the block take part.
//   phase A (parallel): wave w tests the candidates against kept[w], kept[w + nw], ... -> one ballot per wave;
//                       wave w also builds rows w, w + nw, ... of the 64 x 64 intra-chunk overlap matrix
//                       (bit l of row s = box s suppresses box l, l > s).
//   phase B (wave 0):   the greedy order inside the chunk is resolved on 64-bit masks only.
// kept_box / kept_area: LDS scratch of max_keep entries.  dead_or: unused (kept for the callers' LDS layouts).
// keep_out (LDS or global): indices (into the sorted order) of kept boxes.  Returns count via *nk_out.
// skip (optional, global): 1 = the box takes no part in NMS (neither kept nor suppressing).
// Must be called by ALL threads of the block (contains __syncthreads); blockDim.x a multiple of 64, <= 1024.
__device__ inline void block_nms_sorted(const float4* boxes, int n, float thr, int max_keep, float4* kept_box,
                                        float* kept_area, int* dead_or, int* keep_out, int* nk_out,
                                        const unsigned char* skip = nullptr) {
    (void)dead_or;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    __shared__ int s_nk;
    __shared__ unsigned long long s_row[64], s_dead[16];
    if (tid == 0) s_nk = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int nk = s_nk;
        if (nk >= max_keep) break;
        const int j = c0 + lane;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        bool dead = !(j < n);
        if (!dead) { b = boxes[j]; if (skip && skip[j]) dead = true; }   // skip[]: boxes filtered out before NMS (remove_small_boxes)
        if (!dead) {
            for (int t = wave; t < nk; t += nw)
                if (nms_overlaps(kept_box[t], kept_area[t], b, thr)) { dead = true; break; }
        }
        const unsigned long long dm = __ballot(dead);
        if (lane == 0) s_dead[wave] = dm;
        for (int s = wave; s < 64; s += nw) {
            float4 bs;
            bs.x = __shfl(b.x, s, 64); bs.y = __shfl(b.y, s, 64); bs.z = __shfl(b.z, s, 64); bs.w = __shfl(b.w, s, 64);
            const float as = (bs.z - bs.x) * (bs.w - bs.y);
            const unsigned long long row = __ballot(lane > s && nms_overlaps(bs, as, b, thr));
            if (lane == 0) s_row[s] = row;
        }
        __syncthreads();
        if (wave == 0) {
            unsigned long long deadm = 0ull;
            for (int w = 0; w < nw; w++) deadm |= s_dead[w];
            unsigned long long alive = ~deadm, keepm = 0ull;
            int cnt = nk;
            for (int s = 0; s < 64; s++) {
                if (!((alive >> s) & 1ull)) continue;
                keepm |= 1ull << s;
                cnt++;
                if (cnt >= max_keep) break;
                alive &= ~s_row[s];
            }
            if ((keepm >> lane) & 1ull) {
                const int p = nk + __popcll(keepm & ((1ull << lane) - 1ull));
                kept_box[p] = b; kept_area[p] = (b.z - b.x) * (b.w - b.y); keep_out[p] = j;
            }
            if (lane == 0) s_nk = cnt;
        }
        __syncthreads();
    }
    if (tid == 0) *nk_out = s_nk;
    __syncthreads();
}
