// rpn.hip -- region proposals: anchors, box decode, per-level top-k, clip, small-box filter,
// level-batched NMS, post-NMS top-n.  Restates torchvision 0.8.2 RegionProposalNetwork
// (in-repo copy: detection/frcnn_ll.py:284-321 filter_proposals, :323-374 forward; parameters
// detection/frcnn_la.py:154-158, :185-203).  Integer / index work is exact; ties are broken by
// (score desc, anchor index asc).
#include "common.h"
#include "kernels.h"
#include "sortnms.h"

// ---------------------------------------------------------------------------------------------
// Kernel 1: per (level, view): exact top-k of the objectness logits by 8-pass radix select on
// 64-bit keys, bitonic sort of the <=CAP survivors in LDS, anchor decode + clip + min-size test.
// grid = (5, V), block = 1024.  CAP = 1024 (inference: pre_nms_top_n 1000) or 2048 (training: 2000).
// ---------------------------------------------------------------------------------------------
template <int CAP>
__global__ __launch_bounds__(1024) void rpn_topk_kernel(RpnArgs a) {
    __shared__ unsigned long long sel[CAP];
    __shared__ int hist[256];
    __shared__ unsigned long long s_prefix, s_mask;
    __shared__ int s_remaining, s_cnt, s_done;
    const int l = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
    const LevelSeg sg = a.seg[l][v];
    const int A = a.A, n = sg.H * sg.W * A;
    const int k = n < a.pre_n ? n : a.pre_n;
    const float* head = a.head[l] + sg.pix_off * (long long)a.head_ld;
    auto key_of = [&](int i) -> unsigned long long {
        const int pix = i / A, an = i - pix * A;
        const float lg = head[(long long)pix * a.head_ld + an];
        return ((unsigned long long)det_orderable(lg) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    };
    for (int i = tid; i < CAP; i += 1024) sel[i] = 0ull;
    if (tid == 0) { s_prefix = 0ull; s_mask = 0ull; s_remaining = k; s_cnt = 0; s_done = 0; }
    __syncthreads();
    if (n > k) {
        for (int pass = 0; pass < 8; pass++) {
            const int shift = 56 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned long long prefix = s_prefix, mask = s_mask;
            for (int i = tid; i < n; i += 1024) {
                const unsigned long long key = key_of(i);
                if ((key & mask) == prefix) atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int cum = 0, rem = s_remaining;
                for (int b = 255; b >= 0; b--) {
                    const int h = hist[b];
                    if (cum + h >= rem) {
                        s_remaining = rem - cum;
                        s_prefix = prefix | ((unsigned long long)b << shift);
                        s_mask = mask | (255ull << shift);
                        // the bucket holds exactly as many keys as are still wanted: all of them are in, and the k-th largest key is
                        // >= this prefix with zeros below -- the remaining digits (mostly the index half of the key) need no pass
                        if (h == rem - cum) s_done = 1;
                        break;
                    }
                    cum += h;
                }
            }
            __syncthreads();
            if (s_done) break;
        }
    }
    const unsigned long long thresh = (n > k) ? s_prefix : 0ull;   // lower bound of the k largest keys (keys are unique)
    unsigned long long below = 0ull;                                // audit only: the best key the cut leaves out
    for (int i = tid; i < n; i += 1024) {
        const unsigned long long key = key_of(i);
        if (key >= thresh) { const int slot = atomicAdd(&s_cnt, 1); if (slot < CAP) sel[slot] = key; }
        else if (key > below) below = key;
    }
    __syncthreads();
    block_bitonic_sort_desc(sel, CAP);
    if (a.next_key) {       // decision-margin audit: the k-th key and the (k + 1)-th (0 when the level has no more than k anchors)
        if (tid == 0) s_prefix = 0ull;
        __syncthreads();
        if (below) atomicMax(&s_prefix, below);
        __syncthreads();
        if (tid == 0) {
            unsigned long long* nk = a.next_key + ((long long)v * 5 + l) * 2;
            nk[0] = (n > k && k > 0) ? sel[k - 1] : 0ull; nk[1] = (n > k) ? s_prefix : 0ull;
        }
    }
    // decode the t-th best anchor
    for (int t = tid; t < a.pre_n; t += 1024) {
        unsigned long long outkey = 0ull;
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < k) {
            const unsigned long long key = sel[t];
            const int i = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
            const int pix = i / A, an = i - pix * A;
            const int y = pix / sg.W, x = pix - y * sg.W;
            const LevelSeg s0 = a.seg0[v];
            const int sth = s0.H / sg.H, stw = s0.W / sg.W;
            const float* ba = a.base_anchors + (l * A + an) * 4;
            float anchor[4] = {(float)(x * stw) + ba[0], (float)(y * sth) + ba[1], (float)(x * stw) + ba[2], (float)(y * sth) + ba[3]};
            const float* hp = head + (long long)pix * a.head_ld;
            float d[4] = {hp[A + 4 * an], hp[A + 4 * an + 1], hp[A + 4 * an + 2], hp[A + 4 * an + 3]};
            float o[4];
            det_box_decode(anchor, d, 1.0f, 1.0f, 1.0f, 1.0f, o);
            const float Wr = (float)a.views[v].Wr, Hr = (float)a.views[v].Hr;
            box.x = det_clamp(o[0], 0.0f, Wr); box.z = det_clamp(o[2], 0.0f, Wr);
            box.y = det_clamp(o[1], 0.0f, Hr); box.w = det_clamp(o[3], 0.0f, Hr);
            const bool ok = (box.z - box.x) >= a.min_size && (box.w - box.y) >= a.min_size;
            if (ok) {
                outkey = (key & 0xFFFFFFFF00000000ull) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(l * a.pre_n + t));
                // largest coordinate over the view's valid candidates (batched_nms offsets): coordinates are clamped to >= 0, so the
                // bit pattern orders like the value
                const float m = fmaxf(fmaxf(box.x, box.y), fmaxf(box.z, box.w));
                if (m > 0.0f) atomicMax(reinterpret_cast<unsigned*>(a.sorted_count) + v, __float_as_uint(m));
            }
        }
        const long long o = (long long)v * 5 * a.pre_n + l * a.pre_n + t;
        a.cand_key[o] = outkey;
        reinterpret_cast<float4*>(a.cand_box)[o] = box;
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel 2: per (level, view): greedy NMS of the level's <= pre_n candidates, which kernel 1 left in score order.  torchvision's
// batched_nms separates the levels by the coordinate offset level * (max_coord + 1), so NMS never acts across levels: the five
// levels of a view run as five workgroups instead of one (the offset is still applied, in fp32, because it changes the rounding of
// the IoU).  Writes a keep flag per candidate.  grid = (5, V), block = 512, dynamic LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void rpn_level_nms_kernel(RpnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    float4* kept_box = reinterpret_cast<float4*>(dyn);                                     // post_n
    float* kept_area = reinterpret_cast<float*>(dyn + (size_t)a.post_n * 16);
    int* keep_idx = reinterpret_cast<int*>(kept_area + a.post_n);                          // post_n
    __shared__ int s_nk;
    const int l = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
    const long long base = (long long)v * 5 * a.pre_n + (long long)l * a.pre_n;
    const unsigned long long* ck = a.cand_key + base;
    const float4* cb = reinterpret_cast<const float4*>(a.cand_box) + base;
    float4* sb = reinterpret_cast<float4*>(a.sorted_box) + base;
    unsigned char* skip = reinterpret_cast<unsigned char*>(a.sorted_raw) + (long long)v * 10 * a.pre_n + (long long)l * a.pre_n;          // [V][2][5*pre_n] bytes: skip, keep
    unsigned char* keep = skip + 5 * a.pre_n;
    const float maxc = __uint_as_float(reinterpret_cast<const unsigned*>(a.sorted_count)[v]);
    const float off = (float)l * (maxc + 1.0f);
    for (int i = tid; i < a.pre_n; i += 512) {
        const float4 b = cb[i];
        sb[i] = make_float4(b.x + off, b.y + off, b.z + off, b.w + off);
        skip[i] = ck[i] == 0ull;
        keep[i] = 0;
    }
    __syncthreads();   // global writes by this block are read back below by other threads
    block_nms_sorted(sb, a.pre_n, a.nms_thr, a.post_n, kept_box, kept_area, nullptr, keep_idx, &s_nk, skip);
    const int nk = s_nk;
    for (int i = tid; i < nk; i += 512) keep[keep_idx[i]] = 1;
}

// ---------------------------------------------------------------------------------------------
// Kernel 3: per view: the kept candidates of the five levels in (score desc, position asc) order, first post_n of them.
// grid = V, block = 1024, dynamic LDS (NP keys).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void rpn_merge_kernel(RpnArgs a, int NP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(dyn);
    __shared__ int s_nk;
    const int v = blockIdx.x, tid = threadIdx.x;
    const int ntot = 5 * a.pre_n;
    const unsigned long long* ck = a.cand_key + (long long)v * ntot;
    const float4* cb = reinterpret_cast<const float4*>(a.cand_box) + (long long)v * ntot;
    const unsigned char* keep = reinterpret_cast<const unsigned char*>(a.sorted_raw) + (long long)v * 10 * a.pre_n + 5 * a.pre_n;
    if (tid == 0) s_nk = 0;
    int local = 0;
    for (int i = tid; i < NP; i += 1024) {
        const unsigned long long key = (i < ntot && keep[i]) ? ck[i] : 0ull;
        keys[i] = key;
        local += key != 0ull;
    }
    __syncthreads();
    if (local) atomicAdd(&s_nk, local);
    block_bitonic_sort_desc(keys, NP);
    int nk = s_nk; if (nk > a.post_n) nk = a.post_n;
    float4* pr = reinterpret_cast<float4*>(a.proposals) + (long long)v * a.prop_stride;
    for (int i = tid; i < nk; i += 1024) pr[i] = cb[(int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull))];
    if (tid == 0) {
        a.prop_count[v] = nk;
        if (a.trunc_key) {   // decision-margin audit: the last key inside the post-NMS cut and the first one outside
            const bool cut = s_nk > a.post_n;
            a.trunc_key[2 * v] = cut ? keys[a.post_n - 1] : 0ull; a.trunc_key[2 * v + 1] = cut ? keys[a.post_n] : 0ull;
        }
    }
}

void launch_rpn(const RpnArgs& a, hipStream_t st) {
    hipMemsetAsync(a.sorted_count, 0, sizeof(int) * a.V, st);      // per-view max coordinate (float bits), filled by kernel 1
    if (a.pre_n <= 1024) hipLaunchKernelGGL(rpn_topk_kernel<1024>, dim3(5, a.V), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL(rpn_topk_kernel<2048>, dim3(5, a.V), dim3(1024), 0, st, a);     // pre_n <= 2048 (checked by the callers)
    hipLaunchKernelGGL(rpn_level_nms_kernel, dim3(5, a.V), dim3(512), (size_t)a.post_n * 24, st, a);
    int NP = 1024;
    while (NP < 5 * a.pre_n) NP <<= 1;
    static PerDeviceOnce once;
    allow_big_lds(once, rpn_merge_kernel);
    hipLaunchKernelGGL(rpn_merge_kernel, dim3(a.V), dim3(1024), (size_t)NP * 8, st, a, NP);
}
