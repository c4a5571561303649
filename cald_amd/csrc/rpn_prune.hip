// rpn_prune.hip -- certified pruning of the RPN head in the exact (CALD_PRECISION_FP32) sweep.
//
// RegionProposalNetwork.filter_proposals (detection/frcnn_ll.py:284-321) keeps, per pyramid level, the pre_nms_top_n anchors with the
// largest objectness logit -- 1 000 of 91 200 on P2, 1 000 of 22 800 on P3 at VOC size -- and nothing downstream ever reads another
// anchor's logit or deltas.  The RPN head (3 x 3 conv 256 -> 256 + ReLU, 1 x 1 -> 15; frcnn_la.py:199-203) is 23 % of a view's FLOPs and
// P2 + P3 carry 94 % of its pixels.  The exact sweep therefore computes the head in two steps on those two levels:
//   1. everywhere, cheaply: the 3 x 3 conv on the fp16 matrix pipe (conv_h4.hip, 3 MFMAs per product, ~2.7 x the fp32 rate), the 1 x 1 head on
//      the exact kernel -> approximate logits L~ with an error bound B(anchor) against the exact mode's own value L (below);
//   2. only where it can matter, exactly: tau = the k-th largest LOWER bound L~ - B.  At least k anchors have L >= tau, so an anchor with
//      L~ + B < tau is not among the k largest of L, whatever the rounding did.  The pixels that hold a surviving anchor (~10 % of P2,
//      ~35 % of P3 on the configs[1] pool) are recomputed by the exact kernels as gathered rows (ConvArgs::row_map) -- per output element the same k-ordered
//      fp32 fma chain as the dense launch, hence the same bits -- and scattered back; all other anchors get logit -FLT_MAX.
// The top-k, decode, NMS and everything after see the exact mode's values at every anchor that can be selected: the detections, and
// with them scores and selection, are bit-identical to the unpruned sweep (tests: every sweep-vs-oracle test runs through this path;
// test_certified_rpn_pruning_* compares pruned and dense proposals at full size).
//
// The bound.  z = sum_j t_j, t_j = P_j w_cj over the K = 2 304 taps of hidden channel c, j = the tap's position in the k-ordered chain.
// The exact mode computes the fp32 fma chain s_k = fl(s_(k-1) + t_k): one rounding of relative size u = 2^-24 per step, so
// |z_e - z| <= u sum_k |s_k| <= u (1 + g_K) sum_j r_j |t_j| with r_j = K - j (term j takes part in that many partial sums; Higham, Accuracy and
// Stability, sect. 4.2 -- the textbook K u sum |t_j| is the r_j <= K relaxation of it).  Cauchy-Schwarz keeps the weights:
// sum_j r_j |P_j| |w_cj| <= |patch|_2 A_c, A_c = sqrt(sum_j (r_j w_cj)^2) ~ K |w_c|_2 / sqrt 3, with |patch|_2 from the 3 x 3 box sum of the per-pixel
// channel energy.  The look-ahead splits both operands into fp16 hi + lo (relative error <= 2^-22 each, the dropped lo x lo term 2^-22 more:
// exact statements about the formats) and accumulates with 3 MFMA instructions per 16-term k-step in the same chain order, each
// rounding (and aligning) once at <= 2^-23 of the running magnitude: term j is carried by 3 (K - j) / 16 + 3 of them.  ReLU is 1-Lipschitz; the
// bias add rounds once on each side.  The 1 x 1 head is evaluated by the SAME exact kernel on both hidden vectors:
// |L~ - L| <= sum_c |v_ac| |h_e - h_f| + 2 g_h sum_c |v_ac| max(|h_e|, |h_f|), g_h = 256 u / (1 - 256 u), |h| <= |patch|_2 |w_c|_2 + |b_c|.
// Per anchor a this is  B_a(p) = c1_a |patch(p)|_2 + c0_a  with two constants fixed at model finalize (api.hip, in double; inflated by 2 % for
// the float32 evaluation of the bound itself).
// What is a theorem and what is a model: g_e and g_h are the textbook bounds of the fp32 chains the exact mode IS; the split error of the
// operands is exact arithmetic on the formats; "one rounding of relative size 2^-23 per MFMA instruction" is a MODEL of
// v_mfma_f32_32x32x16_f16's internal adder (tools/mfma_f16_probe.hip: the pipe aligns the 16 products to a common exponent with a finite
// width; measured errors of whole layers stay below 2^-20 S, tests/test_gpu_parity.py::test_conv_f16x3_within_split_precision_of_exact,
// i.e. far inside the modelled term).  Because a model is not a proof, every sweep PUTS THE BOUND TO THE TEST: each selected anchor is evaluated
// both ways, prune_scatter_kernel keeps max |L~ - L| / B over all of them (10 - 35 % of all anchors of the two levels, hundreds of thousands
// per forward), and cald_sweep repeats itself with the dense head if the ratio ever exceeds 1 (observed: 5e-5, cald_profile_prune).
// Round 6: the instruction IS now stated bit for bit (oracle/mfma_f16_model.h, pinned to the hardware on > 10^7 dot products), and the
// constants in api.hip are derived from that statement -- a theorem about the model instead of a guess about the pipe; the all-anchor test
// (tests: test_rpn_pruning_bound_holds_on_every_anchor) evaluates the bound on every anchor of P2 / P3, pruned ones included.
#include "common.h"
#include "kernels.h"
#include "h16.h"
#include <cfloat>

namespace {
// per pixel: sum over the 256 channels of P^2 -- and, on the way, the split-fp16 form of the pixel (h16.h) for the look-ahead conv: with it
// the look-ahead runs on conv_h4 (operands HBM -> LDS by DMA, no split arithmetic in its k-loop) instead of conv_h3's fp32 loader; the
// tensor is being read here anyway.  One wavefront per pixel (float4 per lane), 4 pixels per workgroup.
__global__ __launch_bounds__(256) void prune_energy_kernel(RpnPruneArgs a, int l) {
    const int v = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const LevelSeg sg = a.seg[l][v];
    const int n = sg.H * sg.W;
    const float4* f = reinterpret_cast<const float4*>(a.feat[l] + sg.pix_off * 256ll);
    for (int p = blockIdx.x * 4 + wave; p < n; p += gridDim.x * 4) {
        const float4 x = f[(long long)p * 64 + lane];
        if (a.split[l]) h16_store4(reinterpret_cast<unsigned char*>(a.split[l]) + (sg.pix_off + p) * 1024ll, 4 * lane, x);
        float s = (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, 64);
        if (lane == 0) {
            a.energy[l][sg.pix_off + p] = s;
            // |x| >= 4094 would leave fp16's range after the split's 2^4 scale (the look-ahead's hi half becomes inf) and a non-finite activation
            // voids every bound: either is visible in the pixel's energy (|x| >= 4094 => s >= 4094^2).  Flag it: the sweep then repeats itself with
            // the dense head (api.hip) -- a pruned anchor is never evaluated both ways, so this cannot be left to the check on the selected ones.
            if (!(s < 16760836.0f) && a.check) atomicMax(reinterpret_cast<unsigned*>(a.check) + 1, __float_as_uint(1.0f));
        }
    }
}

// per (level, view): tau by radix select over the lower bounds, the pixel mask, the ordered list of selected pixels.
// grid = (2, V), block = 1024, dynamic LDS = one bit per pixel.
// (the 1.0001 covers the fp32 rounding of the 9 x 256 squares' sum in whatever order: <= 2 304 u = 1.4e-4 relative, half of it after the root)
__device__ __forceinline__ float prune_patch_norm(const float* e, int parts, int y, int x, int H, int W) {
    float s = 0.0f;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const float* q = e + (yy * W + xx) * parts;
                s = s + (parts == 4 ? (q[0] + q[1]) + (q[2] + q[3]) : q[0]);
            }
        }
    return sqrtf(s) * 1.0001f;
}
// MODE 0: the single-stage rule (round 5): select every pixel holding an anchor whose upper bound reaches tau, park the rest.
// MODE 1: stage 0 of the two-stage rule: tau as above (kept in a.tau_key), select the pixels holding an anchor whose LOWER bound reaches tau -- at
//         least k anchors, the ones that are certainly good.  Nothing is parked yet.
// MODE 2: stage 1: the exact logits of stage 0's pixels are known now (a.head_rows[0]); tau' = the k-th largest of them is a threshold that k
//         real anchors reach, so an anchor with upper bound < max(tau, tau') cannot be among the k largest.  Select the remaining pixels whose
//         upper bound reaches it, park everything selected by neither stage.  The band of "maybe" anchors below the threshold is one bound wide
//         instead of two: 7.3 % instead of 9.7 % of P2, 24 % instead of 45 % of P3 recomputed (tools/prune_two_stage_potential.py).
template <int MODE>
__global__ __launch_bounds__(1024) void prune_select_kernel(RpnPruneArgs a) {
    extern __shared__ unsigned mask[];
    __shared__ int hist[256];
    __shared__ unsigned s_prefix, s_mask;
    __shared__ int s_remaining, s_total, s_wave_cnt[16];
    const int l = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
    const LevelSeg sg = a.seg[l][v];
    const int H = sg.H, W = sg.W, npx = H * W, A = 3, n = npx * A;
    const int k = n < a.pre_n ? n : a.pre_n;
    const int ld = a.head_ld;
    const float* head = a.head[l] + sg.pix_off * (long long)ld;
    const float* en = a.energy[l] + sg.pix_off * a.energy_parts;
    float* pnv = a.pnorm[l] + sg.pix_off;           // |patch|_2 per pixel: written once below, re-read by the SAME thread in every pass (and by the scatter kernel)
    constexpr int ST = MODE == 2 ? 1 : 0;           // which stage's row list this launch writes
    int* rmap = a.row_map[ST][l] + sg.pix_off;
    const int words = (npx + 31) >> 5;
    for (int i = tid; i < words; i += 1024) mask[i] = 0u;
    if (tid == 0) { s_prefix = 0u; s_mask = 0u; s_remaining = k; }
    __syncthreads();
    auto bound = [&](int an, float pn) { return a.c1[an] * pn + a.c0[an]; };
    if (MODE != 2) {
        for (int p = tid; p < npx; p += 1024) {
            const float pn = prune_patch_norm(en, a.energy_parts, p / W, p % W, H, W);
            pnv[p] = pn;
            // the range of the split (see prune_energy_kernel): a patch norm below 4094 means every |x| in the patch is; anything else -- a large
            // activation, inf, NaN -- sends the sweep back to the dense head
            if (!(pn < 4094.0f) && a.check) atomicMax(reinterpret_cast<unsigned*>(a.check) + 1, __float_as_uint(1.0f));
        }
    }
    // ---- the k-th largest of: the lower bounds over all anchors (MODE 0, 1) / the exact logits of stage 0's pixels (MODE 2) ----
    const int ns0 = MODE == 2 ? a.nsel[0][l * a.V + v] : 0;
    const float* rows0 = MODE == 2 ? a.head_rows[0][l] + sg.pix_off * (long long)ld : nullptr;
    const bool all_kept = n <= k;
    if (!all_kept) {
        for (int pass = 0; pass < 4; pass++) {                  // 8 bits per pass
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix, msk = s_mask;
            if (MODE == 2) {
                for (int i = tid; i < ns0 * A; i += 1024) {
                    const unsigned key = det_orderable(rows0[(long long)(i / A) * ld + (i % A)]);
                    if ((key & msk) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
                }
            } else {
                for (int p = tid; p < npx; p += 1024) {
                    const float pn = pnv[p];
                    for (int an = 0; an < A; an++) {
                        const unsigned key = det_orderable(head[(long long)p * ld + an] - bound(an, pn));
                        if ((key & msk) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
                    }
                }
            }
            __syncthreads();
            if (tid == 0) {
                int cum = 0, rem = s_remaining;
                for (int b = 255; b >= 0; b--) {
                    const int h = hist[b];
                    if (cum + h >= rem) { s_remaining = rem - cum; s_prefix = prefix | ((unsigned)b << shift); s_mask = msk | (255u << shift); break; }
                    cum += h;
                }
            }
            __syncthreads();
        }
    }
    unsigned tau = all_kept ? 0u : s_prefix;                    // orderable key (0: every anchor stays)
    unsigned tau_lb = tau;
    if (MODE == 1 && tid == 0) a.tau_key[l * a.V + v] = tau;
    if (MODE == 2) {
        tau_lb = a.tau_key[l * a.V + v];                        // stage 0's threshold on the lower bounds
        // fewer than k exact logits can only mean stage 0 kept everything (all_kept) -- otherwise its pixels hold >= k anchors
        tau = (ns0 * A >= k && tau > tau_lb) ? tau : tau_lb;
    }
    for (int p = tid; p < npx; p += 1024) {
        const float pn = pnv[p];
        bool keep = false, first = false;
        for (int an = 0; an < A; an++) {
            const float lg = head[(long long)p * ld + an];
            const float B = bound(an, pn);
            const float ub = lg + B, lb = lg - B;
            if (MODE == 0) keep = keep || !(ub == ub) || det_orderable(ub) >= tau;             // NaN: never pruned
            if (MODE == 1) keep = keep || !(lb == lb) || det_orderable(lb) >= tau;
            if (MODE == 2) { first = first || !(lb == lb) || det_orderable(lb) >= tau_lb; keep = keep || !(ub == ub) || det_orderable(ub) >= tau; }
        }
        if (MODE == 2) {
            if (first) keep = false;                                                            // already exact since stage 0
            else if (!keep) { float* hw = a.head_out[l] + (sg.pix_off + p) * (long long)ld; hw[0] = -FLT_MAX; hw[1] = -FLT_MAX; hw[2] = -FLT_MAX; }
        }
        if (keep) atomicOr(&mask[p >> 5], 1u << (p & 31));
    }
    __syncthreads();
    // ordered compaction of the selected pixels (ascending pixel index: neighbouring rows of the gathered conv share input pixels)
    const int lane = tid & 63, wave = tid >> 6;
    int base = 0;
    for (int w0 = 0; w0 < words; w0 += 1024) {
        const int wi = w0 + tid;
        const unsigned bits = wi < words ? mask[wi] : 0u;
        const int cnt = __popc(bits);
        int inc = cnt;                                           // inclusive scan over the wavefront, then over the 16 wavefronts
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
        if (lane == 63) s_wave_cnt[wave] = inc;
        __syncthreads();
        int wbase = 0;
        for (int q = 0; q < wave; q++) wbase += s_wave_cnt[q];
        int o = base + wbase + inc - cnt;
        unsigned b = bits;
        while (b) { const int t = __ffs(b) - 1; b &= b - 1; rmap[o++] = wi * 32 + t; }
        if (tid == 1023) s_total = wbase + inc;
        __syncthreads();
        base += s_total;
        __syncthreads();
    }
    if (tid == 0) {
        a.nsel[ST][l * a.V + v] = base;
        if (a.stat) { atomicAdd(a.stat + 2 * l, (unsigned long long)base); if (MODE != 2) atomicAdd(a.stat + 2 * l + 1, (unsigned long long)npx); }
        if (a.log[ST]) { atomicAdd(a.log[ST] + 2 * l, (unsigned long long)base); atomicAdd(a.log[ST] + 2 * l + 1, (unsigned long long)npx); }
    }
    if (MODE == 0) {
        // unselected pixels: their three logits can never reach the top-k -- park them below every real logit
        float* headw = a.head_out[l] + sg.pix_off * (long long)ld;
        for (int p = tid; p < npx; p += 1024)
            if (!((mask[p >> 5] >> (p & 31)) & 1u)) { headw[(long long)p * ld] = -FLT_MAX; headw[(long long)p * ld + 1] = -FLT_MAX; headw[(long long)p * ld + 2] = -FLT_MAX; }
    }
}

// the exact head rows of the selected pixels back into the dense [pixel][head_ld] map -- and the bound put to the test: every selected anchor
// has both values, the look-ahead's L~ (still in the map) and the exact L; max |L~ - L| / B over all of them goes to a.check[0]
// (non-negative floats order like their bit patterns).  A ratio above 1 means the bound does not hold on this data: the sweep then repeats
// itself with the dense head (api.hip).  grid = (blocks, V, 2 * stages): z = level + 2 * stage
__global__ __launch_bounds__(256) void prune_scatter_kernel(RpnPruneArgs a) {
    const int l = blockIdx.z & 1, st = blockIdx.z >> 1, v = blockIdx.y;
    const LevelSeg sg = a.seg[l][v];
    const int ns = a.nsel[st][l * a.V + v], ld = a.head_ld;
    const int* rmap = a.row_map[st][l] + sg.pix_off;
    const float* src = a.head_rows[st][l] + sg.pix_off * (long long)ld;
    const float* pnv = a.pnorm[l] + sg.pix_off;
    float* dst = a.head_out[l] + sg.pix_off * (long long)ld;
    float worst = 0.0f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ns * ld; i += gridDim.x * 256) {
        const int r = i / ld, c = i - r * ld;
        const int p = rmap[r];
        const float exact = src[i];
        if (c < 3) {
            const float approx = dst[(long long)p * ld + c];
            const float B = a.c1[c] * pnv[p] + a.c0[c];
            const float ratio = fabsf(approx - exact) / B;
            worst = (ratio == ratio) ? fmaxf(worst, ratio) : INFINITY;
        }
        dst[(long long)p * ld + c] = exact;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) worst = fmaxf(worst, __shfl_xor(worst, off, 64));      // one atomic per wavefront, not per element
    if ((threadIdx.x & 63) == 0 && worst > 0.0f && a.check) atomicMax(reinterpret_cast<unsigned*>(a.check), __float_as_uint(worst));
}
}   // namespace

void launch_rpn_prune_energy(const RpnPruneArgs& a, hipStream_t st) {
    for (int l = 0; l < 2; l++) hipLaunchKernelGGL(prune_energy_kernel, dim3(256, a.V), dim3(256), 0, st, a, l);
}
void launch_rpn_prune_select(const RpnPruneArgs& a, int max_pix, int stage, hipStream_t st) {
    const size_t lds = (size_t)((max_pix + 31) / 32) * 4;
    static PerDeviceOnce once0, once1, once2;
    if (a.stages == 1) { allow_big_lds(once0, prune_select_kernel<0>); hipLaunchKernelGGL(prune_select_kernel<0>, dim3(2, a.V), dim3(1024), lds, st, a); }
    else if (stage == 0) { allow_big_lds(once1, prune_select_kernel<1>); hipLaunchKernelGGL(prune_select_kernel<1>, dim3(2, a.V), dim3(1024), lds, st, a); }
    else { allow_big_lds(once2, prune_select_kernel<2>); hipLaunchKernelGGL(prune_select_kernel<2>, dim3(2, a.V), dim3(1024), lds, st, a); }
}
void launch_rpn_prune_scatter(const RpnPruneArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(prune_scatter_kernel, dim3(64, a.V, 2 * a.stages), dim3(256), 0, st, a);
}
