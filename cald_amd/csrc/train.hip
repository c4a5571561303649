// train.hip -- the training step of the detector (cald_train.py:40-74 train_one_epoch ->
// task_model(images, targets) / losses.backward() / optimizer.step(); the arithmetic the reference delegates to
// torchvision 0.8.2 + cuDNN/cuBLAS autograd).
//
// Operator-level C ABI on device pointers (the host side that strings them into the Faster R-CNN training graph is
// cald_amd/train.py).  Activations are dense NHWC batches [N][H][W][C] (torchvision pads a training batch to one common
// size, GeneralizedRCNNTransform.batch_images).
//
//   forward conv / linear      the inference kernels (conv_p4.hip / conv_mfma.hip) on weights packed ON THE DEVICE from the
//                              torch-layout parameter tensor every step (pack_weight_kernel)
//   data gradient              the same kernels on the spatially flipped, channel-transposed filter (mode 1 of the packer);
//                              stride-2 layers first scatter dY onto the stride-1 grid (dilate_kernel)
//   weight gradient            wgrad_kernel: GEMM  dW[co][(tap, ci)] = sum over output pixels  dY[q][co] * X[q @ tap][ci]
//                              on v_mfma_f32_32x32x2_f32; both operands are read in their natural NHWC order (the channel
//                              index is the MFMA row / column, the pixel index is k: no transposes), staged through LDS,
//                              pixels split over workgroups, partial tiles summed in a fixed order (deterministic)
//   RoIAlign forward/backward  roi_align_train_kernel / roi_align_bwd_kernel (bilinear weights scattered with atomics)
//   losses                     softmax cross-entropy, smooth-L1, binary cross-entropy with logits: value + gradient in one pass
//   optimizer                  sgd_kernel: torch.optim.SGD (weight decay, momentum, no dampening / nesterov) over the flat
//                              parameter buffer
#include "common.h"
#include <type_traits>
#include "kernels.h"
#include "sortnms.h"
#include "../../include/cald_hip.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

int cald_internal_fail(int code, const char* fmt, ...);
hipStream_t cald_internal_stream(cald_ctx* c);
int cald_internal_device(cald_ctx* c);
const float* cald_internal_zeros(cald_ctx* c);
int cald_internal_scratch(cald_ctx* c, size_t bytes, void** out);   // grow-only per-context device scratch (stream-ordered reuse)

#define TFAIL(code, ...) return cald_internal_fail(code, __VA_ARGS__)
#define THIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return cald_internal_fail(CALD_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------------
// dense-batch geometry: N equal views of H x W, cached on the device per (context, N, H, W)
// ---------------------------------------------------------------------------------------------------------------------
static std::mutex g_seg_mu;
static std::map<std::tuple<cald_ctx*, int, int, int>, LevelSeg*> g_seg;
static bool g_seg_evict = false;      // the cache went over its bound: empty it at the NEXT entry point, never inside one
static size_t seg_cache_cap() {       // CALD_SEG_CACHE_CAP: tests lower it to exercise the eviction
    static const size_t cap = [] { const char* e = getenv("CALD_SEG_CACHE_CAP"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 ? v : 1024); }();
    return cap;
}
// A public entry point takes several tables from consecutive dense_seg() calls (input / output / up-sampling geometry, s0 plus five
// pyramid levels, ...) before it launches anything, so a table must never be freed while an entry point is running: dense_seg() only
// flags the overflow (the cache then exceeds its bound by the few tables of one call), and the NEXT entry point that uses tables
// empties the cache first thing -- after the launches that may still read the old tables have drained; a rare, synchronous event.
static int seg_entry() {
    std::lock_guard<std::mutex> lk(g_seg_mu);
    if (!g_seg_evict) return 0;
    THIP(hipDeviceSynchronize());
    for (auto& kv : g_seg) hipFree(kv.second);
    g_seg.clear();
    g_seg_evict = false;
    return 0;
}
#define SEG_ENTRY() do { const int rc_ = seg_entry(); if (rc_) return rc_; } while (0)
static int dense_seg(cald_ctx* c, int N, int H, int W, const LevelSeg** out) {
    std::lock_guard<std::mutex> lk(g_seg_mu);
    auto key = std::make_tuple(c, N, H, W);
    auto it = g_seg.find(key);
    if (it == g_seg.end()) {
        if (g_seg.size() >= seg_cache_cap()) g_seg_evict = true;   // a long run over variable padded sizes x pyramid levels x contexts
        std::vector<LevelSeg> h(N + 1);
        const int tiles = (H * W + 127) / 128;
        for (int v = 0; v <= N; v++) { h[v].pix_off = (long long)v * H * W; h[v].H = H; h[v].W = W; h[v].tile_start = v * tiles; h[v].pad_ = 0; }
        LevelSeg* d = nullptr;
        THIP(hipMalloc((void**)&d, sizeof(LevelSeg) * (N + 1)));
        THIP(hipMemcpy(d, h.data(), sizeof(LevelSeg) * (N + 1), hipMemcpyHostToDevice));
        it = g_seg.emplace(key, d).first;
    }
    *out = it->second;
    return 0;
}

// Small host tables on their way to the device without stopping the host: a ring of pinned slots + device slots per context.  A slot
// is reused only after the stream has passed the event recorded behind its last consumer (in practice never waited for).
struct StageRing { char* host; char* dev; hipEvent_t ev[8]; bool used[8]; int next; };
static const size_t kStageSlot = sizeof(ViewDesc) * CALD_MAX_VIEWS;
static std::map<cald_ctx*, StageRing> g_stage;
static int stage_upload(cald_ctx* c, const void* src, size_t bytes, hipStream_t st, const void** dev_out, int* slot_out) {
    if (bytes > kStageSlot) TFAIL(CALD_ERR_INVALID, "staging slot too small");
    std::lock_guard<std::mutex> lk(g_seg_mu);
    auto it = g_stage.find(c);
    if (it == g_stage.end()) {
        StageRing r; memset(&r, 0, sizeof(r));
        THIP(hipHostMalloc((void**)&r.host, kStageSlot * 8, hipHostMallocDefault));
        THIP(hipMalloc((void**)&r.dev, kStageSlot * 8));
        for (int i = 0; i < 8; i++) THIP(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming));
        it = g_stage.emplace(c, r).first;
    }
    StageRing& r = it->second;
    const int s = r.next; r.next = (s + 1) & 7;
    if (r.used[s]) THIP(hipEventSynchronize(r.ev[s]));
    memcpy(r.host + kStageSlot * s, src, bytes);
    THIP(hipMemcpyAsync(r.dev + kStageSlot * s, r.host + kStageSlot * s, bytes, hipMemcpyHostToDevice, st));
    *dev_out = r.dev + kStageSlot * s; *slot_out = s;
    return 0;
}
static int stage_consumed(cald_ctx* c, int slot, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_seg_mu);
    StageRing& r = g_stage[c];
    THIP(hipEventRecord(r.ev[slot], st));
    r.used[slot] = true;
    return 0;
}

extern "C" int cald_train_seg_cache_size(void) { std::lock_guard<std::mutex> lk(g_seg_mu); return (int)g_seg.size(); }
void cald_internal_train_release(cald_ctx* c) {
    std::lock_guard<std::mutex> lk(g_seg_mu);
    for (auto it = g_seg.begin(); it != g_seg.end();) {
        if (std::get<0>(it->first) == c) { hipFree(it->second); it = g_seg.erase(it); } else ++it;
    }
    auto sr = g_stage.find(c);
    if (sr != g_stage.end()) {
        for (int i = 0; i < 8; i++) hipEventDestroy(sr->second.ev[i]);
        hipHostFree(sr->second.host); hipFree(sr->second.dev);
        g_stage.erase(sr);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight packing on the device
// ---------------------------------------------------------------------------------------------------------------------
// mode 0: forward      K rows = (tap, ci) of a Cin_k-channel input (Cin_k >= Cin, the input buffer's channel stride), N = Cout
// mode 1: data grad    K rows = (flipped tap, co) of a Cin_k-channel dY (Cin_k >= Cout),                      N = Cin
// mode 2: forward of a linear layer whose torch weight is [Cout][Cin][taps] (fc6: [1024][256][7*7]) applied to rows laid out
//         [tap][Cin] (the RoIAlign output [R][49][256]): K rows = tap * Cin + ci
// mode 3: data gradient of a mode-2 layer: K rows = co of a CinK-channel dY, N = tap * Cin + ci
struct PackArgs { const float* w; float* wk; float* w4; int Cout, Cin, taps, CinK, Kpad, NPad, mode; const float* rowscale; };
__device__ __forceinline__ void pack_weight_body(const PackArgs& a, long long blk) {
    const long long i = blk * 256 + threadIdx.x;
    if (i >= (long long)a.Kpad * a.NPad) return;
    const int k = (int)(i / a.NPad), n = (int)(i - (long long)k * a.NPad);
    int tap, ci;
    if (a.mode == 2) { tap = k / a.Cin; ci = k - tap * a.Cin; }
    else if (a.mode == 3) { tap = 0; ci = k; }
    else if (a.CinK % 16 == 0 && a.taps <= 32) { const int chunk = k / (16 * a.taps), rem = k - chunk * 16 * a.taps; tap = rem >> 4; ci = chunk * 16 + (rem & 15); }
    else { tap = k / a.CinK; ci = k - tap * a.CinK; }
    float v = 0.0f;
    if (a.mode == 1) {
        if (tap < a.taps && ci < a.Cout && n < a.Cin) v = a.w[((long long)ci * a.Cin + n) * a.taps + (a.taps - 1 - tap)];
        if (a.rowscale && ci < a.Cout) v = v * a.rowscale[ci];       // FrozenBatchNorm scale of the forward layer's output channel
    } else if (a.mode == 3) {
        const int t = n / a.Cin, cc = n - t * a.Cin;
        if (ci < a.Cout && t < a.taps) v = a.w[((long long)ci * a.Cin + cc) * a.taps + t];
        if (a.rowscale && ci < a.Cout) v = v * a.rowscale[ci];
    } else {
        if (tap < a.taps && ci < a.Cin && n < a.Cout) v = a.w[((long long)n * a.Cin + ci) * a.taps + tap];
    }
    a.wk[i] = v;
    const int kt = k >> 4, kk = k & 15, kq = kk >> 3, j = (kk & 7) >> 1, h = kk & 1;
    a.w4[(((long long)(kt * 2 + kq) * a.NPad + n) * 2 + h) * 4 + j] = v;
}
__global__ __launch_bounds__(256) void pack_weight_kernel(PackArgs a) { pack_weight_body(a, blockIdx.x); }
// Forward packs (modes 0, 2) in two coalesced passes: (1) every torch row n (Cin * taps contiguous floats) is read in order and
// written to T[n][k] (k = position in the chain order: neighbours stay inside one 16-channel chunk); (2) a tiled LDS transpose
// T[n][k] -> wk[k][n] + the conv_p4 layout.  (pack_weight_kernel reads with a stride of one whole row between lanes.)
__device__ __forceinline__ void pack_rows_body(const PackArgs& a, float* T, long long blk) {
    const long long i = blk * 256 + threadIdx.x;
    const long long row = (long long)a.Cin * a.taps;
    if (i >= (long long)a.Cout * row) return;
    const int n = (int)(i / row); const int r = (int)(i - (long long)n * row);
    const int ci = r / a.taps, tap = r - ci * a.taps;
    int k;
    if (a.mode == 2) k = tap * a.Cin + ci;
    else k = (a.CinK % 16 == 0 && a.taps <= 32) ? ((ci >> 4) * a.taps + tap) * 16 + (ci & 15) : tap * a.CinK + ci;
    T[(long long)n * a.Kpad + k] = a.w[i];
}
__global__ __launch_bounds__(256) void pack_rows_kernel(PackArgs a, float* T) { pack_rows_body(a, T, blockIdx.x); }
__device__ __forceinline__ void pack_transpose_body(const PackArgs& a, const float* T, int bx, int by) {
    __shared__ float tile[32][33];
    const int k0 = bx * 32, n0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        tile[r][tx] = (n < a.Cout && k < a.Kpad) ? T[(long long)n * a.Kpad + k] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, n = n0 + tx;
        if (k >= a.Kpad || n >= a.NPad) continue;
        const float v = tile[tx][r];
        a.wk[(long long)k * a.NPad + n] = v;
        const int kt = k >> 4, kk = k & 15, kq = kk >> 3, j = (kk & 7) >> 1, h = kk & 1;
        a.w4[(((long long)(kt * 2 + kq) * a.NPad + n) * 2 + h) * 4 + j] = v;
    }
}
__global__ __launch_bounds__(256) void pack_transpose_kernel(PackArgs a, const float* T) { pack_transpose_body(a, T, blockIdx.x, blockIdx.y); }
__device__ __forceinline__ void pack_vec_body(const float* bias, const float* scale, const float* shift, int n, float* dst, int npad, int blk) {
    const int i = blk * 256 + threadIdx.x;
    if (i >= npad) return;
    dst[i] = (bias && i < n) ? bias[i] : 0.0f;
    dst[npad + i] = (scale && i < n) ? scale[i] : 0.0f;
    dst[2 * npad + i] = (shift && i < n) ? shift[i] : 0.0f;
}
__global__ __launch_bounds__(256) void pack_vec_kernel(const float* bias, const float* scale, const float* shift, int n, float* dst, int npad) {
    pack_vec_body(bias, scale, shift, n, dst, npad, blockIdx.x);
}
// Every trainable layer's packs in TWO launches (cald_train_pack_plan_*): the optimizer changes all weights at once, and ~220 launches of
// 2 - 9 us each kept the side stream busy for 1.5 ms at the start of a step -- longer than the frozen stem + layer 1 the main stream
// runs meanwhile.  A launch covers a list of segments (kind, job, first block); a workgroup finds its segment by bisection.
struct PackJobDev { PackArgs a; float* T; const float *bias, *scale, *shift; float* vec; int n_true, gx; };
struct PackSeg { int kind, job; unsigned first; };          // kind 0 rows (modes 0 / 2), 1 whole pack (modes 1 / 3), 2 epilogue vectors, 3 transpose
__global__ __launch_bounds__(256) void pack_plan_kernel(const PackSeg* __restrict__ segs, int nseg, const PackJobDev* __restrict__ jobs) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segs[mid].first <= blockIdx.x) lo = mid; else hi = mid - 1; }
    const PackSeg sg = segs[lo];
    const PackJobDev& j = jobs[sg.job];
    const unsigned blk = blockIdx.x - sg.first;
    if (sg.kind == 0) pack_rows_body(j.a, j.T, blk);
    else if (sg.kind == 1) pack_weight_body(j.a, blk);
    else if (sg.kind == 2) pack_vec_body(j.bias, j.scale, j.shift, j.n_true, j.vec, j.a.NPad, (int)blk);
    else pack_transpose_body(j.a, j.T, (int)(blk % (unsigned)j.gx), (int)(blk / (unsigned)j.gx));
}

struct PackGeom { int K, Kpad, NPad, n_true, cin_conv; long long floats; };
static PackGeom pack_geom(int Cout, int Cin, int KH, int KW, int CinK, int mode) {
    PackGeom g;
    const int taps = KH * KW;
    if (mode == 2) { g.K = taps * Cin; g.n_true = Cout; g.cin_conv = taps * Cin; }
    else if (mode == 3) { g.K = CinK; g.n_true = taps * Cin; g.cin_conv = CinK; }
    else { g.K = taps * CinK; g.n_true = mode == 1 ? Cin : Cout; g.cin_conv = CinK; }
    g.Kpad = round_up(g.K, 16);
    g.NPad = cout_pad(g.n_true);
    g.floats = 2ll * g.Kpad * g.NPad + 3ll * g.NPad;
    return g;
}
extern "C" int cald_train_packed_floats(int Cout, int Cin, int KH, int KW, int CinK, int mode, int64_t* floats_out) {
    if (!floats_out || Cout < 1 || Cin < 1 || KH < 1 || KW < 1 || mode < 0 || mode > 3) TFAIL(CALD_ERR_INVALID, "bad arguments");
    *floats_out = pack_geom(Cout, Cin, KH, KW, CinK, mode).floats;
    return 0;
}
extern "C" int cald_train_pack_conv(cald_ctx* c, const float* w, const float* bias, const float* scale, const float* shift,
                                    int Cout, int Cin, int KH, int KW, int CinK, int mode, float* packed) {
    if (!c || !w || !packed) TFAIL(CALD_ERR_INVALID, "null argument");
    if (mode < 0 || mode > 3) TFAIL(CALD_ERR_INVALID, "mode must be 0 (forward), 1 (data gradient), 2 (tap-major linear) or 3 (its data gradient)");
    if (mode != 2 && (CinK % 4 || CinK < (mode == 0 ? Cin : Cout))) TFAIL(CALD_ERR_INVALID, "CinK must be a multiple of 4 and cover the contracted channels");
    THIP(hipSetDevice(cald_internal_device(c)));
    const PackGeom g = pack_geom(Cout, Cin, KH, KW, CinK, mode);
    PackArgs a{w, packed, packed + (long long)g.Kpad * g.NPad, Cout, Cin, KH * KW, CinK, g.Kpad, g.NPad, mode, (mode == 1 || mode == 3) ? scale : nullptr};
    hipStream_t st = cald_internal_stream(c);
    const long long n = (long long)g.Kpad * g.NPad;
    if (mode == 0 || mode == 2) {
        void* scratch = nullptr;
        const long long tn = (long long)Cout * g.Kpad;
        if (int rc = cald_internal_scratch(c, (size_t)tn * 4, &scratch)) return rc;
        float* T = (float*)scratch;
        if (g.Kpad != Cin * KH * KW)     // k positions no source element maps to (channel / K padding)
            THIP(hipMemsetAsync(T, 0, (size_t)tn * 4, st));
        const long long src = (long long)Cout * Cin * KH * KW;
        hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((src + 255) / 256)), dim3(256), 0, st, a, T);
        hipLaunchKernelGGL(pack_transpose_kernel, dim3((g.Kpad + 31) / 32, (g.NPad + 31) / 32), dim3(256), 0, st, a, (const float*)T);
    } else {
        hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    }
    float* vec = packed + 2 * n;
    const int nt = g.n_true, blocks = (g.NPad + 255) / 256;
    if (mode == 0 || mode == 2)     // the data-gradient packs carry no epilogue vectors (cald_train_conv never reads them: flags 0)
        hipLaunchKernelGGL(pack_vec_kernel, dim3(blocks), dim3(256), 0, st, bias, scale, shift, nt, vec, g.NPad);
    THIP(hipGetLastError());
    return 0;
}

struct cald_pack_plan {
    int device; PackSeg *seg1, *seg2; PackJobDev* jobs; int nseg1, nseg2; unsigned blocks1, blocks2;
};
static long long plan_round64(long long x) { return (x + 63) / 64 * 64; }
static int pack_plan_check(int n, const cald_pack_job* jobs) {
    if (n < 1 || !jobs) TFAIL(CALD_ERR_INVALID, "no jobs");
    for (int i = 0; i < n; i++) {
        const cald_pack_job& q = jobs[i];
        if (!q.weight || !q.packed) TFAIL(CALD_ERR_INVALID, "job %d: null weight / packed", i);
        if (q.Cout < 1 || q.Cin < 1 || q.KH < 1 || q.KW < 1 || q.mode < 0 || q.mode > 3) TFAIL(CALD_ERR_INVALID, "job %d: bad shape / mode", i);
        if (q.mode != 2 && (q.CinK % 4 || q.CinK < (q.mode == 0 ? q.Cin : q.Cout))) TFAIL(CALD_ERR_INVALID, "job %d: CinK must be a multiple of 4 and cover the contracted channels", i);
    }
    return 0;
}
extern "C" int cald_train_pack_plan_scratch_floats(int n, const cald_pack_job* jobs, int64_t* floats_out) {
    if (!floats_out) TFAIL(CALD_ERR_INVALID, "null argument");
    if (int rc = pack_plan_check(n, jobs)) return rc;
    long long tot = 0;
    for (int i = 0; i < n; i++) {
        const cald_pack_job& q = jobs[i];
        if (q.mode == 0 || q.mode == 2) tot += plan_round64((long long)q.Cout * pack_geom(q.Cout, q.Cin, q.KH, q.KW, q.CinK, q.mode).Kpad);
    }
    *floats_out = tot;
    return 0;
}
extern "C" int cald_train_pack_plan_create(cald_ctx* c, int n, const cald_pack_job* jobs, float* scratch, int64_t scratch_floats, cald_pack_plan** out) {
    if (!c || !out) TFAIL(CALD_ERR_INVALID, "null argument");
    int64_t need = 0;
    if (int rc = cald_train_pack_plan_scratch_floats(n, jobs, &need)) return rc;
    if (need && (!scratch || scratch_floats < need)) TFAIL(CALD_ERR_INVALID, "scratch of %lld floats needed", (long long)need);
    THIP(hipSetDevice(cald_internal_device(c)));
    std::vector<PackJobDev> dj(n);
    std::vector<PackSeg> s1, s2;
    unsigned b1 = 0, b2 = 0;
    long long toff = 0;
    for (int i = 0; i < n; i++) {
        const cald_pack_job& q = jobs[i];
        const PackGeom g = pack_geom(q.Cout, q.Cin, q.KH, q.KW, q.CinK, q.mode);
        const long long nn = (long long)g.Kpad * g.NPad;
        const bool fwd = q.mode == 0 || q.mode == 2;
        PackJobDev& d = dj[i];
        d.a = PackArgs{q.weight, q.packed, q.packed + nn, q.Cout, q.Cin, q.KH * q.KW, q.CinK, g.Kpad, g.NPad, q.mode, fwd ? nullptr : q.bn_scale};
        d.T = nullptr; d.bias = q.bias; d.scale = q.bn_scale; d.shift = q.bn_shift; d.vec = q.packed + 2 * nn; d.n_true = g.n_true;
        d.gx = (g.Kpad + 31) / 32;
        if (fwd) {
            d.T = scratch + toff; toff += plan_round64((long long)q.Cout * g.Kpad);
            const long long src = (long long)q.Cout * q.Cin * q.KH * q.KW;
            s1.push_back(PackSeg{0, i, b1}); b1 += (unsigned)((src + 255) / 256);
            s1.push_back(PackSeg{2, i, b1}); b1 += (unsigned)((g.NPad + 255) / 256);
            s2.push_back(PackSeg{3, i, b2}); b2 += (unsigned)(d.gx * ((g.NPad + 31) / 32));
        } else {
            s1.push_back(PackSeg{1, i, b1}); b1 += (unsigned)((nn + 255) / 256);
        }
    }
    cald_pack_plan* p = new cald_pack_plan();
    p->device = cald_internal_device(c); p->seg1 = p->seg2 = nullptr; p->jobs = nullptr;
    p->nseg1 = (int)s1.size(); p->nseg2 = (int)s2.size(); p->blocks1 = b1; p->blocks2 = b2;
    auto fail = [&](hipError_t e) { hipFree(p->seg1); hipFree(p->seg2); hipFree(p->jobs); delete p; return cald_internal_fail(CALD_ERR_HIP, "pack plan: %s", hipGetErrorString(e)); };
    hipError_t e;
    if ((e = hipMalloc((void**)&p->jobs, sizeof(PackJobDev) * n)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&p->seg1, sizeof(PackSeg) * (s1.size() + 1))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&p->seg2, sizeof(PackSeg) * (s2.size() + 1))) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(p->jobs, dj.data(), sizeof(PackJobDev) * n, hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(p->seg1, s1.data(), sizeof(PackSeg) * s1.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if (!s2.empty() && (e = hipMemcpy(p->seg2, s2.data(), sizeof(PackSeg) * s2.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    // k positions no source element maps to (channel / K padding) stay zero for the plan's lifetime: the scratch is the plan's own
    // (on the context stream: ordered after whatever used this memory before and before the first cald_train_pack_plan_run)
    if (need && (e = hipMemsetAsync(scratch, 0, (size_t)need * 4, cald_internal_stream(c))) != hipSuccess) return fail(e);
    *out = p;
    return 0;
}
extern "C" int cald_train_pack_plan_run(cald_ctx* c, const cald_pack_plan* p) {
    if (!c || !p) TFAIL(CALD_ERR_INVALID, "null argument");
    if (p->device != cald_internal_device(c)) TFAIL(CALD_ERR_INVALID, "plan belongs to another device");
    THIP(hipSetDevice(p->device));
    hipStream_t st = cald_internal_stream(c);
    hipLaunchKernelGGL(pack_plan_kernel, dim3(p->blocks1), dim3(256), 0, st, (const PackSeg*)p->seg1, p->nseg1, (const PackJobDev*)p->jobs);
    if (p->blocks2) hipLaunchKernelGGL(pack_plan_kernel, dim3(p->blocks2), dim3(256), 0, st, (const PackSeg*)p->seg2, p->nseg2, (const PackJobDev*)p->jobs);
    THIP(hipGetLastError());
    return 0;
}
extern "C" int cald_train_pack_plan_destroy(cald_pack_plan* p) {
    if (!p) return 0;
    hipSetDevice(p->device);
    hipFree(p->seg1); hipFree(p->seg2); hipFree(p->jobs);
    delete p;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward conv / linear / data gradient: the inference kernels on a dense batch
// ---------------------------------------------------------------------------------------------------------------------
// in  [N][H][W][CinK];  out [N][Ho][Wo][out_ld] (channels >= Cout of a row are not written);  residual: same geometry as out
// with row stride Cout... (out_ld == Cout required when residual / up is given);  up: [N][Hup][Wup][Cout] nearest-upsampled and added.
// flags: bit 0 bias, bit 1 scale/shift (FrozenBatchNorm), bit 2 ReLU.  mode as in cald_train_pack_conv (the packed buffer's).
__global__ void relu_bwd_kernel(float* g, const float* act, const float* scale, long long n4, int C4);
static bool p4_takes(const PackGeom& g, int taps) { return g.NPad % 64 == 0 && g.cin_conv % 16 == 0 && taps <= 32; }
extern "C" int cald_train_conv(cald_ctx* c, int N, int H, int W, const float* in, int CinK, const float* packed, int Cout, int Cin,
                               int KH, int KW, int stride, int pad, int mode, int flags, const float* residual, const float* up,
                               int Hup, int Wup, const float* mask, float* out, int out_ld) {
    if (!c || !in || !packed || !out) TFAIL(CALD_ERR_INVALID, "null argument");
    if (N < 1 || H < 1 || W < 1 || stride < 1) TFAIL(CALD_ERR_INVALID, "bad geometry");
    THIP(hipSetDevice(cald_internal_device(c)));
    SEG_ENTRY();
    const PackGeom g = pack_geom(Cout, Cin, KH, KW, CinK, mode);
    const int kh = mode >= 2 ? 1 : KH, kw = mode >= 2 ? 1 : KW;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho < 1 || Wo < 1) TFAIL(CALD_ERR_INVALID, "empty output");
    if ((residual || up) && out_ld != g.n_true) TFAIL(CALD_ERR_INVALID, "residual / upsample-add need out_ld == Cout");
    if (out_ld < g.n_true) TFAIL(CALD_ERR_INVALID, "out_ld < Cout");
    const LevelSeg *si, *so, *su = nullptr;
    if (int rc = dense_seg(c, N, H, W, &si)) return rc;
    if (int rc = dense_seg(c, N, Ho, Wo, &so)) return rc;
    if (up) { if (int rc = dense_seg(c, N, Hup, Wup, &su)) return rc; }
    const long long n = (long long)g.Kpad * g.NPad;
    ConvArgs a; memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.w = packed; a.w4 = packed + n;
    const float* vec = packed + 2 * n;
    a.bias = (flags & 1) ? vec : nullptr; a.scale = (flags & 2) ? vec + g.NPad : nullptr; a.shift = (flags & 2) ? vec + 2 * g.NPad : nullptr;
    a.residual = residual; a.up = up; a.seg_in = si; a.seg_out = so; a.seg_up = up ? su : so;
    a.V = N; a.Cin = g.cin_conv; a.Cout = g.n_true; a.CoutPad = g.NPad; a.Kpad = g.Kpad; a.KH = kh; a.KW = kw; a.stride = stride; a.pad = pad;
    a.relu = (flags & 4) ? 1 : 0; a.total_mtiles = N * ((Ho * Wo + 127) / 128); a.out_ld = out_ld; a.zeros = cald_internal_zeros(c);
    // mask (ReLU backward of the layer the result flows into): in the epilogue of the tiled kernel where it covers the shape, else a pass
    const bool fused = mask && p4_takes(g, kh * kw);
    if (mask && !fused && (out_ld != g.n_true || out_ld % 4)) TFAIL(CALD_ERR_INVALID, "mask on this shape needs a dense output with C %% 4 == 0");
    a.mask = fused ? mask : nullptr;
    launch_conv(a, cald_internal_stream(c));
    if (mask && !fused) {
        const long long n4 = (long long)N * Ho * Wo * out_ld / 4;
        hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, cald_internal_stream(c), out, mask, (const float*)nullptr, n4, out_ld / 4);
    }
    THIP(hipGetLastError());
    return 0;
}

// The same layer shape applied to several tensors in ONE launch (the pyramid levels under shared-weight heads; problems may carry
// different weights of equal shape, e.g. RetinaNet's two towers): workgroups of the small levels fill the tail of the large ones.
// Falls back to one launch per problem when the shape does not qualify for the grouped kernel (conv_p4.hip launch_conv_p4_group).
extern "C" int cald_train_conv_group(cald_ctx* c, int n, int N, const int* hw, const float* const* ins, int CinK, const float* const* packed,
                                     int Cout, int Cin, int KH, int KW, int stride, int pad, int mode, int flags, const float* const* masks,
                                     float* const* outs, int out_ld) {
    if (!c || !hw || !ins || !packed || !outs) TFAIL(CALD_ERR_INVALID, "null argument");
    if (n < 1 || n > CALD_MAX_GROUP) TFAIL(CALD_ERR_INVALID, "1..%d problems per group", CALD_MAX_GROUP);
    THIP(hipSetDevice(cald_internal_device(c)));
    SEG_ENTRY();
    const PackGeom g = pack_geom(Cout, Cin, KH, KW, CinK, mode);
    const int kh = mode >= 2 ? 1 : KH, kw = mode >= 2 ? 1 : KW;
    if (out_ld < g.n_true) TFAIL(CALD_ERR_INVALID, "out_ld < Cout");
    if (masks && !p4_takes(g, kh * kw)) TFAIL(CALD_ERR_INVALID, "masks need a shape the tiled kernel covers");
    ConvArgs probs[CALD_MAX_GROUP];
    const long long nw = (long long)g.Kpad * g.NPad;
    for (int i = 0; i < n; i++) {
        const int H = hw[2 * i], W = hw[2 * i + 1];
        const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
        if (H < 1 || W < 1 || Ho < 1 || Wo < 1 || !ins[i] || !outs[i] || !packed[i]) TFAIL(CALD_ERR_INVALID, "bad problem %d", i);
        const LevelSeg *si, *so;
        if (int rc = dense_seg(c, N, H, W, &si)) return rc;
        if (int rc = dense_seg(c, N, Ho, Wo, &so)) return rc;
        ConvArgs& a = probs[i]; memset(&a, 0, sizeof(a));
        a.in = ins[i]; a.out = outs[i]; a.w = packed[i]; a.w4 = packed[i] + nw;
        const float* vec = packed[i] + 2 * nw;
        a.bias = (flags & 1) ? vec : nullptr; a.scale = (flags & 2) ? vec + g.NPad : nullptr; a.shift = (flags & 2) ? vec + 2 * g.NPad : nullptr;
        a.seg_in = si; a.seg_out = so; a.seg_up = so;
        a.V = N; a.Cin = g.cin_conv; a.Cout = g.n_true; a.CoutPad = g.NPad; a.Kpad = g.Kpad; a.KH = kh; a.KW = kw; a.stride = stride; a.pad = pad;
        a.relu = (flags & 4) ? 1 : 0; a.total_mtiles = N * ((Ho * Wo + 127) / 128); a.out_ld = out_ld; a.zeros = cald_internal_zeros(c);
        a.mask = masks ? masks[i] : nullptr;
    }
    launch_conv_group(probs, n, cald_internal_stream(c));
    THIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x; const float* g; float* partial;
    int N, H, W, Cin, ldx;
    int Ho, Wo, Cout, ldg;
    int KH, KW, stride, pad;
    int J, JT, MT;
    long long Q, chunk;
};
// grid (MT * JT, S), 256 threads = 4 waves (2 x 2), tile 128 (co) x 128 (j = tap * Cin + ci) x BK output pixels per stage.
// FAST (both tensors < 2 GB): raw buffer loads whose byte offsets advance by additions only; a row outside its image / past the end of
// the split gets the out-of-range offset and loads zeros -- no branches, no 64-bit multiplies in the loop (the fp32 MFMA shares its
// issue slots with the VALU: the first version spent a third of the loop on address arithmetic).
typedef float f32x4v __attribute__((ext_vector_type(4)));
// PW (1 x 1 filter, stride 1, no padding, or a linear layer; implies FAST): both operands are plain row-major matrices over the pixels,
// so the cursors are two additions per row and the buffer descriptors end at the tensors' ends -- rows past the last pixel load zeros
// without a compare (the general path spends ~70 VALU instructions per stage on the pixel cursor and the image-border tests).
// MODE 2 (filters with a spatial extent / strides, Cin % 128 == 0 so that the 128 columns of a tile are ONE tap; implies FAST): the
// byte offset of every pixel under that tap -- or the out-of-range offset outside the image -- is computed 1 024 pixels at a time into
// an LDS table; a stage reads two table words per thread instead of stepping (n, y, x) cursors.
template <bool FAST, int BK, int MODE = 0>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
    constexpr bool PW = MODE == 1, TAB = MODE == 2;
    constexpr int TQ = 1024;                                  // MODE 2: pixels per table fill (a multiple of 2 * BK: every fill starts on stage buffer 0)
    __shared__ int tab[TAB ? TQ + 2 * BK : 1];
    constexpr int R = BK / 8;                                    // rows of both tiles staged per thread
    __shared__ __attribute__((aligned(16))) float sA[2][BK][128];
    __shared__ __attribute__((aligned(16))) float sB[2][BK][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int mt = blockIdx.x / a.JT, jt = blockIdx.x - mt * a.JT;
    const int m0 = mt * 128, j0 = jt * 128;
    const long long q0 = (long long)blockIdx.y * a.chunk;
    long long q1 = q0 + a.chunk; if (q1 > a.Q) q1 = a.Q;
    const int col4 = (tid & 31) * 4, row = tid >> 5;          // this thread stages rows row, row + 8, ... of both tiles
    // B column (fixed for the whole kernel): j -> (tap, ci)
    const int j = j0 + col4;
    const bool jvalid = j < a.J;
    const int tap = jvalid ? j / a.Cin : 0, ci = j - tap * a.Cin;
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    const bool mvalid = m0 + col4 < a.ldg;
    // pixel cursors of the staged rows
    int pn[R], py[R], px[R];
    long long pq[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        pq[r] = q0 + row + 8 * r;
        const long long hw = (long long)a.Ho * a.Wo;
        pn[r] = (int)(pq[r] / hw);
        const int rem = (int)(pq[r] - (long long)pn[r] * hw);
        py[r] = rem / a.Wo; px[r] = rem - py[r] * a.Wo;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int jn = 0; jn < 2; jn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][jn][r] = 0.0f;
    float4 ra[R], rb[R];
    // FAST-path cursor: byte offsets from the tensor bases, remaining rows of the split, input coordinates of this thread's tap
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)a.g, 0, (PW || TAB) ? (int)(a.Q * a.ldg * 4) : 0x7FFE0000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, PW ? (int)(a.Q * a.ldx * 4) : 0x7FFE0000, 0x00020000);
    int voffG[R], voffX[R], left[R], fiy[R], fix[R], fpx[R], fpy[R];
    const int stepG = BK * a.ldg * 4, stepX = BK * a.stride * a.ldx * 4, stepI = BK * a.stride;
    const int rowX = (a.stride * a.W - a.Wo * a.stride) * a.ldx * 4, imgX = (a.H - a.Ho * a.stride) * a.W * a.ldx * 4;
    if (FAST) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            voffG[r] = (int)((pq[r] * a.ldg + m0 + col4) * 4);
            left[r] = (int)(q1 - pq[r]);
            fpy[r] = py[r]; fpx[r] = px[r];
            fiy[r] = py[r] * a.stride + ky - a.pad; fix[r] = px[r] * a.stride + kx - a.pad;
            voffX[r] = (int)(((((long long)pn[r] * a.H + fiy[r]) * a.W + fix[r]) * a.ldx + ci) * 4);
            if (PW || TAB) {                                 // columns beyond the matrix: the out-of-range offset for the whole kernel
                if (!mvalid) voffG[r] = 0x7FFF0000;          // (the cursor advances by less than 2 GB in total: no wrap-around)
                voffX[r] = jvalid ? (int)((pq[r] * a.ldx + ci) * 4) : 0x7FFF0000;
            }
        }
    }
    int ti = row;                                            // MODE 2: table index of this thread's first row of the next stage
    const int ci4 = jvalid ? ci * 4 : 0x7FFF0000;            // added to a table word; any word + 0x7FFF0000 is out of range
    // table words [0, TQ + 2 BK) <-> pixels sub .. : the BK words past TQ are the first stage of the next fill, which the last stage of
    // this one prefetches
    auto fill = [&](long long sub) {
        const int hw = a.Ho * a.Wo;
        for (int e = tid; e < TQ + 2 * BK; e += 256) {
            int off = 0x7FFF0000;
            const long long q = sub + e;
            if (q < q1) {
                const int n = (int)(q / hw), rem = (int)(q - (long long)n * hw);
                const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
                const int iy = oy * a.stride + ky - a.pad, ix = ox * a.stride + kx - a.pad;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) off = (int)(((((long long)n * a.H + iy) * a.W + ix) * a.ldx) * 4);
            }
            tab[e] = off;
        }
        __syncthreads();
    };
    if (TAB) fill(q0);
    auto load = [&]() {
        if (TAB) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int vx = tab[ti + 8 * r] + ci4;
                ra[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsG, voffG[r], 0, 0));
                rb[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, vx, 0, 0));
                voffG[r] += stepG;
            }
            ti += BK;
            return;
        }
        if (PW) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                ra[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsG, voffG[r], 0, 0));
                rb[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, voffX[r], 0, 0));
                voffG[r] += stepG; voffX[r] += stepX;
            }
            return;
        }
        if (FAST) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const bool live = left[r] > 0;
                const int vg = (live && mvalid) ? voffG[r] : 0x7FFF0000;
                const bool inimg = fiy[r] >= 0 && fiy[r] < a.H && fix[r] >= 0 && fix[r] < a.W;
                const int vx = (live && jvalid && inimg) ? voffX[r] : 0x7FFF0000;
                ra[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsG, vg, 0, 0));
                rb[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, vx, 0, 0));
                voffG[r] += stepG; left[r] -= BK;
                voffX[r] += stepX; fix[r] += stepI; fpx[r] += BK;
                while (fpx[r] >= a.Wo) {
                    fpx[r] -= a.Wo; fix[r] -= a.Wo * a.stride; voffX[r] += rowX; fiy[r] += a.stride;
                    if (++fpy[r] >= a.Ho) { fpy[r] = 0; fiy[r] -= a.Ho * a.stride; voffX[r] += imgX; }
                }
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            ra[r] = make_float4(0.f, 0.f, 0.f, 0.f); rb[r] = ra[r];
            if (pq[r] < q1) {
                if (mvalid) ra[r] = *reinterpret_cast<const float4*>(a.g + pq[r] * a.ldg + m0 + col4);
                const int iy = py[r] * a.stride + ky - a.pad, ix = px[r] * a.stride + kx - a.pad;
                if (jvalid && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                    rb[r] = *reinterpret_cast<const float4*>(a.x + (((long long)pn[r] * a.H + iy) * a.W + ix) * a.ldx + ci);
            }
            // advance the cursor by one stage (BK pixels); stages past the end of the split load nothing and contribute zeros
            pq[r] += BK; px[r] += BK;
            while (px[r] >= a.Wo) { px[r] -= a.Wo; if (++py[r] >= a.Ho) { py[r] = 0; pn[r]++; } }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            *reinterpret_cast<float4*>(&sA[buf][row + 8 * r][col4]) = ra[r];
            *reinterpret_cast<float4*>(&sB[buf][row + 8 * r][col4]) = rb[r];
        }
    };
    const int kl = lane >> 5, cl = lane & 31;
    const int stages = (int)((q1 - q0 + BK - 1) / BK);
    load(); store(0);
    __syncthreads();
    // one stage; the buffer index is a compile-time constant (two stages per loop turn), so every LDS address below is a register +
    // an immediate -- with a run-time buffer index each of the 16 fragment reads costs a v_add beside the MFMAs
    auto stage = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        load();                                               // the next stage (zeros past the end)
        // fragments of k-step ks + 1 are read from LDS before the MFMAs of k-step ks issue (the compiler does not hoist them itself)
        float fa0 = sA[buf][kl][wm * 64 + cl], fa1 = sA[buf][kl][wm * 64 + 32 + cl];
        float fb0 = sB[buf][kl][wn * 64 + cl], fb1 = sB[buf][kl][wn * 64 + 32 + cl];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ks++) {
            const float a0 = fa0, a1 = fa1, b0 = fb0, b1 = fb1;
            if (ks + 1 < BK / 2) {
                fa0 = sA[buf][2 * ks + 2 + kl][wm * 64 + cl]; fa1 = sA[buf][2 * ks + 2 + kl][wm * 64 + 32 + cl];
                fb0 = sB[buf][2 * ks + 2 + kl][wn * 64 + cl]; fb1 = sB[buf][2 * ks + 2 + kl][wn * 64 + 32 + cl];
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the reads above the MFMAs: the scheduler otherwise sinks them to their use
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        store(buf ^ 1);
        __syncthreads();
    };
    if (TAB) {
        for (int s0 = 0; s0 < stages; s0 += TQ / BK) {
            if (s0) { fill(q0 + (long long)s0 * BK); ti = row + BK; }      // (the stage before ended on a barrier; stage s0 is already staged)
            const int s1 = s0 + TQ / BK < stages ? s0 + TQ / BK : stages;
            for (int s = s0; s < s1; s += 2) {
                stage(std::integral_constant<int, 0>());
                if (s + 1 < s1) stage(std::integral_constant<int, 1>());
            }
        }
    } else {
        for (int s = 0; s < stages; s += 2) {
            stage(std::integral_constant<int, 0>());
            if (s + 1 < stages) stage(std::integral_constant<int, 1>());
        }
    }
    // partial tile -> partial[S][MT*128][JT*128]
    const long long ldp = (long long)a.JT * 128;
    float* P = a.partial + (long long)blockIdx.y * ((long long)a.MT * 128) * ldp;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int jn = 0; jn < 2; jn++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                const int cc = j0 + wn * 64 + jn * 32 + cl;
                P[(long long)rr * ldp + cc] = acc[i][jn][r];
            }
}
// grad[co][ci][tap] (torch layout) (+)= sum over splits, in split order
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int S, long long split_stride, long long ldp, int Cout, int Cin, int taps,
                                    float* __restrict__ grad, int accumulate, const float* __restrict__ row_scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long J = (long long)taps * Cin;
    if (i >= (long long)Cout * J) return;
    const int co = (int)(i / J); const int j = (int)(i - (long long)co * J);
    const int tap = j / Cin, ci = j - tap * Cin;
    // the splits are summed in split order (fixed, reproducible); sixteen loads are in flight at a time -- a layer with few outputs
    // and hundreds of splits (the 15-column RPN head: 3 840 outputs x 512 splits) spent 0.47 ms per launch waiting for one load after
    // the other
    float s = 0.0f;
    const float* q = partial + (long long)co * ldp + j;
    int k = 0;
    for (; k + 16 <= S; k += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = q[(long long)(k + u) * split_stride];
#pragma unroll
        for (int u = 0; u < 16; u++) s += v[u];
    }
    for (; k < S; k++) s += q[(long long)k * split_stride];
    if (row_scale) s = s * row_scale[co];
    float* dst = grad + ((long long)co * Cin + ci) * taps + tap;
    *dst = accumulate ? *dst + s : s;
}
// the same for taps > 1 with coalesced stores: one workgroup sums 64 input channels x all taps of one output channel (reads run
// along ci), turns the [tap][ci] block into the torch [ci][tap] order in LDS and stores it as one contiguous run
__global__ __launch_bounds__(256) void wgrad_reduce_taps_kernel(const float* __restrict__ partial, int S, long long split_stride, long long ldp, int Cin,
                                                                int taps, float* __restrict__ grad, int accumulate, const float* __restrict__ row_scale) {
    extern __shared__ float blk[];     // [64][taps]
    const int co = blockIdx.y, c0 = blockIdx.x * 64;
    const int nc = Cin - c0 < 64 ? Cin - c0 : 64;
    const float* P = partial + (long long)co * ldp + c0;
    const float sc = row_scale ? row_scale[co] : 1.0f;
    for (int e = threadIdx.x; e < taps * 64; e += 256) {
        const int t = e >> 6, cl = e & 63;
        if (cl >= nc) continue;
        const float* q = P + (long long)t * Cin + cl;
        float s = 0.0f;
        int k = 0;
        for (; k + 8 <= S; k += 8) {          // split order kept; eight loads in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = q[(long long)(k + u) * split_stride];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; k < S; k++) s += q[(long long)k * split_stride];
        if (row_scale) s = s * sc;
        blk[cl * taps + t] = s;
    }
    __syncthreads();
    float* dst = grad + ((long long)co * Cin + c0) * taps;
    for (int e = threadIdx.x; e < taps * nc; e += 256) dst[e] = accumulate ? dst[e] + blk[e] : blk[e];
}
// db[c] = sum over rows of g[q][c]: stage 1 partial sums over row blocks (128 channels x 8 row lanes per workgroup, float4
// loads), stage 2 fixed-order sum over the row blocks
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* g, long long Q, int C, int ld, long long rows_per_block, float* partial) {
    __shared__ float4 red[8][32];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 128 + cq * 4;
    const long long q0 = (long long)blockIdx.y * rows_per_block;
    long long q1 = q0 + rows_per_block; if (q1 > Q) q1 = Q;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < ld)
        for (long long q = q0 + rl; q < q1; q += 8) {
            const float4 v = *reinterpret_cast<const float4*>(g + q * ld + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    red[rl][cq] = s;
    __syncthreads();
    if (rl == 0) {
        for (int r = 1; r < 8; r++) { const float4 v = red[r][cq]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        float* o = partial + (long long)blockIdx.y * C;
        if (c < C) o[c] = s.x;
        if (c + 1 < C) o[c + 1] = s.y;
        if (c + 2 < C) o[c + 2] = s.z;
        if (c + 3 < C) o[c + 3] = s.w;
    }
}
// 16 channels x 16 split lanes per workgroup: lane l sums splits l, l+16, ... in order, then lane 0 adds the 16 lane sums in order
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* partial, int S, int C, float* out, int accumulate) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s = 0.0f;
    if (c < C)
        for (int k = lane; k < S; k += 16) s += partial[(long long)k * C + c];
    red[lane][cl] = s;
    __syncthreads();
    if (lane == 0 && c < C) {
        for (int l = 1; l < 16; l++) s += red[l][cl];
        out[c] = accumulate ? out[c] + s : s;
    }
}

// x [N][H][W][ldx] (Cin channels used, Cin % 4 == 0), g [N][Ho][Wo][ldg] (Cout channels used; ldg % 4 == 0 and the pad channels
// readable).  dw: torch layout [Cout][Cin][KH][KW]; for a tap-major linear layer pass KH*KW = taps, H = W = 1 and x rows
// [R][taps * Cin] as N = R... (see cald_train_linear_wgrad).  db: [Cout] or null.
static int wgrad_impl(cald_ctx* c, long long Q, int N, int H, int W, const float* x, int Cin, int ldx, int Ho, int Wo, const float* g,
                      int Cout, int ldg, int KH, int KW, int stride, int pad, int red_taps, int red_cin, float* dw, float* db, int accumulate,
                      const float* row_scale) {
    if (Cin % 4 || ldx % 4 || ldg % 4) TFAIL(CALD_ERR_INVALID, "Cin and the row strides must be multiples of 4");
    THIP(hipSetDevice(cald_internal_device(c)));
    hipStream_t st = cald_internal_stream(c);
    WgradArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.g = g; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.ldx = ldx; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.ldg = ldg;
    a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.J = KH * KW * Cin; a.JT = (a.J + 127) / 128; a.MT = (Cout + 127) / 128; a.Q = Q;
    const long long tiles = (long long)a.MT * a.JT;
    const long long tile_floats = tiles * 128 * 128;
    // workgroups aimed at per launch: one resident round (2 per CU).  Measured on the whole step (the kernel shares the chip with the
    // data-gradient stream): 512 -> 34.6 / 33.8 ms (Faster R-CNN / RetinaNet), 384 -> 34.3 / 34.7, 1024 -> 34.9, 2048 -> 35.0 / 35.5
    static const long long target = getenv("CALD_WGRAD_TARGET") ? atoll(getenv("CALD_WGRAD_TARGET")) : 512;
    const long long bytesG = Q * ldg * 4, bytesX = (long long)N * H * W * ldx * 4;
    const bool fast = bytesG < 0x7FFE0000ll && bytesX < 0x7FFE0000ll;
    static const bool pw_env = !(getenv("CALD_WGRAD_PW") && atoi(getenv("CALD_WGRAD_PW")) == 0);
    static const bool tab_env = !(getenv("CALD_WGRAD_TAB") && atoi(getenv("CALD_WGRAD_TAB")) == 0);
    // pointwise: every pixel row q of g pairs with row q of x (chunks are multiples of 32 pixels, so only the last split runs past Q)
    const bool pw = fast && pw_env && KH == 1 && KW == 1 && stride == 1 && pad == 0 && Ho == H && Wo == W && Q == (long long)N * H * W;
    const bool tabm = fast && tab_env && !pw && Cin % 128 == 0;
    long long S = target / tiles;                         // rounded down: one workgroup over the resident round costs a second round
    const long long maxS_rows = (Q + 255) / 256; if (S > maxS_rows) S = maxS_rows;
    const long long cap = (512ll << 20) / 4 / tile_floats; if (S > cap) S = cap;
    if (S < 1) S = 1;
    a.chunk = ((Q + S - 1) / S + 31) / 32 * 32;
    S = (Q + a.chunk - 1) / a.chunk;
    const long long csplit = (Q + 255) / 256 > 1024 ? 1024 : (Q + 255) / 256;
    void* scratch = nullptr;
    if (int rc = cald_internal_scratch(c, (size_t)(S * tile_floats + csplit * Cout + 64) * 4, &scratch)) return rc;
    a.partial = (float*)scratch;
    static const int bk_env = getenv("CALD_WGRAD_BK") ? atoi(getenv("CALD_WGRAD_BK")) : 16;     // 32-pixel stages measured slower (109 vs 115 TFLOP/s on the largest layer)
    if (pw) hipLaunchKernelGGL((wgrad_kernel<true, 16, 1>), dim3((unsigned)tiles, (unsigned)S), dim3(256), 0, st, a);
    else if (tabm) hipLaunchKernelGGL((wgrad_kernel<true, 16, 2>), dim3((unsigned)tiles, (unsigned)S), dim3(256), 0, st, a);
    else if (fast && bk_env == 32) hipLaunchKernelGGL((wgrad_kernel<true, 32>), dim3((unsigned)tiles, (unsigned)S), dim3(256), 0, st, a);
    else if (fast) hipLaunchKernelGGL((wgrad_kernel<true, 16>), dim3((unsigned)tiles, (unsigned)S), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad_kernel<false, 16>), dim3((unsigned)tiles, (unsigned)S), dim3(256), 0, st, a);
    const long long nred = (long long)Cout * a.J;
    if (red_taps > 1 && red_taps <= 64)
        hipLaunchKernelGGL(wgrad_reduce_taps_kernel, dim3((unsigned)((red_cin + 63) / 64), (unsigned)Cout), dim3(256), (size_t)red_taps * 64 * 4, st,
                           a.partial, (int)S, tile_floats, (long long)a.JT * 128, red_cin, red_taps, dw, accumulate, row_scale);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nred + 255) / 256)), dim3(256), 0, st, a.partial, (int)S, tile_floats,
                           (long long)a.JT * 128, Cout, red_cin, red_taps, dw, accumulate, row_scale);
    if (db) {
        float* cp = a.partial + S * tile_floats;
        const long long rpb = (Q + csplit - 1) / csplit;
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((Cout + 127) / 128, (unsigned)csplit), dim3(256), 0, st, g, Q, Cout, ldg, rpb, cp);
        hipLaunchKernelGGL(colsum_final_kernel, dim3((Cout + 15) / 16), dim3(256), 0, st, cp, (int)csplit, Cout, db, accumulate);
    }
    THIP(hipGetLastError());
    return 0;
}
extern "C" int cald_train_conv_wgrad(cald_ctx* c, int N, int H, int W, const float* x, int Cin, int ldx, const float* g, int Cout, int ldg,
                                     int KH, int KW, int stride, int pad, const float* row_scale, float* dw, float* db, int accumulate) {
    if (!c || !x || !g || !dw) TFAIL(CALD_ERR_INVALID, "null argument");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (N < 1 || Ho < 1 || Wo < 1) TFAIL(CALD_ERR_INVALID, "bad geometry");
    if (row_scale && db) TFAIL(CALD_ERR_INVALID, "row_scale is for FrozenBatchNorm layers, which carry no bias");
    return wgrad_impl(c, (long long)N * Ho * Wo, N, H, W, x, Cin, ldx, Ho, Wo, g, Cout, ldg, KH, KW, stride, pad, KH * KW, Cin, dw, db, accumulate, row_scale);
}
// linear layer on rows: x [R][K], g [R][ldg] -> dw [Cout][K] torch layout; taps > 1: rows are [tap][K / taps] and the torch
// weight is [Cout][K / taps][taps] (fc6 on the RoIAlign output)
extern "C" int cald_train_linear_wgrad(cald_ctx* c, int R, const float* x, int K, const float* g, int Cout, int ldg, int taps,
                                       float* dw, float* db, int accumulate) {
    if (!c || !x || !g || !dw) TFAIL(CALD_ERR_INVALID, "null argument");
    if (R < 1 || taps < 1 || K % taps) TFAIL(CALD_ERR_INVALID, "bad geometry");
    return wgrad_impl(c, R, 1, 1, R, x, K, K, 1, R, g, Cout, ldg, 1, 1, 1, 0, taps, K / taps, dw, db, accumulate, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------
// elementwise pieces of the backward pass
// ---------------------------------------------------------------------------------------------------------------------
// g[q][c] = (act[q][c] > 0 ? g[q][c] : 0) * (scale ? scale[c] : 1): ReLU backward (act = the layer's post-ReLU output) and
// the FrozenBatchNorm scale in one pass.  act == null: scale only.
__global__ void relu_bwd_kernel(float* g, const float* act, const float* scale, long long n4, int C4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = reinterpret_cast<float4*>(g)[i];
    if (act) { const float4 y = reinterpret_cast<const float4*>(act)[i]; if (!(y.x > 0.f)) v.x = 0.f; if (!(y.y > 0.f)) v.y = 0.f; if (!(y.z > 0.f)) v.z = 0.f; if (!(y.w > 0.f)) v.w = 0.f; }
    if (scale) { const float4 s = reinterpret_cast<const float4*>(scale)[i % C4]; v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w; }
    reinterpret_cast<float4*>(g)[i] = v;
}
extern "C" int cald_train_relu_bwd(cald_ctx* c, long long rows, int C, float* g, const float* act, const float* scale) {
    if (!c || !g || C % 4) TFAIL(CALD_ERR_INVALID, "bad arguments (C must be a multiple of 4)");
    THIP(hipSetDevice(cald_internal_device(c)));
    const long long n4 = rows * C / 4;
    if (n4 > 0) hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, cald_internal_stream(c), g, act, scale, n4, C / 4);
    THIP(hipGetLastError());
    return 0;
}
// dst = a + b (b may be null: copy), float4 granularity
__global__ void add_kernel(float* dst, const float* a, const float* b, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = reinterpret_cast<const float4*>(a)[i];
    if (b) { const float4 w = reinterpret_cast<const float4*>(b)[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    reinterpret_cast<float4*>(dst)[i] = v;
}
extern "C" int cald_train_add(cald_ctx* c, long long n, float* dst, const float* a, const float* b) {
    if (!c || !dst || !a || n % 4) TFAIL(CALD_ERR_INVALID, "bad arguments (n must be a multiple of 4)");
    THIP(hipSetDevice(cald_internal_device(c)));
    if (n > 0) hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, cald_internal_stream(c), dst, a, b, n / 4);
    THIP(hipGetLastError());
    return 0;
}
// stride-s data gradient, step 1: scatter g [N][Ho][Wo][C] onto the stride-1 grid [N][Hd][Wd][C] (zeros elsewhere), Hd = (Ho-1)*s+1 + extra
__global__ void dilate_kernel(const float* g, float* out, int N, int Ho, int Wo, int Hd, int Wd, int C4, int s) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n4 = (long long)N * Hd * Wd * C4;
    if (i >= n4) return;
    const int c = (int)(i % C4); long long p = i / C4;
    const int x = (int)(p % Wd); p /= Wd; const int y = (int)(p % Hd); const int n = (int)(p / Hd);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y % s == 0 && x % s == 0 && y / s < Ho && x / s < Wo) v = reinterpret_cast<const float4*>(g)[(((long long)n * Ho + y / s) * Wo + x / s) * C4 + c];
    reinterpret_cast<float4*>(out)[i] = v;
}
extern "C" int cald_train_dilate(cald_ctx* c, int N, int Ho, int Wo, int C, int s, int Hd, int Wd, const float* g, float* out) {
    if (!c || !g || !out || C % 4 || s < 1) TFAIL(CALD_ERR_INVALID, "bad arguments");
    THIP(hipSetDevice(cald_internal_device(c)));
    const long long n4 = (long long)N * Hd * Wd * (C / 4);
    hipLaunchKernelGGL(dilate_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, cald_internal_stream(c), g, out, N, Ho, Wo, Hd, Wd, C / 4, s);
    THIP(hipGetLastError());
    return 0;
}
// stride-2 data gradient, last step: the four phase results (output pixels (2m + a, 2n + b) depend on disjoint filter taps) are woven
// into dX [N][H][W][C]; phase (a, b) lives in ph[2a + b] = [N][Hp_ab][Wp_ab][C] and its pixel (m + off, n + off) belongs to (2m + a, 2n + b).
// mask (optional, same shape as dX): ReLU backward of the layer dX flows into.
struct WeaveArgs { const float* ph[4]; int Hp[4], Wp[4], off[4]; };
__global__ void weave2_kernel(WeaveArgs a, const float* mask, float* out, int N, int H, int W, int C4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n4 = (long long)N * H * W * C4;
    if (i >= n4) return;
    const int c = (int)(i % C4); long long p = i / C4;
    const int x = (int)(p % W); p /= W; const int y = (int)(p % H); const int n = (int)(p / H);
    const int k = 2 * (y & 1) + (x & 1);
    const int py = (y >> 1) + a.off[k], px = (x >> 1) + a.off[k];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (py < a.Hp[k] && px < a.Wp[k]) v = reinterpret_cast<const float4*>(a.ph[k])[(((long long)n * a.Hp[k] + py) * a.Wp[k] + px) * C4 + c];
    if (mask) { const float4 m = reinterpret_cast<const float4*>(mask)[i]; if (!(m.x > 0.f)) v.x = 0.f; if (!(m.y > 0.f)) v.y = 0.f; if (!(m.z > 0.f)) v.z = 0.f; if (!(m.w > 0.f)) v.w = 0.f; }
    reinterpret_cast<float4*>(out)[i] = v;
}
extern "C" int cald_train_weave2(cald_ctx* c, int N, int H, int W, int C, const float* const* phases, const int* phase_hw, const int* phase_off,
                                 const float* mask, float* out) {
    if (!c || !phases || !phase_hw || !phase_off || !out || C % 4) TFAIL(CALD_ERR_INVALID, "bad arguments");
    THIP(hipSetDevice(cald_internal_device(c)));
    WeaveArgs a;
    for (int k = 0; k < 4; k++) { a.ph[k] = phases[k]; a.Hp[k] = phase_hw[2 * k]; a.Wp[k] = phase_hw[2 * k + 1]; a.off[k] = phase_off[k]; if (!phases[k]) TFAIL(CALD_ERR_INVALID, "null phase"); }
    const long long n4 = (long long)N * H * W * (C / 4);
    hipLaunchKernelGGL(weave2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, cald_internal_stream(c), a, mask, out, N, H, W, C / 4);
    THIP(hipGetLastError());
    return 0;
}
// FPN top-down backward: coarse[n][yc][xc][c] += sum of fine[n][yf][xf][c] over the fine pixels whose nearest source is (yc, xc)
// (F.interpolate(size=fine, mode='nearest'): source = floor(dst * coarse / fine))
__global__ void upsample_bwd_kernel(const float* fine, float* coarse, int N, int Hf, int Wf, int Hc, int Wc, int C4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n4 = (long long)N * Hc * Wc * C4;
    if (i >= n4) return;
    const int c = (int)(i % C4); long long p = i / C4;
    const int xc = (int)(p % Wc); p /= Wc; const int yc = (int)(p % Hc); const int n = (int)(p / Hc);
    // fine rows y with floor(y * Hc / Hf) == yc  <=>  y in [ceil(yc * Hf / Hc), ceil((yc + 1) * Hf / Hc))
    const int y0 = (yc * Hf + Hc - 1) / Hc, y1 = ((yc + 1) * Hf + Hc - 1) / Hc;
    const int x0 = (xc * Wf + Wc - 1) / Wc, x1 = ((xc + 1) * Wf + Wc - 1) / Wc;
    float4 s = reinterpret_cast<float4*>(coarse)[i];
    for (int y = y0; y < y1 && y < Hf; y++)
        for (int x = x0; x < x1 && x < Wf; x++) {
            const float4 v = reinterpret_cast<const float4*>(fine)[(((long long)n * Hf + y) * Wf + x) * C4 + c];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    reinterpret_cast<float4*>(coarse)[i] = s;
}
extern "C" int cald_train_upsample_bwd(cald_ctx* c, int N, int Hf, int Wf, int Hc, int Wc, int C, const float* fine, float* coarse) {
    if (!c || !fine || !coarse || C % 4) TFAIL(CALD_ERR_INVALID, "bad arguments");
    THIP(hipSetDevice(cald_internal_device(c)));
    const long long n4 = (long long)N * Hc * Wc * (C / 4);
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, cald_internal_stream(c), fine, coarse, N, Hf, Wf, Hc, Wc, C / 4);
    THIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// optimizer: torch.optim.SGD(lr, momentum, weight_decay), dampening 0, no nesterov (cald_train.py:397)
//   d = grad + wd * p;  buf = first ? d : momentum * buf + d;  p -= lr * buf
// ---------------------------------------------------------------------------------------------------------------------
__global__ void sgd_kernel(float* p, const float* grad, float* buf, long long n, float lr, float momentum, float wd, int first) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d = grad[i];
    if (wd != 0.0f) d = d + wd * p[i];
    if (momentum != 0.0f) { const float b = first ? d : momentum * buf[i] + d; buf[i] = b; d = b; }
    p[i] = p[i] - lr * d;
}
extern "C" int cald_train_sgd(cald_ctx* c, long long n, float* param, const float* grad, float* momentum_buf, float lr, float momentum,
                              float weight_decay, int first_step) {
    if (!c || !param || !grad || (momentum != 0.0f && !momentum_buf)) TFAIL(CALD_ERR_INVALID, "null argument");
    THIP(hipSetDevice(cald_internal_device(c)));
    if (n > 0) hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cald_internal_stream(c), param, grad, momentum_buf, n, lr, momentum, weight_decay, first_step);
    THIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// region proposals in training mode (torchvision RegionProposalNetwork.filter_proposals with pre/post_nms_top_n_train = 2000,
// detection/frcnn_la.py:154-158): the inference kernels of rpn.hip at the larger capacity
// ---------------------------------------------------------------------------------------------------------------------
// heads[l]: [N][Hl][Wl][head_ld], channel a = objectness logit of anchor a, channel A + 4a + j = delta j.  image_sizes: host [N][2]
// (resized, unpadded h, w).  proposals_out: device [N][post_n][4], counts_out: device int [N].
extern "C" int cald_train_rpn_proposals(cald_ctx* c, int N, int Hp, int Wp, const int* image_sizes, const float* const* heads,
                                        const int* level_hw, int head_ld, int pre_n, int post_n, float nms_thr, float min_size,
                                        float* proposals_out, int* counts_out) {
    if (!c || !image_sizes || !heads || !level_hw || !proposals_out || !counts_out) TFAIL(CALD_ERR_INVALID, "null argument");
    if (N < 1 || N > CALD_MAX_VIEWS || pre_n < 1 || pre_n > 2048 || post_n < 1 || post_n > 2048) TFAIL(CALD_ERR_INVALID, "N <= %d, pre/post top-n <= 2048", CALD_MAX_VIEWS);
    THIP(hipSetDevice(cald_internal_device(c)));
    SEG_ENTRY();
    hipStream_t st = cald_internal_stream(c);
    RpnArgs ra; memset(&ra, 0, sizeof(ra));
    const LevelSeg* s0;
    if (int rc = dense_seg(c, N, Hp, Wp, &s0)) return rc;
    ra.seg0 = s0;
    for (int l = 0; l < 5; l++) {
        const LevelSeg* sl;
        if (int rc = dense_seg(c, N, level_hw[2 * l], level_hw[2 * l + 1], &sl)) return rc;
        ra.seg[l] = sl; ra.head[l] = heads[l];
    }
    const size_t ntot = (size_t)N * 5 * pre_n;
    const size_t b_views = (sizeof(ViewDesc) * N + 255) & ~(size_t)255, b_anch = 256, b_key = (ntot * 8 + 255) & ~(size_t)255, b_box = (ntot * 16 + 255) & ~(size_t)255;
    void* scratch = nullptr;
    if (int rc = cald_internal_scratch(c, b_views + b_anch + b_key + 3 * b_box + 256, &scratch)) return rc;
    char* p = (char*)scratch;
    ViewDesc* d_views = (ViewDesc*)p; p += b_views;
    float* d_base = (float*)p; p += b_anch;
    ra.cand_key = (unsigned long long*)p; p += b_key;
    ra.cand_box = (float*)p; p += b_box; ra.sorted_box = (float*)p; p += b_box; ra.sorted_raw = (float*)p; p += b_box;
    ra.sorted_count = (int*)p;
    std::vector<ViewDesc> hv(N);
    memset(hv.data(), 0, sizeof(ViewDesc) * N);
    for (int v = 0; v < N; v++) { hv[v].Hr = image_sizes[2 * v]; hv[v].Wr = image_sizes[2 * v + 1]; }
    float base[5 * 3 * 4];
    {   // AnchorGenerator base anchors: sizes (32, 64, 128, 256, 512), ratios (0.5, 1, 2)  (frcnn_la.py:185-187)
        const float sizes[5] = {32.f, 64.f, 128.f, 256.f, 512.f}, ratios[3] = {0.5f, 1.0f, 2.0f};
        for (int l = 0; l < 5; l++)
            for (int r = 0; r < 3; r++) {
                const float hr = sqrtf(ratios[r]), wr = 1.0f / hr, ws = wr * sizes[l], hs = hr * sizes[l];
                float* b = &base[(l * 3 + r) * 4];
                b[0] = rintf(-ws / 2.0f); b[1] = rintf(-hs / 2.0f); b[2] = rintf(ws / 2.0f); b[3] = rintf(hs / 2.0f);
            }
    }
    // the two host tables go through the pinned staging ring (one blob: base anchors, then the view descriptors), so the call returns with
    // the proposal kernels enqueued instead of first waiting for everything before them on the stream (the whole body + FPN + RPN head)
    std::vector<char> blob(256 + sizeof(ViewDesc) * N);
    memcpy(blob.data(), base, sizeof(base)); memcpy(blob.data() + 256, hv.data(), sizeof(ViewDesc) * N);
    const bool ring = blob.size() <= kStageSlot;
    const void* dv = nullptr; int slot = 0;
    if (ring) {
        if (int rc = stage_upload(c, blob.data(), blob.size(), st, &dv, &slot)) return rc;
        d_base = (float*)dv; d_views = (ViewDesc*)((const char*)dv + 256);
    } else {
        THIP(hipMemcpyAsync(d_views, hv.data(), sizeof(ViewDesc) * N, hipMemcpyHostToDevice, st));
        THIP(hipMemcpyAsync(d_base, base, sizeof(base), hipMemcpyHostToDevice, st));
        THIP(hipStreamSynchronize(st));   // the host arrays above are stack / vector storage
    }
    ra.views = d_views; ra.base_anchors = d_base; ra.head_ld = head_ld; ra.A = 3; ra.V = N; ra.pre_n = pre_n; ra.post_n = post_n;
    ra.nms_thr = nms_thr; ra.min_size = min_size; ra.proposals = proposals_out; ra.prop_stride = post_n; ra.prop_count = counts_out;
    launch_rpn(ra, st);
    THIP(hipGetLastError());
    if (ring) return stage_consumed(c, slot, st);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// RoI sampling on the host (roi_heads.select_training_samples: labels from the match results, BalancedPositiveNegativeSampler, the
// index lists the loss kernels read) for all images of a batch in one call.  The forward pass stops for this -- the GPU waits while
// the host turns match results into sample indices -- and the numpy version of the loop cost 0.6 ms per step.
// Candidate table of image i: `slots[i]` proposal rows (the first counts[i] are used) followed by n_gt[i] ground-truth rows; images
// back to back.  The sampler keeps the k candidates with the smallest keys of each class (keys: iid uniform draws handed in by the
// caller, one per table row: every k-subset is equally likely -- what randperm(n)[:k] selects).
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int cald_train_roi_sample_host(int N, const int* slots, const int* n_gt, const int* counts, const int32_t* matched,
                                          const int64_t* gt_labels, const double* keys, int batch, double pos_fraction, int pred_ld, int num_classes,
                                          int64_t* keep_rows, int64_t* gt_sel, int64_t* labels_out, float* img_col, int64_t* pos_rows,
                                          int64_t* pred_idx, int* R_out, int* n_pos_out, int* per_image_out) {
    if (N < 1 || !slots || !n_gt || !matched || !keys || !keep_rows || !gt_sel || !labels_out || !img_col || !pos_rows || !pred_idx || !R_out || !n_pos_out)
        TFAIL(CALD_ERR_INVALID, "null argument");
    if (batch < 1 || pos_fraction < 0.0 || pos_fraction > 1.0) TFAIL(CALD_ERR_INVALID, "bad sampler parameters");
    long long row0 = 0, g0 = 0, gtot = 0;
    for (int i = 0; i < N; i++) gtot += n_gt[i];
    int R = 0, npos_all = 0;
    std::vector<std::pair<double, int>> pos, neg;
    std::vector<int> kept;
    std::vector<int64_t> lab;
    for (int i = 0; i < N; i++) {
        const int used = counts ? counts[i] : slots[i];
        if (used < 0 || used > slots[i] || n_gt[i] < 0) TFAIL(CALD_ERR_INVALID, "image %d: bad counts", i);
        const int rows = slots[i] + n_gt[i];
        pos.clear(); neg.clear(); lab.assign(rows, -1);
        for (int r = 0; r < rows; r++) {
            if (r >= used && r < slots[i]) continue;                       // an unused proposal slot: not a candidate
            const int m = matched[row0 + r];
            int64_t l;
            if (n_gt[i] == 0) l = 0;
            else if (m >= 0) { if (m >= n_gt[i]) TFAIL(CALD_ERR_INVALID, "image %d: match index out of range", i); l = gt_labels ? gt_labels[g0 + m] : 1; }
            else l = m == -1 ? 0 : -1;                                     // below the low threshold: background; between thresholds: ignored
            lab[r] = l;
            if (l >= 1) pos.emplace_back(keys[row0 + r], r); else if (l == 0) neg.emplace_back(keys[row0 + r], r);
        }
        int num_pos = (int)(batch * pos_fraction); if (num_pos > (int)pos.size()) num_pos = (int)pos.size();
        int num_neg = batch - num_pos; if (num_neg > (int)neg.size()) num_neg = (int)neg.size();
        if (num_pos < (int)pos.size()) std::nth_element(pos.begin(), pos.begin() + num_pos, pos.end());
        if (num_neg < (int)neg.size()) std::nth_element(neg.begin(), neg.begin() + num_neg, neg.end());
        kept.clear();
        for (int k = 0; k < num_pos; k++) kept.push_back(pos[k].second);
        for (int k = 0; k < num_neg; k++) kept.push_back(neg[k].second);
        std::sort(kept.begin(), kept.end());
        for (int r : kept) {
            const int m = matched[row0 + r];
            keep_rows[R] = row0 + r; labels_out[R] = lab[r]; img_col[R] = (float)i;
            gt_sel[R] = n_gt[i] ? g0 + (m > 0 ? m : 0) : gtot;           // images without ground truth point at the extra all-zero box
            if (lab[r] > 0) { pos_rows[npos_all] = R; pred_idx[npos_all] = (int64_t)R * pred_ld + num_classes + 4 * lab[r]; npos_all++; }
            R++;
        }
        if (per_image_out) per_image_out[i] = (int)kept.size();
        row0 += rows; g0 += n_gt[i];
    }
    *R_out = R; *n_pos_out = npos_all;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Matcher (torchvision 0.8.2 _utils.Matcher): IoU of every candidate box with every ground-truth box, best ground truth per
// candidate, thresholds, optional low-quality matches.  out[i] = ground-truth index, -1 below `lo`, -2 between `lo` and `hi`.
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline float box_iou_tv(const float4 a, const float4 b) {   // torchvision.ops.box_iou
    const float aa = (a.z - a.x) * (a.w - a.y), ab = (b.z - b.x) * (b.w - b.y);
    float w = fminf(a.z, b.z) - fmaxf(a.x, b.x), h = fminf(a.w, b.w) - fmaxf(a.y, b.y);
    w = w < 0.0f ? 0.0f : w; h = h < 0.0f ? 0.0f : h;
    const float inter = w * h;
    return inter / ((aa + ab) - inter);
}
__global__ void match_gtmax_kernel(const float4* boxes, int nB, const float4* gt, int nG, unsigned* gtmax_bits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nB) return;
    const float4 b = boxes[i];
    for (int g = 0; g < nG; g++) {
        const float v = box_iou_tv(gt[g], b);
        if (v > 0.0f) atomicMax(&gtmax_bits[g], __float_as_uint(v));   // IoU >= 0: the bit pattern orders like the value
    }
}
__global__ void match_assign_kernel(const float4* boxes, int nB, const float4* gt, int nG, const unsigned* gtmax_bits, float hi, float lo,
                                    int allow_low, int* out, float* best_iou) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nB) return;
    const float4 b = boxes[i];
    float best = -1.0f; int arg = 0; bool low = false;
    for (int g = 0; g < nG; g++) {
        const float v = box_iou_tv(gt[g], b);
        if (v > best) { best = v; arg = g; }                         // first maximum, as torch.max(dim=0)
        if (allow_low && v == __uint_as_float(gtmax_bits[g])) low = true;
    }
    int m = arg;
    if (best < lo) m = -1; else if (best < hi) m = -2;
    if (low) m = arg;
    out[i] = m;
    if (best_iou) best_iou[i] = best;
}
extern "C" int cald_train_match(cald_ctx* c, int n_boxes, const float* boxes, int n_gt, const float* gt, float hi, float lo,
                                int allow_low_quality, int* matched_out, float* best_iou_out) {
    if (!c || !boxes || !gt || !matched_out) TFAIL(CALD_ERR_INVALID, "null argument");
    if (n_boxes < 1 || n_gt < 1) TFAIL(CALD_ERR_INVALID, "needs at least one box and one ground-truth box");
    THIP(hipSetDevice(cald_internal_device(c)));
    hipStream_t st = cald_internal_stream(c);
    void* scratch = nullptr;
    if (int rc = cald_internal_scratch(c, (size_t)n_gt * 4 + 256, &scratch)) return rc;
    unsigned* gm = (unsigned*)scratch;
    THIP(hipMemsetAsync(gm, 0, (size_t)n_gt * 4, st));
    const int blocks = (n_boxes + 255) / 256;
    if (allow_low_quality)
        hipLaunchKernelGGL(match_gtmax_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)boxes, n_boxes, (const float4*)gt, n_gt, gm);
    hipLaunchKernelGGL(match_assign_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)boxes, n_boxes, (const float4*)gt, n_gt, gm, hi, lo,
                       allow_low_quality, matched_out, best_iou_out);
    THIP(hipGetLastError());
    return 0;
}

// all anchors of one image, in torchvision's order (level, y, x, anchor): [sum_l Hl*Wl*A][4]
__global__ void anchors_kernel(float4* out, int Hl, int Wl, int sy, int sx, const float* base, int A, long long off) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)Hl * Wl * A) return;
    const int an = (int)(i % A); const long long pix = i / A;
    const int x = (int)(pix % Wl), y = (int)(pix / Wl);
    out[off + i] = make_float4((float)(x * sx) + base[an * 4], (float)(y * sy) + base[an * 4 + 1], (float)(x * sx) + base[an * 4 + 2], (float)(y * sy) + base[an * 4 + 3]);
}
// kind 0: Faster R-CNN AnchorGenerator, sizes (32, 64, 128, 256, 512) x ratios (0.5, 1, 2), A = 3 (frcnn_la.py:185-187)
// kind 1: RetinaNet, sizes (x, int(x 2^(1/3)), int(x 2^(2/3))) for x in the same five, A = 9, index = ratio * 3 + size (retinanet_cal.py:346-351)
static int base_anchors_host(int kind, float* base) {
    const float ratios[3] = {0.5f, 1.0f, 2.0f};
    for (int l = 0; l < 5; l++) {
        const int x = 32 << l;
        const float sizes[3] = {(float)x, (float)(int)((double)x * pow(2.0, 1.0 / 3)), (float)(int)((double)x * pow(2.0, 2.0 / 3))};
        const int ns = kind == 1 ? 3 : 1;
        for (int r = 0; r < 3; r++) {
            const float hr = sqrtf(ratios[r]), wr = 1.0f / hr;
            for (int si = 0; si < ns; si++) {
                const float ws = wr * sizes[si], hs = hr * sizes[si];
                float* b = &base[((l * 3 + r) * ns + si) * 4];
                b[0] = rintf(-ws / 2.0f); b[1] = rintf(-hs / 2.0f); b[2] = rintf(ws / 2.0f); b[3] = rintf(hs / 2.0f);
            }
        }
    }
    return kind == 1 ? 9 : 3;
}
extern "C" int cald_train_anchors(cald_ctx* c, int kind, int Hp, int Wp, const int* level_hw, float* anchors_out) {
    if (!c || !level_hw || !anchors_out || kind < 0 || kind > 1) TFAIL(CALD_ERR_INVALID, "bad arguments");
    THIP(hipSetDevice(cald_internal_device(c)));
    hipStream_t st = cald_internal_stream(c);
    float base[5 * 9 * 4];
    const int A = base_anchors_host(kind, base);
    void* scratch = nullptr;
    if (int rc = cald_internal_scratch(c, sizeof(base), &scratch)) return rc;
    THIP(hipMemcpyAsync(scratch, base, sizeof(base), hipMemcpyHostToDevice, st));
    THIP(hipStreamSynchronize(st));
    long long off = 0;
    for (int l = 0; l < 5; l++) {
        const int Hl = level_hw[2 * l], Wl = level_hw[2 * l + 1];
        const long long n = (long long)Hl * Wl * A;
        hipLaunchKernelGGL(anchors_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (float4*)anchors_out, Hl, Wl, Hp / Hl, Wp / Wl,
                           (const float*)scratch + l * A * 4, A, off);
        off += n;
    }
    THIP(hipGetLastError());
    THIP(hipStreamSynchronize(st));   // the scratch holding the base anchors may be reused by the next call
    return 0;
}

// BoxCoder.encode_single (torchvision 0.8.2 _utils.py): regression targets of `proposals` towards `reference` boxes
__device__ __forceinline__ float4 box_encode_one(const float4 r, const float4 p, float wx, float wy, float ww, float wh) {
    const float ex_w = p.z - p.x, ex_h = p.w - p.y, ex_cx = p.x + 0.5f * ex_w, ex_cy = p.y + 0.5f * ex_h;
    const float gt_w = r.z - r.x, gt_h = r.w - r.y, gt_cx = r.x + 0.5f * gt_w, gt_cy = r.y + 0.5f * gt_h;
    return make_float4(wx * (gt_cx - ex_cx) / ex_w, wy * (gt_cy - ex_cy) / ex_h, ww * det_logf(gt_w / ex_w), wh * det_logf(gt_h / ex_h));
}
__global__ void box_encode_kernel(const float4* reference, const float4* proposals, int n, float wx, float wy, float ww, float wh, float4* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = box_encode_one(reference[i], proposals[i], wx, wy, ww, wh);
}
// What follows the RoI sampler on the device, in one launch: the sampled boxes gathered into RoIAlign's [R][5] rows (image index, box)
// and the regression targets of the foreground rows.  idx: the sampler's int64 block with stride `cap` between its lists
// (cald_train_roi_sample_host's outputs uploaded as they are: table rows | ground-truth rows | labels | pred_idx | pos_rows | image
// index as float32 in the first 4 * R bytes of the sixth list).
__global__ void roi_gather_kernel(const float4* __restrict__ table, const float4* __restrict__ gts, const long long* __restrict__ idx, int cap, int R, int n_pos,
                                  float wx, float wy, float ww, float wh, float* __restrict__ rois, float4* __restrict__ box_tgt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) {
        const float4 b = table[idx[i]];
        float* o = rois + (long long)i * 5;
        o[0] = reinterpret_cast<const float*>(idx + 5ll * cap)[i]; o[1] = b.x; o[2] = b.y; o[3] = b.z; o[4] = b.w;
    }
    if (i < n_pos) {
        const long long r = idx[4ll * cap + i];
        box_tgt[i] = box_encode_one(gts[idx[(long long)cap + r]], table[idx[r]], wx, wy, ww, wh);
    }
}
extern "C" int cald_train_roi_gather(cald_ctx* c, const float* table, const float* gts, const int64_t* idx, int cap, int R, int n_pos,
                                     float wx, float wy, float ww, float wh, float* rois_out, float* box_tgt_out) {
    if (!c || !table || !gts || !idx || !rois_out || (n_pos > 0 && !box_tgt_out)) TFAIL(CALD_ERR_INVALID, "null argument");
    if (R < 0 || n_pos < 0 || n_pos > R || R > cap) TFAIL(CALD_ERR_INVALID, "0 <= n_pos <= R <= cap");
    THIP(hipSetDevice(cald_internal_device(c)));
    if (R > 0) hipLaunchKernelGGL(roi_gather_kernel, dim3((R + 255) / 256), dim3(256), 0, cald_internal_stream(c), (const float4*)table, (const float4*)gts,
                                  (const long long*)idx, cap, R, n_pos, wx, wy, ww, wh, rois_out, (float4*)box_tgt_out);
    THIP(hipGetLastError());
    return 0;
}
extern "C" int cald_train_box_encode(cald_ctx* c, int n, const float* reference, const float* proposals, float wx, float wy, float ww, float wh,
                                     float* out) {
    if (!c || (n > 0 && (!reference || !proposals || !out))) TFAIL(CALD_ERR_INVALID, "null argument");
    THIP(hipSetDevice(cald_internal_device(c)));
    if (n > 0) hipLaunchKernelGGL(box_encode_kernel, dim3((n + 255) / 256), dim3(256), 0, cald_internal_stream(c), (const float4*)reference,
                                  (const float4*)proposals, n, wx, wy, ww, wh, (float4*)out);
    THIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// MultiScaleRoIAlign (7x7, sampling_ratio 2, aligned=False) on a dense batch, forward and backward
// ---------------------------------------------------------------------------------------------------------------------
struct RoiTrainArgs {
    const float* feat[4]; float* gfeat[4]; int H[4], W[4];
    int C, R;
    const float* rois;     // [R][5]: image index, x1, y1, x2, y2 (resized-image coordinates)
    float* out;            // forward: [R][49][C]
    const float* gout;     // backward: [R][49][C]
};
__device__ inline void roi_setup(const RoiTrainArgs& a, int r, RoiSample* sy, RoiSample* sx, int* n_out, int* l_out) {
    const float* rp = a.rois + (long long)r * 5;
    const float4 box = make_float4(rp[1], rp[2], rp[3], rp[4]);
    const int l = roi_level(box), tid = threadIdx.x;
    if (tid < 28) {
        const float scale = 1.0f / (float)(4 << l);
        const float x1 = box.x * scale, y1 = box.y * scale, x2 = box.z * scale, y2 = box.w * scale;
        float rw = x2 - x1; if (!(rw >= 1.0f)) rw = 1.0f;
        float rh = y2 - y1; if (!(rh >= 1.0f)) rh = 1.0f;
        const float bw = rw / 7.0f, bh = rh / 7.0f;
        if (tid < 14) sy[tid] = roi_sample(y1, bh, tid >> 1, tid & 1, a.H[l]);
        else sx[tid - 14] = roi_sample(x1, bw, (tid - 14) >> 1, (tid - 14) & 1, a.W[l]);
    }
    *n_out = (int)rp[0]; *l_out = l;
    __syncthreads();
}
// forward: a thread owns (bin, 4 consecutive channels): 16 float4 gathers in flight per bin, one float4 store
__global__ __launch_bounds__(256) void roi_align_train_kernel(RoiTrainArgs a) {
    __shared__ RoiSample sy[14], sx[14];
    const int r = blockIdx.x, tid = threadIdx.x;
    int n, l;
    roi_setup(a, r, sy, sx, &n, &l);
    const int Hf = a.H[l], Wf = a.W[l], Cq = a.C >> 2;
    const float4* f = reinterpret_cast<const float4*>(a.feat[l]) + (long long)n * Hf * Wf * Cq;
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
                if (!(Y.valid && X.valid)) continue;
                const float w1 = Y.h * X.h, w2 = Y.h * X.l, w3 = Y.l * X.h, w4 = Y.l * X.l;
                const float4 v1 = f[(long long)(Y.lo * Wf + X.lo) * Cq + q], v2 = f[(long long)(Y.lo * Wf + X.hi) * Cq + q];
                const float4 v3 = f[(long long)(Y.hi * Wf + X.lo) * Cq + q], v4 = f[(long long)(Y.hi * Wf + X.hi) * Cq + q];
                acc.x = acc.x + (((w1 * v1.x + w2 * v2.x) + w3 * v3.x) + w4 * v4.x);
                acc.y = acc.y + (((w1 * v1.y + w2 * v2.y) + w3 * v3.y) + w4 * v4.y);
                acc.z = acc.z + (((w1 * v1.z + w2 * v2.z) + w3 * v3.z) + w4 * v4.z);
                acc.w = acc.w + (((w1 * v1.w + w2 * v2.w) + w3 * v3.w) + w4 * v4.w);
            }
        }
        reinterpret_cast<float4*>(a.out)[(long long)r * 49 * Cq + idx] = make_float4(acc.x / 4.0f, acc.y / 4.0f, acc.z / 4.0f, acc.w / 4.0f);
    }
}
// backward: a thread owns (bin, ONE channel), consecutive lanes = consecutive channels, so that every atomic instruction of a
// wavefront lands on 256 contiguous bytes (two cache lines) of one feature pixel
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(RoiTrainArgs a) {
    __shared__ RoiSample sy[14], sx[14];
    const int r = blockIdx.x, tid = threadIdx.x;
    int n, l;
    roi_setup(a, r, sy, sx, &n, &l);
    const int Hf = a.H[l], Wf = a.W[l], C = a.C;
    float* gf = a.gfeat[l] + (long long)n * Hf * Wf * C;
    for (int idx = tid; idx < 49 * C; idx += 256) {
        const int bin = idx / C, c = idx - bin * C;
        const int ph = bin / 7, pw = bin - ph * 7;
        const float go = a.gout[(long long)r * 49 * C + idx] * 0.25f;
#pragma unroll
        for (int iy = 0; iy < 2; iy++) {
            const RoiSample Y = sy[ph * 2 + iy];
#pragma unroll
            for (int ix = 0; ix < 2; ix++) {
                const RoiSample X = sx[pw * 2 + ix];
                if (!(Y.valid && X.valid)) continue;
                unsafeAtomicAdd(gf + (long long)(Y.lo * Wf + X.lo) * C + c, Y.h * X.h * go);
                unsafeAtomicAdd(gf + (long long)(Y.lo * Wf + X.hi) * C + c, Y.h * X.l * go);
                unsafeAtomicAdd(gf + (long long)(Y.hi * Wf + X.lo) * C + c, Y.l * X.h * go);
                unsafeAtomicAdd(gf + (long long)(Y.hi * Wf + X.hi) * C + c, Y.l * X.l * go);
            }
        }
    }
}
static int roi_args(RoiTrainArgs& a, const float* const* feats, float* const* gfeats, const int* level_hw, int C, int R, const float* rois) {
    if (C % 4 || R < 1 || !rois || !level_hw) return cald_internal_fail(CALD_ERR_INVALID, "bad RoIAlign arguments");
    memset(&a, 0, sizeof(a));
    for (int l = 0; l < 4; l++) { a.feat[l] = feats ? feats[l] : nullptr; a.gfeat[l] = gfeats ? gfeats[l] : nullptr; a.H[l] = level_hw[2 * l]; a.W[l] = level_hw[2 * l + 1]; }
    a.C = C; a.R = R; a.rois = rois;
    return 0;
}
extern "C" int cald_train_roi_align(cald_ctx* c, const float* const* feats, const int* level_hw, int C, int R, const float* rois, float* out) {
    if (!c || !feats || !out) TFAIL(CALD_ERR_INVALID, "null argument");
    THIP(hipSetDevice(cald_internal_device(c)));
    RoiTrainArgs a; if (int rc = roi_args(a, feats, nullptr, level_hw, C, R, rois)) return rc;
    a.out = out;
    hipLaunchKernelGGL(roi_align_train_kernel, dim3(R), dim3(256), 0, cald_internal_stream(c), a);
    THIP(hipGetLastError());
    return 0;
}
// Deterministic scatter: contributions are accumulated as 64-bit FIXED-POINT integers (integer addition is associative, so the
// arrival order of the atomics does not matter), scaled by 2^k with k chosen from max|gout| so that the largest single term is
// 2^40 (resolution 2^-40 of it; 2^21 such terms still fit), then converted and added to the float gradient.
__global__ void absmax_kernel(const float* x, long long n, unsigned* out_bits) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.0f;
    for (long long j = i; j < n; j += (long long)gridDim.x * blockDim.x) { const float v = fabsf(x[j]); if (v > m) m = v; }   // NaN never wins
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(out_bits, __float_as_uint(m));
}
__device__ inline double fixed_scale(unsigned max_bits) {
    const int e = (int)((max_bits >> 23) & 255) - 127;          // max = 1.f x 2^e  (0 if the tensor is all zero)
    return max_bits ? ldexp(1.0, 40 - (e + 1)) : 1.0;
}
struct RoiAccPtrs { long long* p[4]; };
__global__ __launch_bounds__(256) void roi_align_bwd_fixed_kernel(RoiTrainArgs a, RoiAccPtrs acc, const unsigned* max_bits) {
    __shared__ RoiSample sy[14], sx[14];
    const int r = blockIdx.x, tid = threadIdx.x;
    int n, l;
    roi_setup(a, r, sy, sx, &n, &l);
    const int Hf = a.H[l], Wf = a.W[l], C = a.C;
    unsigned long long* gf = reinterpret_cast<unsigned long long*>(acc.p[l]) + (long long)n * Hf * Wf * C;
    const double scale = fixed_scale(*max_bits);
    for (int idx = tid; idx < 49 * C; idx += 256) {
        const int bin = idx / C, c = idx - bin * C;
        const int ph = bin / 7, pw = bin - ph * 7;
        const float go = a.gout[(long long)r * 49 * C + idx] * 0.25f;
        if (!(go == go) || fabsf(go) == INFINITY) continue;          // non-finite gradients are reported by the loss check, not spread
#pragma unroll
        for (int iy = 0; iy < 2; iy++) {
            const RoiSample Y = sy[ph * 2 + iy];
#pragma unroll
            for (int ix = 0; ix < 2; ix++) {
                const RoiSample X = sx[pw * 2 + ix];
                if (!(Y.valid && X.valid)) continue;
                const float w[4] = {Y.h * X.h, Y.h * X.l, Y.l * X.h, Y.l * X.l};
                const long long o[4] = {(long long)(Y.lo * Wf + X.lo) * C + c, (long long)(Y.lo * Wf + X.hi) * C + c,
                                        (long long)(Y.hi * Wf + X.lo) * C + c, (long long)(Y.hi * Wf + X.hi) * C + c};
#pragma unroll
                for (int k = 0; k < 4; k++)
                    atomicAdd(gf + o[k], (unsigned long long)(long long)rint((double)(w[k] * go) * scale));      // two's complement wrap = signed add
            }
        }
    }
}
// C == 256: the 16 contributions of a (bin, channel) -- 2 x 2 samples x 4 bilinear taps -- fall on far fewer than 16 feature pixels
// (the two samples of a bin are half a bin apart: usually they share a pixel row / column or sit on the same ones).  The integer
// contributions are summed per distinct pixel in registers first -- exact, the accumulation is integer -- and one atomic goes out per
// pixel.  Which samples share rows / columns is the same for the whole workgroup (one bin per pass), so the sharing pattern is a
// scalar branch:  P 0: slots (lo0, hi0, lo1, hi1);  P 1: lo1 == hi0 -> (lo0, hi0, hi1);  P 2: sample 1 on sample 0's pair -> (lo0, hi0).
template <int PR, int PC>
__device__ __forceinline__ void roi_bwd_bin(unsigned long long* const gf, const int Wf, const int c, const float go, const double scale,
                                            const RoiSample Y0, const RoiSample Y1, const RoiSample X0, const RoiSample X1) {
    constexpr int NRS = PR == 0 ? 4 : (PR == 1 ? 3 : 2), NCS = PC == 0 ? 4 : (PC == 1 ? 3 : 2);
    long long acc[NRS][NCS];
#pragma unroll
    for (int i = 0; i < NRS; i++)
#pragma unroll
        for (int j = 0; j < NCS; j++) acc[i][j] = 0;
#pragma unroll
    for (int iy = 0; iy < 2; iy++) {
        const RoiSample Y = iy ? Y1 : Y0;
        const int rl = iy == 0 ? 0 : (PR == 0 ? 2 : (PR == 1 ? 1 : 0)), rh = iy == 0 ? 1 : (PR == 0 ? 3 : (PR == 1 ? 2 : 1));
#pragma unroll
        for (int ix = 0; ix < 2; ix++) {
            const RoiSample X = ix ? X1 : X0;
            const int cl = ix == 0 ? 0 : (PC == 0 ? 2 : (PC == 1 ? 1 : 0)), ch = ix == 0 ? 1 : (PC == 0 ? 3 : (PC == 1 ? 2 : 1));
            if (!(Y.valid && X.valid)) continue;
            acc[rl][cl] += (long long)rint((double)((Y.h * X.h) * go) * scale);
            acc[rl][ch] += (long long)rint((double)((Y.h * X.l) * go) * scale);
            acc[rh][cl] += (long long)rint((double)((Y.l * X.h) * go) * scale);
            acc[rh][ch] += (long long)rint((double)((Y.l * X.l) * go) * scale);
        }
    }
    const int rows[4] = {Y0.lo, Y0.hi, PR == 1 ? Y1.hi : Y1.lo, Y1.hi};
    const int cols[4] = {X0.lo, X0.hi, PC == 1 ? X1.hi : X1.lo, X1.hi};
#pragma unroll
    for (int i = 0; i < NRS; i++)
#pragma unroll
        for (int j = 0; j < NCS; j++)
            if (acc[i][j]) atomicAdd(gf + (long long)(rows[i] * Wf + cols[j]) * 256 + c, (unsigned long long)acc[i][j]);
}
__global__ __launch_bounds__(256) void roi_align_bwd_fixed_merge_kernel(RoiTrainArgs a, RoiAccPtrs acc, const unsigned* max_bits) {
    __shared__ RoiSample sy[14], sx[14];
    const int r = blockIdx.x, tid = threadIdx.x;
    int n, l;
    roi_setup(a, r, sy, sx, &n, &l);
    const int Hf = a.H[l], Wf = a.W[l];
    unsigned long long* gf = reinterpret_cast<unsigned long long*>(acc.p[l]) + (long long)n * Hf * Wf * 256;
    const double scale = fixed_scale(*max_bits);
    for (int bin = 0; bin < 49; bin++) {
        const int ph = bin / 7, pw = bin - ph * 7;
        float go = a.gout[((long long)r * 49 + bin) * 256 + tid] * 0.25f;
        if (!(go == go) || fabsf(go) == INFINITY) go = 0.0f;          // non-finite gradients are reported by the loss check, not spread
        const RoiSample Y0 = sy[ph * 2], Y1 = sy[ph * 2 + 1], X0 = sx[pw * 2], X1 = sx[pw * 2 + 1];
        const int y0l = __builtin_amdgcn_readfirstlane(Y0.lo), y0h = __builtin_amdgcn_readfirstlane(Y0.hi);
        const int y1l = __builtin_amdgcn_readfirstlane(Y1.lo), y1h = __builtin_amdgcn_readfirstlane(Y1.hi);
        const int x0l = __builtin_amdgcn_readfirstlane(X0.lo), x0h = __builtin_amdgcn_readfirstlane(X0.hi);
        const int x1l = __builtin_amdgcn_readfirstlane(X1.lo), x1h = __builtin_amdgcn_readfirstlane(X1.hi);
        const int pr = (y1l == y0l && y1h == y0h) ? 2 : (y1l == y0h ? 1 : 0);
        const int pc = (x1l == x0l && x1h == x0h) ? 2 : (x1l == x0h ? 1 : 0);
        switch (pr * 3 + pc) {
            case 0: roi_bwd_bin<0, 0>(gf, Wf, tid, go, scale, Y0, Y1, X0, X1); break;
            case 1: roi_bwd_bin<0, 1>(gf, Wf, tid, go, scale, Y0, Y1, X0, X1); break;
            case 2: roi_bwd_bin<0, 2>(gf, Wf, tid, go, scale, Y0, Y1, X0, X1); break;
            case 3: roi_bwd_bin<1, 0>(gf, Wf, tid, go, scale, Y0, Y1, X0, X1); break;
            case 4: roi_bwd_bin<1, 1>(gf, Wf, tid, go, scale, Y0, Y1, X0, X1); break;
            case 5: roi_bwd_bin<1, 2>(gf, Wf, tid, go, scale, Y0, Y1, X0, X1); break;
            case 6: roi_bwd_bin<2, 0>(gf, Wf, tid, go, scale, Y0, Y1, X0, X1); break;
            case 7: roi_bwd_bin<2, 1>(gf, Wf, tid, go, scale, Y0, Y1, X0, X1); break;
            default: roi_bwd_bin<2, 2>(gf, Wf, tid, go, scale, Y0, Y1, X0, X1); break;
        }
    }
}
// two accumulators per thread; the scale is a power of two, so multiplying by its reciprocal is the exact division
__global__ void fixed_to_float_kernel(const long long* acc, float* g, long long n, const unsigned* max_bits) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    const double inv = 1.0 / fixed_scale(*max_bits);
    if (i + 1 < n) {
        const longlong2 v = *reinterpret_cast<const longlong2*>(acc + i);
        if (v.x | v.y) {
            float2 o = *reinterpret_cast<float2*>(g + i);
            if (v.x) o.x = o.x + (float)((double)v.x * inv);
            if (v.y) o.y = o.y + (float)((double)v.y * inv);
            *reinterpret_cast<float2*>(g + i) = o;
        }
    } else {
        const long long v = acc[i];
        if (v) g[i] = g[i] + (float)((double)v * inv);
    }
}
/* gfeats[l] += scatter of gout through the bilinear weights.  Deterministic (fixed-point accumulation, see above); CALD_ROI_BWD_FLOAT=1
 * selects plain float atomics (summation in arrival order). */
extern "C" int cald_train_roi_align_bwd(cald_ctx* c, int N, float* const* gfeats, const int* level_hw, int C, int R, const float* rois, const float* gout) {
    if (!c || !gfeats || !gout || N < 1) TFAIL(CALD_ERR_INVALID, "bad arguments");
    THIP(hipSetDevice(cald_internal_device(c)));
    hipStream_t st = cald_internal_stream(c);
    RoiTrainArgs a; if (int rc = roi_args(a, (const float* const*)gfeats, gfeats, level_hw, C, R, rois)) return rc;
    a.gout = gout;
    static const bool float_atomics = getenv("CALD_ROI_BWD_FLOAT") && atoi(getenv("CALD_ROI_BWD_FLOAT")) != 0;
    if (float_atomics) {
        hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(R), dim3(256), 0, st, a);
        THIP(hipGetLastError());
        return 0;
    }
    long long n[4], total = 0;
    for (int l = 0; l < 4; l++) { n[l] = (long long)N * a.H[l] * a.W[l] * C; total += (n[l] + 1) & ~1ll; }
    void* scratch = nullptr;
    if (int rc = cald_internal_scratch(c, (size_t)total * 8 + 256, &scratch)) return rc;
    long long* acc = (long long*)scratch;
    unsigned* d_max = (unsigned*)(acc + total);
    RoiAccPtrs ptrs; long long off = 0;
    for (int l = 0; l < 4; l++) { ptrs.p[l] = acc + off; off += (n[l] + 1) & ~1ll; }     // 16-byte aligned level bases
    THIP(hipMemsetAsync(acc, 0, (size_t)total * 8 + 4, st));
    const long long ng = (long long)R * 49 * C;
    hipLaunchKernelGGL(absmax_kernel, dim3(1024), dim3(256), 0, st, gout, ng, d_max);
    static const bool merge = !(getenv("CALD_ROI_BWD_MERGE") && atoi(getenv("CALD_ROI_BWD_MERGE")) == 0);
    if (merge && C == 256) hipLaunchKernelGGL(roi_align_bwd_fixed_merge_kernel, dim3(R), dim3(256), 0, st, a, ptrs, (const unsigned*)d_max);
    else hipLaunchKernelGGL(roi_align_bwd_fixed_kernel, dim3(R), dim3(256), 0, st, a, ptrs, (const unsigned*)d_max);
    for (int l = 0; l < 4; l++)
        hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((n[l] + 511) / 512)), dim3(256), 0, st, (const long long*)ptrs.p[l], gfeats[l], n[l], (const unsigned*)d_max);
    THIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// losses: value and gradient in one pass.  Each kernel is ONE workgroup (the row counts are a few thousand), so the sums are
// taken in a fixed order.  `gscale` multiplies the gradient (the upstream gradient of the scalar loss, normally 1).
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline float block_sum_256(float v, float* red) {
    const int tid = threadIdx.x;
    red[tid] = v; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const float r = red[0]; __syncthreads();
    return r;
}
// F.cross_entropy(logits[R][C], labels) (mean over rows); grad[r][c] = (softmax - onehot) / R; rows have stride ld
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* logits, const long long* labels, int R, int C, int ld, float gscale,
                                                         float* loss, float* grad) {
    __shared__ float red[256];
    float local = 0.0f;
    for (int r = threadIdx.x; r < R; r += 256) {
        const float* z = logits + (long long)r * ld;
        float m = z[0];
        for (int k = 1; k < C; k++) m = fmaxf(m, z[k]);
        float s = 0.0f;
        for (int k = 0; k < C; k++) s += det_expf(z[k] - m);
        const int y = (int)labels[r];
        local += (det_logf(s) + m) - z[y];
        if (grad) {
            float* g = grad + (long long)r * ld;
            const float inv = gscale / (float)R;
            for (int k = 0; k < C; k++) g[k] = (det_expf(z[k] - m) / s - (k == y ? 1.0f : 0.0f)) * inv;
        }
    }
    const float tot = block_sum_256(local, red);
    if (threadIdx.x == 0) *loss = tot / (float)R;
}
extern "C" int cald_train_softmax_ce(cald_ctx* c, int R, int C, int ld, const float* logits, const int64_t* labels, float gscale, float* loss_out,
                                     float* grad_out) {
    if (!c || !logits || !labels || !loss_out || R < 1 || C < 1) TFAIL(CALD_ERR_INVALID, "bad arguments");
    THIP(hipSetDevice(cald_internal_device(c)));
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(256), 0, cald_internal_stream(c), logits, (const long long*)labels, R, C, ld, gscale, loss_out, grad_out);
    THIP(hipGetLastError());
    return 0;
}
// det_utils.smooth_l1_loss(pred, target, beta, size_average=False) / denom over n gathered 4-vectors: pred 4-vector i starts at
// float offset idx[i] of `pred` (and of `grad`, which the caller has zeroed).  beta = 0 is the plain L1 loss (retinanet_cal.py:217);
// weights (optional, one per 4-vector) multiply each vector's term (per-image 1 / num_foreground of RetinaNet).
__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* pred, const long long* idx, const float* target, int n, float beta, float denom,
                                                        const float* weights, float gscale, float* loss, float* grad) {
    __shared__ float red[256];
    float local = 0.0f;
    for (int e = threadIdx.x; e < 4 * n; e += 256) {
        const long long o = idx[e >> 2] + (e & 3);
        const float d = pred[o] - target[e], ad = fabsf(d);
        const float w = weights ? weights[e >> 2] : 1.0f;
        local += w * (ad < beta ? 0.5f * d * d / beta : ad - 0.5f * beta);
        if (grad) grad[o] = (ad < beta ? d / beta : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f))) * (w * gscale / denom);
    }
    const float tot = block_sum_256(local, red);
    if (threadIdx.x == 0) *loss = tot / denom;
}
extern "C" int cald_train_smooth_l1(cald_ctx* c, int n, const float* pred, const int64_t* idx, const float* target, float beta, float denom,
                                    const float* weights, float gscale, float* loss_out, float* grad) {
    if (!c || !loss_out || (n > 0 && (!pred || !idx || !target))) TFAIL(CALD_ERR_INVALID, "bad arguments");
    THIP(hipSetDevice(cald_internal_device(c)));
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(1), dim3(256), 0, cald_internal_stream(c), pred, (const long long*)idx, target, n, beta, denom, weights, gscale, loss_out, grad);
    THIP(hipGetLastError());
    return 0;
}
// F.binary_cross_entropy_with_logits(x[idx], y) (mean over n gathered logits)
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* x, const long long* idx, const float* y, int n, float gscale, float* loss, float* grad) {
    __shared__ float red[256];
    float local = 0.0f;
    for (int e = threadIdx.x; e < n; e += 256) {
        const long long o = idx[e];
        const float z = x[o], t = y[e];
        // max(z, 0) - z * t + log(1 + exp(-|z|))
        local += ((z > 0.0f ? z : 0.0f) - z * t) + det_logf(1.0f + det_expf(-fabsf(z)));
        if (grad) grad[o] = (det_sigmoidf(z) - t) * (gscale / (float)n);
    }
    const float tot = block_sum_256(local, red);
    if (threadIdx.x == 0) *loss = tot / (float)n;
}
extern "C" int cald_train_bce_logits(cald_ctx* c, int n, const float* logits, const int64_t* idx, const float* labels, float gscale, float* loss_out,
                                     float* grad) {
    if (!c || !loss_out || n < 1 || !logits || !idx || !labels) TFAIL(CALD_ERR_INVALID, "bad arguments");
    THIP(hipSetDevice(cald_internal_device(c)));
    hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, cald_internal_stream(c), logits, (const long long*)idx, labels, n, gscale, loss_out, grad);
    THIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// input side of the training forward: GeneralizedRCNNTransform (normalize, bilinear resize, zero-pad to the batch size), the
// stem's max-pool, LastLevelMaxPool -- the inference kernels of elementwise.hip on a dense batch
// ---------------------------------------------------------------------------------------------------------------------
// images[i]: device uint8 [H_i][W_i][3]; hw = host {H_0, W_0, Hr_0, Wr_0, ...} (source size, resized size); remainders[i]: optional
// device float [3][H_i][W_i] added to image / 255 (inputs that are not on the uint8 grid), or null.  out [N][Hp][Wp][4] (channel 3 = 0).
extern "C" int cald_train_preprocess(cald_ctx* c, int N, const uint8_t* const* images, const float* const* remainders, const int* hw, int Hp, int Wp,
                                     float* out) {
    if (!c || !images || !hw || !out) TFAIL(CALD_ERR_INVALID, "null argument");
    if (N < 1 || N > CALD_MAX_VIEWS) TFAIL(CALD_ERR_INVALID, "N must be 1..%d", CALD_MAX_VIEWS);
    THIP(hipSetDevice(cald_internal_device(c)));
    SEG_ENTRY();
    hipStream_t st = cald_internal_stream(c);
    std::vector<ViewDesc> hv(N);
    memset(hv.data(), 0, sizeof(ViewDesc) * N);
    for (int v = 0; v < N; v++) {
        hv[v].src = images[v]; hv[v].H = hw[4 * v]; hv[v].W = hw[4 * v + 1]; hv[v].Hr = hw[4 * v + 2]; hv[v].Wr = hw[4 * v + 3];
        hv[v].Ho = hv[v].H; hv[v].Wo = hv[v].W; hv[v].noise = remainders ? remainders[v] : nullptr;
        if (hv[v].Hr > Hp || hv[v].Wr > Wp) TFAIL(CALD_ERR_INVALID, "resized image %d exceeds the padded batch size", v);
    }
    // the descriptors travel through a pinned staging ring: the call returns without waiting for the stream (it used to stop the host
    // twice, and with it the start of every training step until the previous step had left the GPU)
    const void* dv = nullptr; int slot = 0;
    if (int rc = stage_upload(c, hv.data(), sizeof(ViewDesc) * N, st, &dv, &slot)) return rc;
    const LevelSeg* s0;
    if (int rc = dense_seg(c, N, Hp, Wp, &s0)) return rc;
    launch_preprocess((const ViewDesc*)dv, s0, out, N, Hp * Wp, st);
    THIP(hipGetLastError());
    if (int rc = stage_consumed(c, slot, st)) return rc;
    static const bool sync_env = getenv("CALD_TRAIN_PREPROCESS_SYNC") && atoi(getenv("CALD_TRAIN_PREPROCESS_SYNC"));   // A/B switch
    if (sync_env) THIP(hipStreamSynchronize(st));
    return 0;
}
/* max_pool2d(3, 2, 1): in [N][H][W][C] -> out [N][(H-1)/2+1][(W-1)/2+1][C] */
extern "C" int cald_train_maxpool(cald_ctx* c, int N, int H, int W, int C, const float* in, float* out) {
    if (!c || !in || !out || C % 4) TFAIL(CALD_ERR_INVALID, "bad arguments");
    THIP(hipSetDevice(cald_internal_device(c)));
    SEG_ENTRY();
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const LevelSeg *si, *so;
    if (int rc = dense_seg(c, N, H, W, &si)) return rc;
    if (int rc = dense_seg(c, N, Ho, Wo, &so)) return rc;
    launch_maxpool(in, out, si, so, C, N, Ho * Wo, cald_internal_stream(c));
    THIP(hipGetLastError());
    return 0;
}
/* max_pool2d(1, 2, 0) (LastLevelMaxPool): every second pixel */
extern "C" int cald_train_subsample2(cald_ctx* c, int N, int H, int W, int C, const float* in, float* out) {
    if (!c || !in || !out || C % 4) TFAIL(CALD_ERR_INVALID, "bad arguments");
    THIP(hipSetDevice(cald_internal_device(c)));
    SEG_ENTRY();
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const LevelSeg *si, *so;
    if (int rc = dense_seg(c, N, H, W, &si)) return rc;
    if (int rc = dense_seg(c, N, Ho, Wo, &so)) return rc;
    launch_subsample2(in, out, si, so, C, N, Ho * Wo, cald_internal_stream(c));
    THIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// RetinaNet classification loss (retinanet_cal.py:100-133): torchvision.ops.sigmoid_focal_loss(alpha 0.25, gamma 2, 'sum') over the
// anchors whose match is not BETWEEN_THRESHOLDS, / max(1, #foreground) per image, mean over images.
//   logits: five level blocks in one buffer, level l = [N][pix_l][ld] with channel a * K + k; anchor order (level, y, x, a).
//   target(n, anchor, k) = matched >= 0 and gt_labels[gt_off[n] + matched] == k;  img_weight[n] = 1 / (max(1, #fg_n) * N).
// One thread per (image, anchor); block partial sums are added in a fixed order by the second kernel.
// ---------------------------------------------------------------------------------------------------------------------
struct FocalArgs {
    const float* logits; float* grad; const int* matched; const long long* gt_labels; const int* gt_off; const float* img_weight;
    long long lvl_anchor0[6], lvl_off[5]; int lvl_pix[5];
    int N, A, K, ld; long long A_tot; float alpha, gscale;
};
__global__ __launch_bounds__(256) void focal_loss_kernel(FocalArgs a, float* partial) {
    __shared__ float red[256];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float local = 0.0f;
    if (i < a.N * a.A_tot) {
        const int n = (int)(i / a.A_tot); const long long an = i - (long long)n * a.A_tot;
        const int m = a.matched[i];
        if (m != -2) {
            int l = 0;
            while (l < 4 && an >= a.lvl_anchor0[l + 1]) l++;
            const long long rel = an - a.lvl_anchor0[l];
            const long long pix = rel / a.A; const int aa = (int)(rel - pix * a.A);
            const long long base = a.lvl_off[l] + ((long long)n * a.lvl_pix[l] + pix) * a.ld + (long long)aa * a.K;
            const int cls = m >= 0 ? (int)a.gt_labels[a.gt_off[n] + m] : -1;
            const float w = a.img_weight[n];
            for (int k = 0; k < a.K; k++) {
                const float x = a.logits[base + k];
                const float lse = det_logf(1.0f + det_expf(-fabsf(x)));          // log(1 + exp(-|x|))
                const float logp = -((x < 0.0f ? -x : 0.0f) + lse);               // log sigmoid(x)
                const float log1mp = -((x > 0.0f ? x : 0.0f) + lse);              // log (1 - sigmoid(x))
                const float p = det_sigmoidf(x), q = 1.0f - p;
                float loss, g;
                if (k == cls) { loss = -a.alpha * q * q * logp; g = a.alpha * q * q * (2.0f * p * logp - q); }
                else { loss = -(1.0f - a.alpha) * p * p * log1mp; g = (1.0f - a.alpha) * p * p * (p - 2.0f * q * log1mp); }
                local += w * loss;
                if (a.grad) a.grad[base + k] = g * w * a.gscale;
            }
        }
    }
    const float tot = block_sum_256(local, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* partial, int n, float* out) {
    __shared__ float red[256];
    float local = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) local += partial[i];
    const float tot = block_sum_256(local, red);
    if (threadIdx.x == 0) *out = tot;
}
extern "C" int cald_train_focal_loss(cald_ctx* c, int N, const int* level_pix, int A, int K, int ld, const float* logits, const int* matched,
                                     const int64_t* gt_labels, const int* gt_off, const float* img_weight, float alpha, float gscale,
                                     float* loss_out, float* grad) {
    if (!c || !level_pix || !logits || !matched || !gt_labels || !gt_off || !img_weight || !loss_out) TFAIL(CALD_ERR_INVALID, "null argument");
    if (N < 1 || A < 1 || K < 1 || ld < A * K) TFAIL(CALD_ERR_INVALID, "bad geometry");
    THIP(hipSetDevice(cald_internal_device(c)));
    hipStream_t st = cald_internal_stream(c);
    FocalArgs a; memset(&a, 0, sizeof(a));
    a.logits = logits; a.grad = grad; a.matched = matched; a.gt_labels = (const long long*)gt_labels; a.gt_off = gt_off; a.img_weight = img_weight;
    long long an = 0, off = 0;
    for (int l = 0; l < 5; l++) {
        a.lvl_anchor0[l] = an; a.lvl_off[l] = off; a.lvl_pix[l] = level_pix[l];
        an += (long long)level_pix[l] * A; off += (long long)N * level_pix[l] * ld;
    }
    a.lvl_anchor0[5] = an; a.A_tot = an; a.N = N; a.A = A; a.K = K; a.ld = ld; a.alpha = alpha; a.gscale = gscale;
    const long long total = (long long)N * an;
    const int blocks = (int)((total + 255) / 256);
    void* scratch = nullptr;
    if (int rc = cald_internal_scratch(c, (size_t)blocks * 4 + 64, &scratch)) return rc;
    hipLaunchKernelGGL(focal_loss_kernel, dim3(blocks), dim3(256), 0, st, a, (float*)scratch);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, (const float*)scratch, blocks, loss_out);
    THIP(hipGetLastError());
    return 0;
}
