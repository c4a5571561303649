// elementwise.hip -- HBM-bound kernels of the sweep front end: view construction (augmentation +
// detector transform), PIL-exact resize for the `smaller_resize` augmentation, max-pool, and the
// FPN 'pool' level.  All are coalesced NHWC / float4 kernels; none touches MFMA.
//
// Reference call sites: cald_train.py:107 (to_tensor), :124-179 (flip / cut_out / resize views),
// cald/cald_helper.py:23-30, :47-53, :88-132; detector transform = torchvision
// GeneralizedRCNNTransform constructed at detection/frcnn_la.py:230-234.
#include "common.h"
#include "kernels.h"
#include "h16.h"

// ---------------------------------------------------------------------------------------------
// One view: uint8 HWC source -> (flip | cutout rectangles) -> /255 -> (x-mean)/std -> bilinear
// resize (align_corners=False, scale=in/out) -> zero pad to a multiple of 32.  NHWC, 4 channels.
// grid = (ceil(maxHp*maxWp/256), V)
// ---------------------------------------------------------------------------------------------
// cald_helper.ColorSwap (cald_helper.py:56-62): image[perms[k]] -> output channel c reads source channel perms[k][c]
__device__ inline int color_perm(int k, int c) {
    // perms = ((0,1,2), (0,2,1), (1,0,2), (1,2,0), (2,0,1), (2,1,0)), packed 2 bits per entry
    const unsigned packed[6] = {0u | (1u << 2) | (2u << 4), 0u | (2u << 2) | (1u << 4), 1u | (0u << 2) | (2u << 4),
                                1u | (2u << 2) | (0u << 4), 2u | (0u << 2) | (1u << 4), 2u | (1u << 2) | (0u << 4)};
    return (int)((packed[k] >> (2 * c)) & 3u);
}
__global__ __launch_bounds__(256) void preprocess_kernel(const ViewDesc* views, const LevelSeg* seg0, float* out) {
    // (byte / 255 - mean) / std has 256 x 3 possible values: two IEEE divisions per tap and channel (24 per output pixel) become one
    // LDS lookup of the same two divisions done once per workgroup -- the same bits.  (Views with additive noise divide per tap.)
    __shared__ float s_norm[3][256];
    {
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
        for (int c = 0; c < 3; c++) s_norm[c][threadIdx.x] = ((float)threadIdx.x / 255.0f - mean[c]) / stdv[c];
    }
    __syncthreads();
    const int v = blockIdx.y;
    const ViewDesc& vd = views[v];   // by reference: a by-value copy puts rects[] (dynamically indexed) in scratch
    const LevelSeg s = seg0[v];
    const int Hp = s.H, Wp = s.W;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= Hp * Wp) return;
    const int y = pix / Wp, x = pix - y * Wp;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y < vd.Hr && x < vd.Wr) {
        const int H = vd.H, W = vd.W, swap = vd.swap;
        const float sh = (float)H / (float)vd.Hr, sw = (float)W / (float)vd.Wr;
        float fy = sh * ((float)y + 0.5f) - 0.5f; if (fy < 0.0f) fy = 0.0f;
        float fx = sw * ((float)x + 0.5f) - 0.5f; if (fx < 0.0f) fx = 0.0f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - (float)y0, hy = 1.0f - ly, lx = fx - (float)x0, hx = 1.0f - lx;
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
        float val[2][2][3];
        const int nrect = vd.nrect;
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int yy = a ? y1 : y0;
            bool cut[2];
            int sxb[2];
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const int xx = b ? x1 : x0;
                bool ct = false;
                for (int r = 0; r < nrect; r++) {
                    const int* q = vd.rects + 4 * r;
                    ct |= (xx >= q[0] && xx < q[2] && yy >= q[1] && yy < q[3]);
                }
                cut[b] = ct;
                sxb[b] = vd.flip ? (W - 1 - xx) : xx;
            }
            if (vd.noise) {
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const uint8_t* p = vd.src + ((long long)yy * W + sxb[b]) * 3;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        float u = cut[b] ? 0.0f : (float)p[color_perm(swap, c)] / 255.0f;
                        u = u + vd.noise[((long long)c * H + yy) * W + sxb[b]];
                        val[a][b][c] = (u - mean[c]) / stdv[c];
                    }
                }
            } else {
                // the two taps of a row are adjacent pixels = six consecutive bytes: read as the (up to three) ALIGNED dwords that hold
                // them -- a dword that holds a requested byte lies inside the image buffer's pages -- and shifted into place, instead of
                // six byte loads (the kernel was bound by its twelve byte-load instructions per pixel, not by HBM)
                const int lo = sxb[0] < sxb[1] ? sxb[0] : sxb[1];
                const int nb = x1 != x0 ? 6 : 3;
                const uintptr_t ad = (uintptr_t)(vd.src + ((long long)yy * W + lo) * 3);
                const int bsh = (int)(ad & 3);
                const unsigned* q = reinterpret_cast<const unsigned*>(ad - bsh);
                const unsigned d0 = q[0];
                const unsigned d1 = bsh + nb > 4 ? q[1] : 0u;
                const unsigned d2 = bsh + nb > 8 ? q[2] : 0u;
                const unsigned w0 = __builtin_amdgcn_alignbyte(d1, d0, bsh), w1 = __builtin_amdgcn_alignbyte(d2, d1, bsh);
                const unsigned tap_lo = w0, tap_hi = (w0 >> 24) | (w1 << 8);          // three bytes each, in the low 24 bits
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const unsigned t = sxb[b] == lo ? tap_lo : tap_hi;
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        val[a][b][c] = s_norm[c][cut[b] ? 0 : (int)((t >> (8 * color_perm(swap, c))) & 255u)];
                }
            }
        }
        float r3[3];
#pragma unroll
        for (int c = 0; c < 3; c++)
            r3[c] = hy * (hx * val[0][0][c] + lx * val[0][1][c]) + ly * (hx * val[1][0][c] + lx * val[1][1][c]);
        o = make_float4(r3[0], r3[1], r3[2], 0.0f);
    }
    reinterpret_cast<float4*>(out + s.pix_off * 4)[pix] = o;
}

void launch_preprocess(const ViewDesc* views, const LevelSeg* seg0, float* out, int V, int max_pix, hipStream_t st) {
    dim3 grid((max_pix + 255) / 256, V);
    hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, st, views, seg0, out);
}

// ---------------------------------------------------------------------------------------------
// PIL Image.resize(BILINEAR) on 8-bit RGB: two fixed-point passes (Pillow Resample.c algorithm).
// Coefficients (bounds, kk) are computed on the host in double and int32, exactly as Pillow does.
// ---------------------------------------------------------------------------------------------
#define PIL_BITS 22
__device__ inline uint8_t pil_clip8(int v) {
    v >>= PIL_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
// horizontal: src [H][W][3] -> dst [H][ow][3]; grid = (ceil(ow*3/256), H)
__global__ __launch_bounds__(256) void pil_horizontal_kernel(const uint8_t* src, int H, int W, uint8_t* dst, int ow,
                                                             const int* bounds, const int* kk, int ksize) {
    const int y = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ow * 3) return;
    const int xx = e / 3, c = e - xx * 3;
    const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
    const int* k = kk + (long long)xx * ksize;
    int ss = 1 << (PIL_BITS - 1);
    const uint8_t* row = src + (long long)y * W * 3;
    for (int x = 0; x < xmax; x++) ss += (int)row[(x + xmin) * 3 + c] * k[x];
    dst[((long long)y * ow + xx) * 3 + c] = pil_clip8(ss);
}
// vertical: src [H][W][3] -> dst [oh][W][3]; grid = (ceil(W*3/256), oh)
__global__ __launch_bounds__(256) void pil_vertical_kernel(const uint8_t* src, int H, int W, uint8_t* dst, int oh,
                                                           const int* bounds, const int* kk, int ksize) {
    const int yy = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= W * 3) return;
    const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
    const int* k = kk + (long long)yy * ksize;
    int ss = 1 << (PIL_BITS - 1);
    for (int y = 0; y < ymax; y++) ss += (int)src[(long long)(y + ymin) * W * 3 + e] * k[y];
    dst[(long long)yy * W * 3 + e] = pil_clip8(ss);
}
void launch_pil_horizontal(const uint8_t* src, int H, int W, uint8_t* dst, int ow, const int* bounds, const int* kk,
                           int ksize, hipStream_t st) {
    dim3 grid((ow * 3 + 255) / 256, H);
    hipLaunchKernelGGL(pil_horizontal_kernel, grid, dim3(256), 0, st, src, H, W, dst, ow, bounds, kk, ksize);
}
void launch_pil_vertical(const uint8_t* src, int H, int W, uint8_t* dst, int oh, const int* bounds, const int* kk,
                         int ksize, hipStream_t st) {
    dim3 grid((W * 3 + 255) / 256, oh);
    hipLaunchKernelGGL(pil_vertical_kernel, grid, dim3(256), 0, st, src, H, W, dst, oh, bounds, kk, ksize);
}

// ---------------------------------------------------------------------------------------------
// max_pool2d(3, 2, 1) NHWC, C % 4 == 0; grid = (ceil(maxHo*maxWo*C/4/256), V)
// ---------------------------------------------------------------------------------------------
__device__ inline float nanmax(float m, float v) { return (v > m || v != v) ? v : m; }
// out16 != 0: the pooled tensor is stored in the split form conv_h3.hip / conv_h4.hip consume (h16.h, ConvArgs::in16) instead of fp32
__global__ __launch_bounds__(256) void maxpool_kernel(const float* in, float* out, const LevelSeg* sin, const LevelSeg* sout, int C, int out16) {
    const int v = blockIdx.y;
    const LevelSeg si = sin[v], so = sout[v];
    const int C4 = C >> 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)so.H * so.W * C4) return;
    const int c4 = (int)(e % C4);
    const int pix = (int)(e / C4);
    const int oy = pix / so.W, ox = pix - oy * so.W;
    const float4* ip = reinterpret_cast<const float4*>(in + si.pix_off * C);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int kh = 0; kh < 3; kh++)
#pragma unroll
        for (int kw = 0; kw < 3; kw++) {
            const int iy = oy * 2 - 1 + kh, ix = ox * 2 - 1 + kw;
            if (iy < 0 || iy >= si.H || ix < 0 || ix >= si.W) continue;
            const float4 t = ip[(long long)(iy * si.W + ix) * C4 + c4];
            m.x = nanmax(m.x, t.x); m.y = nanmax(m.y, t.y); m.z = nanmax(m.z, t.z); m.w = nanmax(m.w, t.w);
        }
    if (out16) h16_store4(reinterpret_cast<unsigned char*>(out + (so.pix_off + pix) * C), 4 * c4, m);
    else reinterpret_cast<float4*>(out + so.pix_off * C)[e] = m;
}
void launch_maxpool(const float* in, float* out, const LevelSeg* sin, const LevelSeg* sout, int C, int V, int max_out_pix,
                    hipStream_t st, bool out16) {
    dim3 grid((unsigned)(((long long)max_out_pix * (C / 4) + 255) / 256), V);
    hipLaunchKernelGGL(maxpool_kernel, grid, dim3(256), 0, st, in, out, sin, sout, C, out16 ? 1 : 0);
}

// LastLevelMaxPool = max_pool2d(x, 1, 2, 0): every second pixel of P5.
__global__ __launch_bounds__(256) void subsample2_kernel(const float* in, float* out, const LevelSeg* sin, const LevelSeg* sout, int C) {
    const int v = blockIdx.y;
    const LevelSeg si = sin[v], so = sout[v];
    const int C4 = C >> 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)so.H * so.W * C4) return;
    const int c4 = (int)(e % C4);
    const int pix = (int)(e / C4);
    const int oy = pix / so.W, ox = pix - oy * so.W;
    const float4* ip = reinterpret_cast<const float4*>(in + si.pix_off * C);
    reinterpret_cast<float4*>(out + so.pix_off * C)[e] = ip[(long long)((oy * 2) * si.W + ox * 2) * C4 + c4];
}
void launch_subsample2(const float* in, float* out, const LevelSeg* sin, const LevelSeg* sout, int C, int V, int max_out_pix,
                       hipStream_t st) {
    dim3 grid((unsigned)(((long long)max_out_pix * (C / 4) + 255) / 256), V);
    hipLaunchKernelGGL(subsample2_kernel, grid, dim3(256), 0, st, in, out, sin, sout, C);
}

// ---------------------------------------------------------------------------------------------
// PIL Image.rotate(expand=True, NEAREST): Pillow's 16.16 fixed-point affine loop (Geometry.c
// affine_fixed), one thread per output pixel.  a[6] = FIX()ed coefficients computed on the host.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_nearest_kernel(const uint8_t* src, int H, int W, uint8_t* dst, int oh, int ow,
                                                             int a0, int a1, int a2, int a3, int a4, int a5) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= ow) return;
    const int xx = a2 + a1 * y + a0 * x, yy = a5 + a4 * y + a3 * x;
    const int xin = xx >> 16, yin = yy >> 16;
    uint8_t r = 0, g = 0, b = 0;
    if (xin >= 0 && xin < W && yin >= 0 && yin < H) {
        const uint8_t* p = src + ((long long)yin * W + xin) * 3;
        r = p[0]; g = p[1]; b = p[2];
    }
    uint8_t* o = dst + ((long long)y * ow + x) * 3;
    o[0] = r; o[1] = g; o[2] = b;
}
void launch_affine_nearest(const uint8_t* src, int H, int W, uint8_t* dst, int oh, int ow, const int* a, hipStream_t st) {
    dim3 grid((ow + 255) / 256, oh);
    hipLaunchKernelGGL(affine_nearest_kernel, grid, dim3(256), 0, st, src, H, W, dst, oh, ow, a[0], a[1], a[2], a[3], a[4], a[5]);
}

// ---------------------------------------------------------------------------------------------
// cald_helper.GaussianNoise / SaltPepperNoise views of ONE image, drawn from ONE torch CPU generator in call order
// (get_uncertainty, cald_train.py:127-157; ls_c_train.py:129-131).  One workgroup per image runs MT19937 (seeded like
// torch.manual_seed: init_genrand(seed)) in LDS -- the 624-word twist is done in three dependency-free phases
// (k < 227, < 454, < 624) -- and keeps the tempered 24-bit uniforms of the last two twists in an LDS ring.
//   kind 0  torch.randn(3, H, W) * std / 255.0 (additive term, CHW float): uniforms -> Box-Muller per 16-chunk
//           (elements j, j + 8 pair up), and the "last 16 from 16 NEW uniforms" tail when the size is not a multiple
//           of 16; consumes n (+16) draws.  A chunk is transformed as soon as the twist holding its last draw exists.
//   kind 1  torch.rand(3, H, W) (CHW): u < prob/2 -> max(image), u > 1 - prob/2 -> min(image) on the uint8 image;
//           consumes n draws.
// ---------------------------------------------------------------------------------------------
__device__ inline unsigned mt_temper(unsigned y) {
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
}
__device__ inline float mt_u24(unsigned y) { return (float)((double)(y & 0xffffffu) * (1.0 / 16777216.0)); }
__device__ inline void box_muller_pair(float ua, float ub, float std, float* oa, float* ob) {
    const float u1 = 1.0f - ua;
    const float radius = sqrtf(-2.0f * det_logf(u1));
    float sn, cs;
    det_sincosf(6.283185307179586f * ub, &sn, &cs);
    *oa = ((radius * cs) * std) / 255.0f;
    *ob = ((radius * sn) * std) / 255.0f;
}
__global__ __launch_bounds__(256) void noise_stream_kernel(const NoiseJob* jobs) {
    __shared__ unsigned mt[624];
    __shared__ float ring[2][624];
    __shared__ long long seg_start[CALD_MAX_NOISE_SEG + 1];
    __shared__ int s_mx, s_mn;
    const NoiseJob& j = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    const long long n = (long long)j.H * j.W * 3, plane = (long long)j.H * j.W;
    const long long rem = n % 16, ncs = n / 16;                       // standard chunks per randn segment
    const int nseg = j.nseg;
    if (tid == 0) {
        unsigned x = (unsigned)(j.seed & 0xffffffffull);
        mt[0] = x;
        for (int i = 1; i < 624; i++) { x = 1812433253u * (x ^ (x >> 30)) + (unsigned)i; mt[i] = x; }
        long long d = 0;
        for (int g = 0; g < nseg; g++) { seg_start[g] = d; d += n + ((j.seg[g].kind == 0 && rem && n >= 16) ? 16 : 0); }
        seg_start[nseg] = d;
        s_mx = 0; s_mn = 255;
    }
    __syncthreads();
    bool any_sp = false;
    for (int g = 0; g < nseg; g++) any_sp |= j.seg[g].kind == 1;
    if (any_sp) {
        int mx = 0, mn = 255;
        for (long long i = tid; i < n; i += 256) { const int v = j.src[i]; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
        atomicMax(&s_mx, mx); atomicMin(&s_mn, mn);
        __syncthreads();
    }
    const unsigned char vmax = (unsigned char)s_mx, vmin = (unsigned char)s_mn;
    const long long total = seg_start[nseg];
    const long long nblk = (total + 623) / 624;
    for (long long blk = 0; blk < nblk; blk++) {
        // twist: new[k] = old[k+397 mod 624] ^ f(old[k], old[k+1]); phases keep every read well-defined
        for (int ph = 0; ph < 3; ph++) {
            const int lo = ph == 0 ? 0 : (ph == 1 ? 227 : 454), hi = ph == 0 ? 227 : (ph == 1 ? 454 : 624);
            unsigned nv = 0; const int k = lo + tid;
            if (k < hi) {
                const unsigned y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                nv = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            __syncthreads();
            if (k < hi) mt[k] = nv;
            __syncthreads();
        }
        for (int t = tid; t < 624; t += 256) ring[blk & 1][t] = mt_u24(mt_temper(mt[t]));
        __syncthreads();
        const long long b0 = blk * 624;
        const long long d_lo = b0 - 15, d_hi = b0 + 608;                 // chunk starts whose last draw is in this twist
        for (int g = 0; g < nseg; g++) {
            const long long base = seg_start[g];
            if (j.seg[g].kind == 1) {
                // draws [base, base + n) intersected with this twist
                const long long lo = base > b0 ? base : b0, hi = (base + n < b0 + 624) ? base + n : b0 + 624;
                unsigned char* dst = reinterpret_cast<unsigned char*>(j.seg[g].dst);
                const float plo = j.seg[g].p0, phi = j.seg[g].p1;
                for (long long d = lo + tid; d < hi; d += 256) {
                    const float u = ring[blk & 1][d - b0];
                    const long long e = d - base;
                    const int c = (int)(e / plane);
                    const long long o = (e - (long long)c * plane) * 3 + c;
                    unsigned char v = j.src[o];
                    if (u < plo) v = vmax;
                    if (u > phi) v = vmin;
                    dst[o] = v;
                }
                continue;
            }
            const long long seg_draws = seg_start[g + 1] - base;
            if (base > d_hi || base + seg_draws - 16 < d_lo) continue;
            // chunk ids: 0 .. ncs-1 standard (start base + 16c), id ncs = tail (start base + n) when rem
            long long c_lo = d_lo - base; c_lo = c_lo <= 0 ? 0 : (c_lo + 15) / 16;
            long long c_hi = d_hi - base; c_hi = c_hi < 0 ? -1 : c_hi / 16;
            if (c_hi > ncs - 1) c_hi = ncs - 1;
            float* dst = reinterpret_cast<float*>(j.seg[g].dst);
            const float std = j.seg[g].p0;
            for (long long w = c_lo * 8 + tid; w < (c_hi + 1) * 8; w += 256) {
                const long long c = w >> 3; const int jj = (int)(w & 7);
                const long long s = base + 16 * c + jj, s8 = s + 8;
                float oa, ob;
                box_muller_pair(ring[(s / 624) & 1][s % 624], ring[(s8 / 624) & 1][s8 % 624], std, &oa, &ob);
                const long long e0 = 16 * c + jj;
                if (!rem || e0 < n - 16) dst[e0] = oa;
                if (!rem || e0 + 8 < n - 16) dst[e0 + 8] = ob;
            }
            if (rem && n >= 16) {
                const long long ts = base + n;                          // tail chunk start
                if (ts >= d_lo && ts <= d_hi && tid < 8) {
                    const long long s = ts + tid, s8 = s + 8;
                    float oa, ob;
                    box_muller_pair(ring[(s / 624) & 1][s % 624], ring[(s8 / 624) & 1][s8 % 624], std, &oa, &ob);
                    dst[n - 16 + tid] = oa; dst[n - 16 + tid + 8] = ob;
                }
            }
        }
        __syncthreads();
    }
}
void launch_noise_stream(const NoiseJob* jobs, int n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(noise_stream_kernel, dim3(n), dim3(256), 0, st, jobs);
}

// ---------------------------------------------------------------------------------------------
// cald_helper.ColorAdjust (cald_helper.py:65-69) = PIL.ImageEnhance Brightness -> Contrast -> Color, each
// Image.blend(degenerate, image, factor): temp = (float)(in1 + alpha * (in2 - in1)) in C float arithmetic (one float
// multiply, one float add), clipped to [0, 255] and truncated when alpha is outside [0, 1].
//   pass 1 (grid over pixels): brightness (degenerate = black) -> tmp; sum of L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16
//   pass 2: contrast against the rounded mean of L, then saturation against the pixel's own L -> dst
// ---------------------------------------------------------------------------------------------
__device__ inline unsigned char pil_blend1(int in1, int in2, float alpha) {
    const float temp = (float)in1 + alpha * (float)(in2 - in1);
    if (alpha >= 0.0f && alpha <= 1.0f) return (unsigned char)temp;
    if (temp <= 0.0f) return 0;
    if (temp >= 255.0f) return 255;
    return (unsigned char)temp;
}
__device__ inline int pil_l(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }
__global__ __launch_bounds__(256) void color_brightness_kernel(const uint8_t* src, long long npx, float f, uint8_t* tmp,
                                                               unsigned long long* lsum) {
    __shared__ unsigned long long red[256];
    unsigned long long acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npx; i += (long long)gridDim.x * 256) {
        int c[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int v = src[3 * i + k];
            c[k] = f == 1.0f ? v : (f == 0.0f ? 0 : pil_blend1(0, v, f));
            tmp[3 * i + k] = (uint8_t)c[k];
        }
        acc += (unsigned long long)pil_l(c[0], c[1], c[2]);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) atomicAdd(lsum, red[0]);
}
__global__ __launch_bounds__(256) void color_contrast_saturation_kernel(const uint8_t* tmp, long long npx, float f,
                                                                        const unsigned long long* lsum, uint8_t* dst) {
    const int mean = (int)((double)(*lsum) / (double)npx + 0.5);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npx; i += (long long)gridDim.x * 256) {
        int c[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int v = tmp[3 * i + k];
            c[k] = f == 1.0f ? v : (f == 0.0f ? mean : pil_blend1(mean, v, f));
        }
        const int l = pil_l(c[0], c[1], c[2]);
#pragma unroll
        for (int k = 0; k < 3; k++) dst[3 * i + k] = (uint8_t)(f == 1.0f ? c[k] : (f == 0.0f ? l : pil_blend1(l, c[k], f)));
    }
}
void launch_color_adjust(const uint8_t* src, int H, int W, float factor, uint8_t* tmp, unsigned long long* lsum, uint8_t* dst,
                         hipStream_t st) {
    const long long npx = (long long)H * W;
    const int grid = (int)((npx + 255) / 256 < 1024 ? (npx + 255) / 256 : 1024);
    hipMemsetAsync(lsum, 0, sizeof(unsigned long long), st);
    hipLaunchKernelGGL(color_brightness_kernel, dim3(grid), dim3(256), 0, st, src, npx, factor, tmp, lsum);
    hipLaunchKernelGGL(color_contrast_saturation_kernel, dim3(grid), dim3(256), 0, st, tmp, npx, factor, lsum, dst);
}
