// jpeg.hip -- input side of the sweep: batched baseline-JPEG decode on the GPU,
// producing the uint8 HWC RGB image that PIL.Image.open(path).convert('RGB') hands to the reference
// (torchvision VOCDetection.__getitem__ via detection/voc_utils.py:47-58; cald_train.py:434 DataLoader).
//
// The decoder follows libjpeg(-turbo)'s defaults, which is what Pillow runs: ITU T.81 Huffman decoding,
// the 13-bit fixed-point LL&M inverse DCT (JDCT_ISLOW), triangle-filter ("fancy") chroma upsampling and the
// 16-bit fixed-point YCbCr->RGB conversion.  All integer arithmetic: results are bit-identical to Pillow's.
//
// Mapping onto the GPU
//   host     : marker parsing only (a few hundred bytes per file); the entropy-coded bytes are copied verbatim
//              (byte stuffing is removed on the device).
//   kernel 1 : Huffman decode, ONE WAVEFRONT PER IMAGE (the bit stream of a baseline JPEG without restart markers
//              is strictly serial; a pool has thousands of images, so images are the parallel axis).  The decode
//              loop is wave-uniform and runs mostly on the scalar unit; 9-bit lookahead tables in LDS, 64-bit
//              bit buffer fed by prefetched 8-byte scalar loads.
//   kernel 2 : dequantise + 8x8 inverse DCT, 8 lanes per block (column pass, LDS transpose, row pass).
//   kernel 3 : chroma upsampling + colour conversion, one lane per output pixel.
#include "../../include/cald_hip.h"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <vector>

int cald_internal_fail(int code, const char* fmt, ...);
hipStream_t cald_internal_stream(cald_ctx* c);
#define JHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return cald_internal_fail(CALD_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

namespace {

struct JTab {                 // one Huffman table in device layout
    uint16_t lut[512];        // 9-bit lookahead: (length << 8) | symbol; 0 = code longer than 9 bits
    int maxcode[17];          // largest code of length l (1..16), -1 if there is none
    int valoff[17];           // valptr[l] - mincode[l]
    uint8_t vals[256];
};

struct JImg {                 // one image, device-visible
    int W, H, nc, hmax, vmax, mcux, mcuy, restart;
    int ch[3], cv[3];         // sampling factors
    int bw[3], bh[3];         // blocks per row / column (MCU-padded)
    int dw[3], dh[3];         // downsampled width / height (real samples)
    long long scan_off;       // byte offset (8-aligned) of the entropy-coded segment in the packed stream buffer
    int scan_len;
    long long coef_off[3];    // int16 element offsets
    long long plane_off[3];   // byte offsets
    unsigned short q[3][64];  // dequantisation tables per component, natural order
    int tab[3][2];            // JTab indices: [component][0 = DC, 1 = AC]
    unsigned char* out;       // [H][W][3]
};

struct HostHuff { bool set = false; uint8_t bits[17]; uint8_t vals[256]; };

struct ZigZag {
    uint8_t zz[64];
    ZigZag() {
        int k = 0;
        for (int s = 0; s < 15; s++) {
            if (s & 1) { for (int r = 0; r < 8; r++) { int c = s - r; if (c >= 0 && c < 8) zz[k++] = (uint8_t)(r * 8 + c); } }
            else       { for (int c = 0; c < 8; c++) { int r = s - c; if (r >= 0 && r < 8) zz[k++] = (uint8_t)(r * 8 + c); } }
        }
    }
};
const uint8_t* zigzag_table() {
    static const ZigZag t;     // function-local static: initialised once, thread-safely
    return t.zz;
}

void build_jtab(const HostHuff& h, JTab* t) {
    memset(t, 0, sizeof(*t));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        const int mincode = code;
        t->valoff[l] = k - mincode;
        for (int i = 0; i < h.bits[l]; i++, k++, code++) {
            if (l <= 9) {
                const int lo = code << (9 - l), n = 1 << (9 - l);
                for (int j = 0; j < n && lo + j < 512; j++) t->lut[lo + j] = (uint16_t)((l << 8) | h.vals[k]);
            }
        }
        t->maxcode[l] = h.bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    memcpy(t->vals, h.vals, 256);
}

inline int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

struct Parsed {
    JImg im;
    HostHuff dc[4], ac[4];
    int td[3], ta[3];
    const uint8_t* scan = nullptr;
    size_t scan_len = 0;
};

// returns 0, CALD_ERR_INVALID (broken file) or CALD_ERR_UNSUPPORTED
int parse_jpeg(const uint8_t* d, size_t n, Parsed* P) {
    const uint8_t* zz = zigzag_table();
    JImg& j = P->im;
    memset(&j, 0, sizeof(j));
    unsigned short q[4][64]; bool qset[4] = {false, false, false, false};
    int cid[3] = {0, 0, 0}, ctq[3] = {0, 0, 0};
    bool saw_jfif = false, saw_adobe = false, have_sof = false; int adobe_transform = 0;
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return CALD_ERR_INVALID;
    size_t p = 2;
    while (p + 4 <= n) {
        if (d[p] != 0xFF) return CALD_ERR_INVALID;
        while (p < n && d[p] == 0xFF) p++;
        if (p >= n) return CALD_ERR_INVALID;
        const int m = d[p++];
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) return CALD_ERR_INVALID;
        if (p + 2 > n) return CALD_ERR_INVALID;
        const int len = rd16(d + p);
        if (len < 2 || p + len > n) return CALD_ERR_INVALID;
        const uint8_t* s = d + p + 2;
        const int sl = len - 2;
        if (m == 0xDB) {
            int o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15; o++;
                if (tq > 3 || o + (pq ? 128 : 64) > sl) return CALD_ERR_INVALID;
                for (int i = 0; i < 64; i++) q[tq][zz[i]] = (unsigned short)(pq ? rd16(s + o + 2 * i) : s[o + i]);
                o += pq ? 128 : 64;
                qset[tq] = true;
            }
        } else if (m == 0xC4) {
            int o = 0;
            while (o < sl) {
                if (o + 17 > sl) return CALD_ERR_INVALID;
                const int tc = s[o] >> 4, th = s[o] & 15; o++;
                if (tc > 1 || th > 3) return CALD_ERR_INVALID;
                HostHuff& t = tc ? P->ac[th] : P->dc[th];
                int cnt = 0;
                t.bits[0] = 0;
                for (int i = 1; i <= 16; i++) { t.bits[i] = s[o + i - 1]; cnt += t.bits[i]; }
                o += 16;
                if (cnt > 256 || o + cnt > sl) return CALD_ERR_INVALID;
                memset(t.vals, 0, 256);
                memcpy(t.vals, s + o, cnt);
                o += cnt;
                t.set = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6) return CALD_ERR_INVALID;
            if (s[0] != 8) return CALD_ERR_UNSUPPORTED;
            j.H = rd16(s + 1); j.W = rd16(s + 3); j.nc = s[5];
            if (j.H <= 0 || j.W <= 0) return CALD_ERR_INVALID;
            if (j.nc != 1 && j.nc != 3) return CALD_ERR_UNSUPPORTED;
            if (sl < 6 + 3 * j.nc) return CALD_ERR_INVALID;
            for (int i = 0; i < j.nc; i++) {
                cid[i] = s[6 + 3 * i];
                j.ch[i] = s[7 + 3 * i] >> 4; j.cv[i] = s[7 + 3 * i] & 15;
                ctq[i] = s[8 + 3 * i];
                if (ctq[i] > 3) return CALD_ERR_INVALID;
            }
            have_sof = true;
        } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            return CALD_ERR_UNSUPPORTED;         // progressive / lossless / arithmetic / hierarchical
        } else if (m == 0xDD) {
            if (sl < 2) return CALD_ERR_INVALID;
            j.restart = rd16(s);
        } else if (m == 0xE0) {
            if (sl >= 5 && !memcmp(s, "JFIF\0", 5)) saw_jfif = true;
        } else if (m == 0xEE) {
            if (sl >= 12 && !memcmp(s, "Adobe", 5)) { saw_adobe = true; adobe_transform = s[11]; }
        } else if (m == 0xDA) {
            if (!have_sof) return CALD_ERR_INVALID;
            if (sl < 1 || s[0] != j.nc) return CALD_ERR_UNSUPPORTED;      // non-interleaved multi-scan file
            if (sl < 1 + 2 * j.nc + 3) return CALD_ERR_INVALID;
            for (int i = 0; i < j.nc; i++) {
                if (s[1 + 2 * i] != cid[i]) return CALD_ERR_UNSUPPORTED;
                P->td[i] = s[2 + 2 * i] >> 4; P->ta[i] = s[2 + 2 * i] & 15;
                if (P->td[i] > 3 || P->ta[i] > 3) return CALD_ERR_INVALID;
                if (!P->dc[P->td[i]].set || !P->ac[P->ta[i]].set || !qset[ctq[i]]) return CALD_ERR_INVALID;
                memcpy(j.q[i], q[ctq[i]], sizeof(j.q[i]));
            }
            P->scan = d + p + len;
            P->scan_len = n - (p + len);
            break;
        }
        p += len;
    }
    if (!P->scan) return CALD_ERR_INVALID;
    if (j.nc == 3) {                             // libjpeg default_decompress_parms colour-space rule
        bool ycc = true;
        if (saw_jfif) ycc = true;
        else if (saw_adobe) ycc = adobe_transform != 0;
        else if (cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B') ycc = false;
        if (!ycc) return CALD_ERR_UNSUPPORTED;
    }
    j.hmax = j.ch[0]; j.vmax = j.cv[0];
    if (j.nc == 1) { j.ch[0] = j.cv[0] = 1; j.hmax = j.vmax = 1; }
    else {
        if (j.ch[1] != 1 || j.cv[1] != 1 || j.ch[2] != 1 || j.cv[2] != 1) return CALD_ERR_UNSUPPORTED;
        if (!((j.hmax == 1 && j.vmax == 1) || (j.hmax == 2 && j.vmax == 1) || (j.hmax == 2 && j.vmax == 2))) return CALD_ERR_UNSUPPORTED;
    }
    j.mcux = (j.W + 8 * j.hmax - 1) / (8 * j.hmax);
    j.mcuy = (j.H + 8 * j.vmax - 1) / (8 * j.vmax);
    for (int i = 0; i < j.nc; i++) {
        j.bw[i] = j.mcux * j.ch[i]; j.bh[i] = j.mcuy * j.cv[i];
        j.dw[i] = (j.W * j.ch[i] + j.hmax - 1) / j.hmax;
        j.dh[i] = (j.H * j.cv[i] + j.vmax - 1) / j.vmax;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// kernel 1: entropy decode, ONE WAVEFRONT PER IMAGE.  Everything in the decode loop is wave-uniform, so the
// compiler keeps the bit buffer and the control flow on the scalar unit (s_load for the byte stream, SALU shifts);
// the image's Huffman tables are staged into LDS once; lane 0 stores the coefficients.
// ---------------------------------------------------------------------------------------------
struct BitSrc {
    const unsigned long long* q;   // next 8-byte word to fetch
    unsigned long long w, wnext;   // current / prefetched word
    int wn;                        // bytes left in w (next byte = low byte)
    unsigned long long acc; int n; // left-aligned bit buffer
    int pending;                   // marker seen in the stream (0 = none): feed zeros, do not read on
};
__device__ inline int src_byte(BitSrc& s) {
    if (s.wn == 0) { s.w = s.wnext; s.wnext = *s.q++; s.wn = 8; }
    const int b = (int)(s.w & 0xFFull);
    s.w >>= 8; s.wn--;
    return b;
}
__device__ inline void src_fill(BitSrc& s) {
    while (s.n <= 56) {
        int byte = 0;
        if (!s.pending) {
            byte = src_byte(s);
            if (byte == 0xFF) {
                int nb = src_byte(s);
                while (nb == 0xFF) nb = src_byte(s);
                if (nb != 0) { s.pending = nb; byte = 0; }
            }
        }
        s.acc |= (unsigned long long)byte << (56 - s.n);
        s.n += 8;
    }
}
__device__ inline int src_bits(BitSrc& s, int k) {   // k in 1..16, caller guarantees n >= k
    const int v = (int)(s.acc >> (64 - k));
    s.acc <<= k; s.n -= k;
    return v;
}
__device__ inline int huff_symbol(BitSrc& s, const JTab* t) {
    const int peek = (int)(s.acc >> 48);          // 16 bits
    const int e = t->lut[peek >> 7];
    if (e) { const int l = e >> 8; s.acc <<= l; s.n -= l; return e & 255; }
    for (int l = 10; l <= 16; l++) {
        const int code = peek >> (16 - l);
        if (code <= t->maxcode[l]) { s.acc <<= l; s.n -= l; return t->vals[code + t->valoff[l]]; }
    }
    s.acc <<= 16; s.n -= 16;
    return 0;
}
__device__ inline int huff_extend(int r, int k) { return r < (1 << (k - 1)) ? r - (1 << k) + 1 : r; }

__device__ inline void decode_block(BitSrc& s, const JTab* tdc, const JTab* tac, const unsigned char* zz, int& pred,
                                    short* blk, bool writer) {
    src_fill(s);
    int k = huff_symbol(s, tdc) & 15;
    if (k) { const int r = src_bits(s, k); pred += huff_extend(r, k); }
    if (writer && pred) blk[0] = (short)pred;
    for (int z = 1; z < 64; z++) {
        src_fill(s);
        const int rs = huff_symbol(s, tac);
        const int r = rs >> 4, sz = rs & 15;
        if (sz) {
            z += r;
            const int val = huff_extend(src_bits(s, sz), sz);
            if (writer && z < 64) blk[zz[z]] = (short)val;
        } else {
            if (r == 15) z += 15; else break;
        }
    }
}

// grid = n_img, block = 64
__global__ __launch_bounds__(64) void jpeg_huffman_kernel(const JImg* imgs, const JTab* tabs, const unsigned char* stream,
                                                          short* coef) {
    __shared__ unsigned char zz[64];
    __shared__ JTab lt[6];
    const int lane = threadIdx.x;
    const JImg& im = imgs[blockIdx.x];
    const int nc = im.nc;
    {
        if (lane == 0) {      // zigzag -> natural order (T.81 figure A.6)
            int k = 0;
            for (int s = 0; s < 15; s++) {
                if (s & 1) { for (int r = 0; r < 8; r++) { const int c = s - r; if (c >= 0 && c < 8) zz[k++] = (unsigned char)(r * 8 + c); } }
                else       { for (int c = 0; c < 8; c++) { const int r = s - c; if (r >= 0 && r < 8) zz[k++] = (unsigned char)(r * 8 + c); } }
            }
        }
        for (int c = 0; c < nc; c++)
            for (int d = 0; d < 2; d++) {
                const unsigned* src = reinterpret_cast<const unsigned*>(tabs + im.tab[c][d]);
                unsigned* dst = reinterpret_cast<unsigned*>(&lt[c * 2 + d]);
                for (int e = lane; e < (int)(sizeof(JTab) / 4); e += 64) dst[e] = src[e];
            }
        __syncthreads();
    }
    BitSrc s;
    s.q = reinterpret_cast<const unsigned long long*>(stream + im.scan_off);
    s.wnext = *s.q++;
    s.w = 0; s.wn = 0; s.acc = 0; s.n = 0; s.pending = 0;
    int pred0 = 0, pred1 = 0, pred2 = 0;
    const int restart = im.restart, mcux = im.mcux, mcuy = im.mcuy;
    const int h0 = im.ch[0], v0 = im.cv[0], bw0 = im.bw[0], bw1 = im.bw[1], bw2 = im.bw[2];
    short* c0 = coef + im.coef_off[0];
    short* c1 = coef + im.coef_off[1];
    short* c2 = coef + im.coef_off[2];
    const bool writer = lane == 0;
    int left = restart;
    for (int my = 0; my < mcuy; my++)
        for (int mx = 0; mx < mcux; mx++) {
            if (restart && left == 0) {
                if (s.pending >= 0xD0 && s.pending <= 0xD7) s.pending = 0;
                else if (s.pending == 0) {
                    for (;;) {
                        int b = src_byte(s);
                        if (b != 0xFF) continue;
                        int nb = src_byte(s);
                        while (nb == 0xFF) nb = src_byte(s);
                        if (nb >= 0xD0 && nb <= 0xD7) break;
                        if (nb != 0) { s.pending = nb; break; }
                    }
                }
                s.acc = 0; s.n = 0;
                pred0 = pred1 = pred2 = 0;
                left = restart;
            }
            for (int v = 0; v < v0; v++)
                for (int h = 0; h < h0; h++)
                    decode_block(s, &lt[0], &lt[1], zz, pred0, c0 + ((long long)(my * v0 + v) * bw0 + (mx * h0 + h)) * 64, writer);
            if (nc == 3) {
                decode_block(s, &lt[2], &lt[3], zz, pred1, c1 + ((long long)my * bw1 + mx) * 64, writer);
                decode_block(s, &lt[4], &lt[5], zz, pred2, c2 + ((long long)my * bw2 + mx) * 64, writer);
            }
            if (restart) left--;
        }
}

// ---------------------------------------------------------------------------------------------
// kernel 2: dequantise + inverse DCT (jidctint.c arithmetic: CONST_BITS 13, PASS1_BITS 2)
// grid = (ceil(max_blocks / 32), n_img * 3), block = 256: 8 lanes per 8x8 block
// ---------------------------------------------------------------------------------------------
__device__ inline int jdescale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ inline void idct8(const int in[8], int out[8], int shift) {
    int z1, z2, z3, z4, z5, t0, t1, t2, t3, t10, t11, t12, t13;
    z2 = in[2]; z3 = in[6];
    z1 = (z2 + z3) * 4433;
    t2 = z1 + z3 * (-15137);
    t3 = z1 + z2 * 6270;
    z2 = in[0]; z3 = in[4];
    t0 = (z2 + z3) * 8192; t1 = (z2 - z3) * 8192;
    t10 = t0 + t3; t13 = t0 - t3; t11 = t1 + t2; t12 = t1 - t2;
    t0 = in[7]; t1 = in[5]; t2 = in[3]; t3 = in[1];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3;
    z5 = (z3 + z4) * 9633;
    t0 = t0 * 2446; t1 = t1 * 16819; t2 = t2 * 25172; t3 = t3 * 12299;
    z1 = z1 * (-7373); z2 = z2 * (-20995); z3 = z3 * (-16069); z4 = z4 * (-3196);
    z3 += z5; z4 += z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    out[0] = jdescale(t10 + t3, shift); out[7] = jdescale(t10 - t3, shift);
    out[1] = jdescale(t11 + t2, shift); out[6] = jdescale(t11 - t2, shift);
    out[2] = jdescale(t12 + t1, shift); out[5] = jdescale(t12 - t1, shift);
    out[3] = jdescale(t13 + t0, shift); out[4] = jdescale(t13 - t0, shift);
}
__device__ inline unsigned range_limit_idct(int x) {   // sample_range_limit + CENTERJSAMPLE, index masked to 10 bits
    x &= 1023;
    if (x < 128) return (unsigned)(x + 128);
    if (x < 512) return 255u;
    if (x < 896) return 0u;
    return (unsigned)(x - 896);
}
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const JImg* imgs, const short* coef, unsigned char* planes) {
    __shared__ int ws[32][8][9];
    const int img = blockIdx.y / 3, c = blockIdx.y - img * 3;
    const JImg& im = imgs[img];
    if (c >= im.nc) return;
    const int g = threadIdx.x >> 3, t = threadIdx.x & 7;
    const int b = blockIdx.x * 32 + g;
    const int nb = im.bw[c] * im.bh[c];
    const bool live = b < nb;
    int in[8], o[8];
    if (live) {
        const short* blk = coef + im.coef_off[c] + (long long)b * 64;
#pragma unroll
        for (int r = 0; r < 8; r++) in[r] = (int)blk[r * 8 + t] * (int)im.q[c][r * 8 + t];
        idct8(in, o, 11);
#pragma unroll
        for (int r = 0; r < 8; r++) ws[g][r][t] = o[r];
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int k = 0; k < 8; k++) in[k] = ws[g][t][k];
        idct8(in, o, 18);
        const int by = b / im.bw[c], bx = b - by * im.bw[c];
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { lo |= range_limit_idct(o[k]) << (8 * k); hi |= range_limit_idct(o[k + 4]) << (8 * k); }
        unsigned char* row = planes + im.plane_off[c] + ((long long)(by * 8 + t) * im.bw[c] + bx) * 8;
        *reinterpret_cast<uint2*>(row) = make_uint2(lo, hi);
    }
}

// ---------------------------------------------------------------------------------------------
// kernel 3: jdsample.c triangle upsampling + jdcolor.c YCbCr -> RGB; grid = (ceil(maxW*maxH/256), n_img)
// ---------------------------------------------------------------------------------------------
__device__ inline int chroma_at(const JImg& im, int c, const unsigned char* P, int x, int y) {
    const int stride = im.bw[c] * 8, dw = im.dw[c], dh = im.dh[c];
    if (im.hmax == 1) return P[(long long)y * stride + x];
    const bool fancy = dw > 2;
    const int cx = x >> 1;
    if (im.vmax == 1) {
        const unsigned char* row = P + (long long)y * stride;
        if (!fancy) return row[cx];
        if (x & 1) return cx == dw - 1 ? row[cx] : (row[cx] * 3 + row[cx + 1] + 2) >> 2;
        return cx == 0 ? row[cx] : (row[cx] * 3 + row[cx - 1] + 1) >> 2;
    }
    const int cy = y >> 1;
    if (!fancy) return P[(long long)cy * stride + cx];
    int ny = (y & 1) ? cy + 1 : cy - 1;
    ny = ny < 0 ? 0 : (ny > dh - 1 ? dh - 1 : ny);
    const unsigned char* r0 = P + (long long)cy * stride;
    const unsigned char* r1 = P + (long long)ny * stride;
    const int cur = r0[cx] * 3 + r1[cx];
    if (x & 1) return cx == dw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + (r0[cx + 1] * 3 + r1[cx + 1]) + 7) >> 4;
    return cx == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + (r0[cx - 1] * 3 + r1[cx - 1]) + 8) >> 4;
}
__device__ inline unsigned char clamp8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
__global__ __launch_bounds__(256) void jpeg_color_kernel(const JImg* imgs, const unsigned char* planes) {
    const JImg& im = imgs[blockIdx.y];
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= im.W * im.H) return;
    const int y = pix / im.W, x = pix - y * im.W;
    const int Y = planes[im.plane_off[0] + (long long)y * im.bw[0] * 8 + x];
    unsigned char* o = im.out + (long long)pix * 3;
    if (im.nc == 1) { o[0] = o[1] = o[2] = (unsigned char)Y; return; }
    const int cb = chroma_at(im, 1, planes + im.plane_off[1], x, y) - 128;
    const int cr = chroma_at(im, 2, planes + im.plane_off[2], x, y) - 128;
    o[0] = clamp8(Y + ((91881 * cr + 32768) >> 16));
    o[1] = clamp8(Y + ((-22554 * cb + 32768 + -46802 * cr) >> 16));
    o[2] = clamp8(Y + ((116130 * cb + 32768) >> 16));
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" int cald_jpeg_info(const uint8_t* data, size_t size, int* H, int* W, int* ncomp) {
    if (!data || !H || !W) return cald_internal_fail(CALD_ERR_INVALID, "cald_jpeg_info: null argument");
    Parsed P;
    const int rc = parse_jpeg(data, size, &P);
    if (rc == CALD_ERR_UNSUPPORTED) return cald_internal_fail(rc, "cald_jpeg_info: JPEG flavour outside the supported set (8-bit baseline Huffman, gray / YCbCr 4:4:4, 4:2:2, 4:2:0, one interleaved scan)");
    if (rc) return cald_internal_fail(rc, "cald_jpeg_info: not a decodable JPEG stream");
    *H = P.im.H; *W = P.im.W;
    if (ncomp) *ncomp = P.im.nc;
    return CALD_OK;
}

extern "C" int cald_jpeg_decode_batch(cald_ctx* ctx, int n, const uint8_t* const* data, const size_t* sizes,
                                      uint8_t* const* out_dev) {
    if (!ctx || n < 0 || (n && (!data || !sizes || !out_dev))) return cald_internal_fail(CALD_ERR_INVALID, "cald_jpeg_decode_batch: null argument");
    if (n == 0) return CALD_OK;
    hipStream_t st = cald_internal_stream(ctx);
    std::vector<JImg> imgs(n);
    std::vector<JTab> tabs;
    tabs.reserve((size_t)n * 4);
    size_t stream_bytes = 0, coef_elems = 0, plane_bytes = 0;
    std::vector<const uint8_t*> scans(n);
    int max_blocks = 0, max_pix = 0;
    for (int i = 0; i < n; i++) {
        Parsed P;
        const int rc = parse_jpeg(data[i], sizes[i], &P);
        if (rc == CALD_ERR_UNSUPPORTED) return cald_internal_fail(rc, "cald_jpeg_decode_batch: image %d: JPEG flavour outside the supported set (8-bit baseline Huffman, gray / YCbCr 4:4:4, 4:2:2, 4:2:0, one interleaved scan)", i);
        if (rc) return cald_internal_fail(rc, "cald_jpeg_decode_batch: image %d is not a decodable JPEG stream", i);
        if (!out_dev[i]) return cald_internal_fail(CALD_ERR_INVALID, "cald_jpeg_decode_batch: image %d: null output", i);
        JImg& im = P.im;
        // Huffman tables: one JTab per distinct (class, id) used by this image
        int dc_idx[4] = {-1, -1, -1, -1}, ac_idx[4] = {-1, -1, -1, -1};
        for (int c = 0; c < im.nc; c++) {
            if (dc_idx[P.td[c]] < 0) { dc_idx[P.td[c]] = (int)tabs.size(); tabs.emplace_back(); build_jtab(P.dc[P.td[c]], &tabs.back()); }
            if (ac_idx[P.ta[c]] < 0) { ac_idx[P.ta[c]] = (int)tabs.size(); tabs.emplace_back(); build_jtab(P.ac[P.ta[c]], &tabs.back()); }
            im.tab[c][0] = dc_idx[P.td[c]]; im.tab[c][1] = ac_idx[P.ta[c]];
        }
        im.scan_off = (long long)stream_bytes;
        im.scan_len = (int)P.scan_len;
        stream_bytes += (P.scan_len + 2 + 16 + 7) & ~(size_t)7;       // + FF D9 + slack, 8-byte aligned
        for (int c = 0; c < im.nc; c++) {
            const size_t nb = (size_t)im.bw[c] * im.bh[c];
            im.coef_off[c] = (long long)coef_elems; coef_elems += nb * 64;
            im.plane_off[c] = (long long)plane_bytes; plane_bytes += nb * 64;
            if ((int)nb > max_blocks) max_blocks = (int)nb;
        }
        if (im.W * im.H > max_pix) max_pix = im.W * im.H;
        im.out = out_dev[i];
        scans[i] = P.scan;
        imgs[i] = im;
    }
    // pack the entropy-coded segments (verbatim) + an EOI so that a truncated file ends in a marker
    unsigned char* h_stream = nullptr;
    JHIP(hipHostMalloc((void**)&h_stream, stream_bytes, hipHostMallocDefault));
    memset(h_stream, 0, stream_bytes);
    for (int i = 0; i < n; i++) {
        memcpy(h_stream + imgs[i].scan_off, scans[i], (size_t)imgs[i].scan_len);
        h_stream[imgs[i].scan_off + imgs[i].scan_len] = 0xFF;
        h_stream[imgs[i].scan_off + imgs[i].scan_len + 1] = 0xD9;
    }
    unsigned char *d_stream = nullptr, *d_planes = nullptr; short* d_coef = nullptr; JImg* d_imgs = nullptr; JTab* d_tabs = nullptr;
    int rc = CALD_OK;
    auto cleanup = [&]() {
        if (d_stream) hipFree(d_stream);
        if (d_planes) hipFree(d_planes);
        if (d_coef) hipFree(d_coef);
        if (d_imgs) hipFree(d_imgs);
        if (d_tabs) hipFree(d_tabs);
        hipHostFree(h_stream);
    };
#define JTRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = cald_internal_fail(CALD_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); cleanup(); return rc; } } while (0)
    JTRY(hipMalloc((void**)&d_stream, stream_bytes));
    JTRY(hipMalloc((void**)&d_planes, plane_bytes));
    JTRY(hipMalloc((void**)&d_coef, coef_elems * sizeof(short)));
    JTRY(hipMalloc((void**)&d_imgs, sizeof(JImg) * n));
    JTRY(hipMalloc((void**)&d_tabs, sizeof(JTab) * tabs.size()));
    JTRY(hipMemcpyAsync(d_stream, h_stream, stream_bytes, hipMemcpyHostToDevice, st));
    JTRY(hipMemcpyAsync(d_imgs, imgs.data(), sizeof(JImg) * n, hipMemcpyHostToDevice, st));
    JTRY(hipMemcpyAsync(d_tabs, tabs.data(), sizeof(JTab) * tabs.size(), hipMemcpyHostToDevice, st));
    JTRY(hipMemsetAsync(d_coef, 0, coef_elems * sizeof(short), st));
    hipLaunchKernelGGL(jpeg_huffman_kernel, dim3(n), dim3(64), 0, st, d_imgs, d_tabs, d_stream, d_coef);
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((max_blocks + 31) / 32, n * 3), dim3(256), 0, st, d_imgs, d_coef, d_planes);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((max_pix + 255) / 256, n), dim3(256), 0, st, d_imgs, d_planes);
    JTRY(hipGetLastError());
    JTRY(hipStreamSynchronize(st));
    cleanup();
    return CALD_OK;
}
