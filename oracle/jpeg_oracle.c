/* jpeg_oracle.c -- TEST INFRASTRUCTURE ONLY: nothing under cald_amd/ includes, links or calls this file.
 *
 * CPU restatement of the JPEG decode the reference's input side performs:
 * torchvision.datasets.VOCDetection.__getitem__ -> PIL.Image.open(path).convert('RGB')
 * (/root/reference/detection/voc_utils.py:47-58, cald_train.py:434), i.e. Pillow's bundled libjpeg-turbo with its
 * defaults: baseline sequential Huffman, JDCT_ISLOW inverse DCT, "fancy" (triangle) chroma upsampling, fixed-point
 * YCbCr -> RGB.  The algorithm lives in a third-party dependency that is not part of /root/reference
 * (libjpeg-turbo, libjpeg API 6.2, inside Pillow 12.2.0 in this image); this file restates its published
 * algorithm (ITU T.81 Huffman decoding; jidctint.c 13-bit fixed-point LL&M IDCT; jdsample.c h2v1/h2v2 triangle
 * filters; jdcolor.c 16-bit fixed-point colour tables) and is PINNED against Pillow's output on the files under
 * tests/golden/jpeg_*.npz and, on any box with Pillow, against PIL directly (tests/test_jpeg.py).
 *
 * Supported: 8-bit baseline/extended-sequential Huffman (SOF0/SOF1), one interleaved scan, grayscale or YCbCr with
 * luma sampling 1x1 / 2x1 / 2x2 and chroma 1x1, restart intervals.  Anything else returns ORC_JPEG_UNSUPPORTED.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))
#define ORC_JPEG_OK 0
#define ORC_JPEG_BAD -1
#define ORC_JPEG_UNSUPPORTED -2

typedef struct {
    int set;
    uint8_t bits[17];
    uint8_t vals[256];
    int mincode[17], maxcode[17], valptr[17];
} HuffTab;

typedef struct {
    int id, h, v, tq, td, ta;
    int bw, bh;   /* blocks per row / column in the MCU-padded plane */
    int dw, dh;   /* downsampled_width / _height (real samples) */
    uint8_t* plane;
    int pred;
} Comp;

typedef struct {
    int W, H, nc, hmax, vmax, mcux, mcuy, restart;
    Comp c[3];
    uint16_t q[4][64];
    int qset[4];
    HuffTab dc[4], ac[4];
    const uint8_t* scan;
    const uint8_t* end;
    int saw_jfif, saw_adobe, adobe_transform;
} Jpg;

static int zigzag[64];
static void init_zigzag(void) {
    /* natural_order[k]: position (row*8+col) of the k-th coefficient of the zigzag scan (T.81 figure A.6) */
    int k = 0;
    for (int s = 0; s < 15; s++) {
        if (s & 1) { for (int r = 0; r < 8; r++) { int c = s - r; if (c >= 0 && c < 8) zigzag[k++] = r * 8 + c; } }
        else       { for (int c = 0; c < 8; c++) { int r = s - c; if (r >= 0 && r < 8) zigzag[k++] = r * 8 + c; } }
    }
}

static void huff_build(HuffTab* t) {
    /* T.81 annex C (code sizes -> codes) and F.2.2.3 (mincode / maxcode / valptr) */
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        t->valptr[l] = k;
        t->mincode[l] = code;
        code += t->bits[l];
        k += t->bits[l];
        t->maxcode[l] = t->bits[l] ? code - 1 : -1;
        code <<= 1;
    }
}

static int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

static int parse(const uint8_t* d, size_t n, Jpg* j) {
    memset(j, 0, sizeof(*j));
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return ORC_JPEG_BAD;
    size_t p = 2;
    int have_sof = 0;
    while (p + 4 <= n) {
        if (d[p] != 0xFF) return ORC_JPEG_BAD;
        while (p < n && d[p] == 0xFF) p++;      /* fill bytes */
        if (p >= n) return ORC_JPEG_BAD;
        const int m = d[p++];
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) return ORC_JPEG_BAD;     /* EOI before SOS */
        if (p + 2 > n) return ORC_JPEG_BAD;
        const int len = rd16(d + p);
        if (len < 2 || p + len > n) return ORC_JPEG_BAD;
        const uint8_t* s = d + p + 2;
        const int sl = len - 2;
        if (m == 0xDB) {
            int o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15; o++;
                if (tq > 3) return ORC_JPEG_BAD;
                if (o + (pq ? 128 : 64) > sl) return ORC_JPEG_BAD;
                for (int i = 0; i < 64; i++) {
                    const int v = pq ? rd16(s + o + 2 * i) : s[o + i];
                    j->q[tq][zigzag[i]] = (uint16_t)v;
                }
                o += pq ? 128 : 64;
                j->qset[tq] = 1;
            }
        } else if (m == 0xC4) {
            int o = 0;
            while (o < sl) {
                if (o + 17 > sl) return ORC_JPEG_BAD;
                const int tc = s[o] >> 4, th = s[o] & 15; o++;
                if (tc > 1 || th > 3) return ORC_JPEG_BAD;
                HuffTab* t = tc ? &j->ac[th] : &j->dc[th];
                int cnt = 0;
                t->bits[0] = 0;
                for (int i = 1; i <= 16; i++) { t->bits[i] = s[o + i - 1]; cnt += t->bits[i]; }
                o += 16;
                if (cnt > 256 || o + cnt > sl) return ORC_JPEG_BAD;
                memcpy(t->vals, s + o, cnt);
                o += cnt;
                huff_build(t);
                t->set = 1;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6) return ORC_JPEG_BAD;
            if (s[0] != 8) return ORC_JPEG_UNSUPPORTED;
            j->H = rd16(s + 1); j->W = rd16(s + 3); j->nc = s[5];
            if (j->H <= 0 || j->W <= 0) return ORC_JPEG_BAD;
            if (j->nc != 1 && j->nc != 3) return ORC_JPEG_UNSUPPORTED;
            if (sl < 6 + 3 * j->nc) return ORC_JPEG_BAD;
            for (int i = 0; i < j->nc; i++) {
                j->c[i].id = s[6 + 3 * i];
                j->c[i].h = s[7 + 3 * i] >> 4; j->c[i].v = s[7 + 3 * i] & 15;
                j->c[i].tq = s[8 + 3 * i];
                if (j->c[i].tq > 3) return ORC_JPEG_BAD;
            }
            have_sof = 1;
        } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            return ORC_JPEG_UNSUPPORTED;        /* progressive, lossless, arithmetic, hierarchical */
        } else if (m == 0xDD) {
            if (sl < 2) return ORC_JPEG_BAD;
            j->restart = rd16(s);
        } else if (m == 0xE0) {
            if (sl >= 5 && s[0] == 'J' && s[1] == 'F' && s[2] == 'I' && s[3] == 'F' && s[4] == 0) j->saw_jfif = 1;
        } else if (m == 0xEE) {
            if (sl >= 12 && s[0] == 'A' && s[1] == 'd' && s[2] == 'o' && s[3] == 'b' && s[4] == 'e') {
                j->saw_adobe = 1; j->adobe_transform = s[11];
            }
        } else if (m == 0xDA) {
            if (!have_sof) return ORC_JPEG_BAD;
            if (sl < 1 || s[0] != j->nc) return ORC_JPEG_UNSUPPORTED;     /* non-interleaved multi-scan */
            if (sl < 1 + 2 * j->nc + 3) return ORC_JPEG_BAD;
            for (int i = 0; i < j->nc; i++) {
                if (s[1 + 2 * i] != j->c[i].id) return ORC_JPEG_UNSUPPORTED;
                j->c[i].td = s[2 + 2 * i] >> 4; j->c[i].ta = s[2 + 2 * i] & 15;
                if (j->c[i].td > 3 || j->c[i].ta > 3) return ORC_JPEG_BAD;
                if (!j->dc[j->c[i].td].set || !j->ac[j->c[i].ta].set || !j->qset[j->c[i].tq]) return ORC_JPEG_BAD;
            }
            j->scan = d + p + len;
            j->end = d + n;
            break;
        }
        p += len;
    }
    if (!j->scan) return ORC_JPEG_BAD;
    /* colour space rule of libjpeg's default_decompress_parms */
    if (j->nc == 3) {
        int ycc = 1;
        if (j->saw_jfif) ycc = 1;
        else if (j->saw_adobe) ycc = (j->adobe_transform != 0);
        else if (j->c[0].id == 'R' && j->c[1].id == 'G' && j->c[2].id == 'B') ycc = 0;
        if (!ycc) return ORC_JPEG_UNSUPPORTED;
    }
    j->hmax = j->c[0].h; j->vmax = j->c[0].v;
    if (j->nc == 1) { j->c[0].h = j->c[0].v = 1; j->hmax = j->vmax = 1; }   /* single-component scans are never interleaved */
    else {
        if (j->c[1].h != 1 || j->c[1].v != 1 || j->c[2].h != 1 || j->c[2].v != 1) return ORC_JPEG_UNSUPPORTED;
        if (!((j->hmax == 1 && j->vmax == 1) || (j->hmax == 2 && j->vmax == 1) || (j->hmax == 2 && j->vmax == 2)))
            return ORC_JPEG_UNSUPPORTED;
    }
    j->mcux = (j->W + 8 * j->hmax - 1) / (8 * j->hmax);
    j->mcuy = (j->H + 8 * j->vmax - 1) / (8 * j->vmax);
    for (int i = 0; i < j->nc; i++) {
        Comp* c = &j->c[i];
        c->bw = j->mcux * c->h; c->bh = j->mcuy * c->v;
        c->dw = (j->W * c->h + j->hmax - 1) / j->hmax;
        c->dh = (j->H * c->v + j->vmax - 1) / j->vmax;
    }
    return ORC_JPEG_OK;
}

/* ---- entropy decoding (T.81 F.2.2) ---- */
typedef struct { const uint8_t* p; const uint8_t* end; uint32_t acc; int n; } Bits;

static void fill(Bits* b) {
    while (b->n <= 24) {
        int byte = 0;
        if (b->p < b->end) {
            if (b->p[0] == 0xFF) {
                if (b->p + 1 < b->end && b->p[1] == 0x00) { byte = 0xFF; b->p += 2; }
                else byte = 0;                                   /* marker: feed zeros, do not advance */
            } else byte = *b->p++;
        }
        b->acc |= (uint32_t)byte << (24 - b->n);
        b->n += 8;
    }
}
static int getbits(Bits* b, int s) {
    if (s == 0) return 0;
    fill(b);
    const int v = (int)(b->acc >> (32 - s));
    b->acc <<= s; b->n -= s;
    return v;
}
static int decode_sym(Bits* b, const HuffTab* t) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | getbits(b, 1);
        if (t->maxcode[l] >= 0 && code <= t->maxcode[l] && code >= t->mincode[l])
            return t->vals[t->valptr[l] + code - t->mincode[l]];
    }
    return 0;   /* corrupt code: libjpeg substitutes zero */
}
static int extend(int r, int s) { return r < (1 << (s - 1)) ? r - (1 << s) + 1 : r; }

/* ---- jidctint.c: LL&M inverse DCT, CONST_BITS 13, PASS1_BITS 2 ---- */
#define F_0_298631336 2446
#define F_0_390180644 3196
#define F_0_541196100 4433
#define F_0_765366865 6270
#define F_0_899976223 7373
#define F_1_175875602 9633
#define F_1_501321110 12299
#define F_1_847759065 15137
#define F_1_961570560 16069
#define F_2_053119869 16819
#define F_2_562915447 20995
#define F_3_072711026 25172
static int32_t descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }
static uint8_t range_limit_idct(int32_t x) {
    /* sample_range_limit + CENTERJSAMPLE indexed with (x & 1023) */
    x &= 1023;
    if (x < 128) return (uint8_t)(x + 128);
    if (x < 512) return 255;
    if (x < 896) return 0;
    return (uint8_t)(x - 896);
}
static void idct_1d(const int32_t in[8], int32_t out[8], int shift, int pass1) {
    int32_t z1, z2, z3, z4, z5, tmp0, tmp1, tmp2, tmp3, tmp10, tmp11, tmp12, tmp13;
    z2 = in[2]; z3 = in[6];
    z1 = (z2 + z3) * F_0_541196100;
    tmp2 = z1 + z3 * (-F_1_847759065);
    tmp3 = z1 + z2 * F_0_765366865;
    z2 = in[0]; z3 = in[4];
    tmp0 = (z2 + z3) * 8192; tmp1 = (z2 - z3) * 8192;
    tmp10 = tmp0 + tmp3; tmp13 = tmp0 - tmp3; tmp11 = tmp1 + tmp2; tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; z4 = tmp1 + tmp3;
    z5 = (z3 + z4) * F_1_175875602;
    tmp0 = tmp0 * F_0_298631336; tmp1 = tmp1 * F_2_053119869; tmp2 = tmp2 * F_3_072711026; tmp3 = tmp3 * F_1_501321110;
    z1 = z1 * (-F_0_899976223); z2 = z2 * (-F_2_562915447); z3 = z3 * (-F_1_961570560); z4 = z4 * (-F_0_390180644);
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    (void)pass1;
    out[0] = descale(tmp10 + tmp3, shift); out[7] = descale(tmp10 - tmp3, shift);
    out[1] = descale(tmp11 + tmp2, shift); out[6] = descale(tmp11 - tmp2, shift);
    out[2] = descale(tmp12 + tmp1, shift); out[5] = descale(tmp12 - tmp1, shift);
    out[3] = descale(tmp13 + tmp0, shift); out[4] = descale(tmp13 - tmp0, shift);
}
static void idct_block(const int16_t coef[64], const uint16_t q[64], uint8_t* out, int stride) {
    int32_t ws[64], in[8], o[8];
    for (int c = 0; c < 8; c++) {
        for (int r = 0; r < 8; r++) in[r] = (int32_t)coef[r * 8 + c] * (int32_t)q[r * 8 + c];
        idct_1d(in, o, 13 - 2, 1);
        for (int r = 0; r < 8; r++) ws[r * 8 + c] = o[r];
    }
    for (int r = 0; r < 8; r++) {
        idct_1d(ws + r * 8, o, 13 + 2 + 3, 0);
        for (int c = 0; c < 8; c++) out[r * stride + c] = range_limit_idct(o[c]);
    }
}

/* ---- jdsample.c triangle ("fancy") upsampling of one chroma sample at output pixel (x, y) ---- */
static int chroma_at(const Comp* c, int hs, int vs, int x, int y) {
    const uint8_t* P = c->plane;
    const int stride = c->bw * 8;
    if (hs == 1 && vs == 1) return P[y * stride + x];
    const int fancy = c->dw > 2;
    if (hs == 2 && vs == 1) {
        const uint8_t* row = P + (size_t)y * stride;
        const int cx = x >> 1;
        if (!fancy) return row[cx];
        if (x & 1) return cx == c->dw - 1 ? row[cx] : (row[cx] * 3 + row[cx + 1] + 2) >> 2;
        return cx == 0 ? row[cx] : (row[cx] * 3 + row[cx - 1] + 1) >> 2;
    }
    /* h2v2 */
    const int cy = y >> 1, cx = x >> 1;
    if (!fancy) return P[(size_t)cy * stride + cx];
    int ny = (y & 1) ? cy + 1 : cy - 1;
    if (ny < 0) ny = 0;
    if (ny > c->dh - 1) ny = c->dh - 1;
    const uint8_t* r0 = P + (size_t)cy * stride;
    const uint8_t* r1 = P + (size_t)ny * stride;
    const int cur = r0[cx] * 3 + r1[cx];
    if (x & 1) {
        if (cx == c->dw - 1) return (cur * 4 + 7) >> 4;
        return (cur * 3 + (r0[cx + 1] * 3 + r1[cx + 1]) + 7) >> 4;
    }
    if (cx == 0) return (cur * 4 + 8) >> 4;
    return (cur * 3 + (r0[cx - 1] * 3 + r1[cx - 1]) + 8) >> 4;
}
static uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

ORC_API int orc_jpeg_info(const uint8_t* data, size_t n, int* H, int* W, int* ncomp) {
    if (!zigzag[1]) init_zigzag();
    Jpg j;
    const int rc = parse(data, n, &j);
    if (rc) return rc;
    *H = j.H; *W = j.W; *ncomp = j.nc;
    return 0;
}

/* rgb: H*W*3 bytes (HWC) */
ORC_API int orc_jpeg_decode(const uint8_t* data, size_t n, uint8_t* rgb) {
    if (!zigzag[1]) init_zigzag();
    Jpg j;
    int rc = parse(data, n, &j);
    if (rc) return rc;
    for (int i = 0; i < j.nc; i++) {
        j.c[i].plane = (uint8_t*)calloc((size_t)j.c[i].bw * 8 * j.c[i].bh * 8, 1);
        j.c[i].pred = 0;
    }
    Bits b = {j.scan, j.end, 0, 0};
    int16_t coef[64];
    int mcus_left = j.restart, next_rst = 0;
    for (int my = 0; my < j.mcuy; my++)
        for (int mx = 0; mx < j.mcux; mx++) {
            if (j.restart && mcus_left == 0) {
                /* byte-align, expect RSTn, reset predictors (T.81 F.2.1.3.1) */
                b.acc = 0; b.n = 0;
                while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) b.p++;
                if (b.p + 1 < b.end) b.p += 2;
                next_rst = (next_rst + 1) & 7;
                for (int i = 0; i < j.nc; i++) j.c[i].pred = 0;
                mcus_left = j.restart;
            }
            for (int i = 0; i < j.nc; i++) {
                Comp* c = &j.c[i];
                for (int v = 0; v < c->v; v++)
                    for (int h = 0; h < c->h; h++) {
                        memset(coef, 0, sizeof(coef));
                        int s = decode_sym(&b, &j.dc[c->td]);
                        if (s) { const int r = getbits(&b, s); c->pred += extend(r, s); }
                        coef[0] = (int16_t)c->pred;
                        for (int k = 1; k < 64; k++) {
                            const int rs = decode_sym(&b, &j.ac[c->ta]);
                            const int r = rs >> 4; s = rs & 15;
                            if (s) {
                                k += r;
                                const int val = extend(getbits(&b, s), s);
                                if (k < 64) coef[zigzag[k]] = (int16_t)val;
                            } else {
                                if (r == 15) k += 15; else break;
                            }
                        }
                        const int by = my * c->v + v, bx = mx * c->h + h;
                        idct_block(coef, j.q[c->tq], c->plane + ((size_t)by * 8 * c->bw + bx) * 8, c->bw * 8);
                    }
            }
            if (j.restart) mcus_left--;
        }
    /* upsample + jdcolor.c ycc_rgb_convert */
    const int stride0 = j.c[0].bw * 8;
    for (int y = 0; y < j.H; y++)
        for (int x = 0; x < j.W; x++) {
            const int Y = j.c[0].plane[(size_t)y * stride0 + x];
            uint8_t* o = rgb + ((size_t)y * j.W + x) * 3;
            if (j.nc == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; continue; }
            const int cb = chroma_at(&j.c[1], j.hmax, j.vmax, x, y) - 128;
            const int cr = chroma_at(&j.c[2], j.hmax, j.vmax, x, y) - 128;
            const int r = Y + ((91881 * cr + 32768) >> 16);
            const int g = Y + ((-22554 * cb + 32768 + -46802 * cr) >> 16);
            const int bl = Y + ((116130 * cb + 32768) >> 16);
            o[0] = clamp8(r); o[1] = clamp8(g); o[2] = clamp8(bl);
        }
    for (int i = 0; i < j.nc; i++) free(j.c[i].plane);
    return 0;
}
