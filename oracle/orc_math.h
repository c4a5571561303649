/* TEST INFRASTRUCTURE ONLY (oracle).  Deterministic float32 elementary functions.
 *
 * The reference calls torch.exp / torch.sigmoid / torch.softmax / numpy.log
 * (frcnn_la.py:40, retinanet_cal.py:411, cald_train.py:214 via scipy.stats.entropy).
 * Those library functions differ from one another in the last ulp; the oracle
 * restates them as fixed fmaf polynomials so that the same operation sequence
 * can be reproduced bit-for-bit by the HIP kernels.  Accuracy: <= 2 ulp, i.e.
 * far inside the 1e-4 float tolerance of BASELINE.json.
 * Build with -ffp-contract=off: every fused multiply-add below is explicit.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float orc_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t orc_f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* e^x.  x > 88.72 -> +inf, x < -87 -> 0 (no denormal results by contract). */
static inline float orc_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return INFINITY;
    if (x < -87.0f) return 0.0f;
    float n = rintf(x * 1.44269504f);
    float r = fmaf(n, -0.693145752f, x);
    r = fmaf(n, -1.42860677e-6f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    int ni = (int)n;
    int n1 = ni / 2, n2 = ni - n1; /* both in [-64, 64]: exact power-of-two scalings */
    y = y * orc_bits2f((uint32_t)(n1 + 127) << 23);
    y = y * orc_bits2f((uint32_t)(n2 + 127) << 23);
    return y;
}

/* natural log.  x==0 -> -inf, x<0 -> nan, inf -> inf. */
static inline float orc_logf(float x) {
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    uint32_t u = orc_f2bits(x);
    if (u < 0x00800000u) { /* denormal: scale by 2^23 */
        x = x * 8388608.0f; u = orc_f2bits(x); e = -23;
    }
    e += (int)(u >> 23) - 126;
    float m = orc_bits2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
    float f;
    if (m < 0.70710678f) { e -= 1; f = (m + m) - 1.0f; } else { f = m - 1.0f; }
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float y = (p * f) * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float r = f + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

/* sin and cos for x in [0, 2*pi] (Box-Muller angle): quadrant reduction + Cephes minimax polynomials */
static inline void orc_sincosf(float x, float* s, float* c) {
    float q = rintf(x * 0.636619772f);                 /* 2/pi */
    float r = fmaf(q, -1.5703125f, x);
    r = fmaf(q, -4.837512969970703125e-4f, r);
    r = fmaf(q, -7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = fmaf(ps, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    float sn = fmaf(ps * z, r, r);
    float pc = 2.443315711809948e-5f;
    pc = fmaf(pc, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    float cs = fmaf(pc * z, z, fmaf(z, -0.5f, 1.0f));
    int qi = ((int)q) & 3;
    float ss = (qi & 1) ? cs : sn, cc = (qi & 1) ? sn : cs;
    if (qi == 2 || qi == 3) ss = -ss;
    if (qi == 1 || qi == 2) cc = -cc;
    *s = ss; *c = cc;
}
static inline float orc_sigmoidf(float x) { return 1.0f / (1.0f + orc_expf(-x)); }
static inline float orc_log2f(float x) { return orc_logf(x) * 1.44269504f; }

#endif
