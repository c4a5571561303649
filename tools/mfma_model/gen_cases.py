"""gen_cases.py -- deterministic operand families for tools/mfma_model/probe.hip (test infrastructure of oracle/mfma_f16_model.h).

Each case is one 16-term dot product of fp16 operands plus an fp32 addend, the unit v_mfma_f32_32x32x16_f16 evaluates per output.
The families are DIRECTED: each isolates one property of the datapath (alignment width, which terms share an alignment group, whether
the addend joins the group, truncation vs rounding, what cancellation keeps), because random operands of similar magnitude hit the
interesting paths rarely (profiles/HISTORY.md: three guesses, best 94 %).

    python tools/mfma_model/gen_cases.py OUTDIR        writes OUTDIR/cases.bin and OUTDIR/index.json
    cases.bin = int64 n | n x 16 u16 (A) | n x 16 u16 (B) | n x u32 (C)

The same seed gives the same bytes here and on the GPU box (same image, same numpy); only the results travel back.
"""
import json
import sys

import numpy as np

RNG = np.random.default_rng(20260929)


def h(sign, e, m):
    """fp16 bits of (-1)^sign * 2^e * (1 + m / 1024), e in [-14, 15]; e == -15 means subnormal m * 2^-24."""
    sign = np.asarray(sign, np.int64); e = np.asarray(e, np.int64); m = np.asarray(m, np.int64)
    return ((sign << 15) | ((e + 15) << 10) | m).astype(np.uint16)


def f32(sign, e, m):
    """fp32 bits of (-1)^sign * 2^e * (1 + m / 2^23); e == -127 -> subnormal / zero."""
    sign = np.asarray(sign, np.int64); e = np.asarray(e, np.int64); m = np.asarray(m, np.int64)
    return ((sign << 31) | ((e + 127) << 23) | m).astype(np.uint32)


class Cases:
    def __init__(self):
        self.A, self.B, self.C, self.index, self.n = [], [], [], {}, 0

    def add(self, name, A, B, C, meta=None):
        A = np.ascontiguousarray(A, np.uint16).reshape(-1, 16); B = np.ascontiguousarray(B, np.uint16).reshape(-1, 16)
        C = np.ascontiguousarray(C, np.uint32).reshape(-1)
        assert A.shape == B.shape and A.shape[0] == C.shape[0]
        self.index[name] = {"start": self.n, "n": int(A.shape[0]), **(meta or {})}
        self.A.append(A); self.B.append(B); self.C.append(C); self.n += A.shape[0]

    def write(self, outdir):
        A = np.concatenate(self.A); B = np.concatenate(self.B); C = np.concatenate(self.C)
        with open(f"{outdir}/cases.bin", "wb") as f:
            f.write(np.int64(self.n).tobytes()); f.write(A.tobytes()); f.write(B.tobytes()); f.write(C.tobytes())
        with open(f"{outdir}/index.json", "w") as f:
            json.dump(self.index, f, indent=1)


def rmant(n, bits=10):
    return RNG.integers(0, 1 << bits, n)


def product_at(n, pos, ea, eb, sign=None, ma=None, mb=None):
    """A, B with one non-zero product at position pos[i]: (-1)^s 2^(ea+eb) (1+ma/1024)(1+mb/1024)"""
    A = np.zeros((n, 16), np.uint16); B = np.zeros((n, 16), np.uint16)
    s = RNG.integers(0, 2, n) if sign is None else np.broadcast_to(sign, (n,))
    ma = rmant(n) if ma is None else np.broadcast_to(ma, (n,)); mb = rmant(n) if mb is None else np.broadcast_to(mb, (n,))
    r = np.arange(n)
    A[r, pos] = h(s, np.broadcast_to(ea, (n,)), ma); B[r, pos] = h(0, np.broadcast_to(eb, (n,)), mb)
    return A, B


def split_exp(e):
    """a pair (ea, eb) of fp16 exponents with ea + eb == e, both in [-14, 15]"""
    e = np.asarray(e)
    ea = np.clip(e // 2, -14, 15); eb = e - ea
    assert (eb >= -14).all() and (eb <= 15).all(), (e.min(), e.max())
    return ea, eb


def fam_single(c):
    """one product + addend, exponent difference d = e_prod - e_c in [-50, 50]: does the addend join the alignment, how wide, how rounded"""
    per = 256
    ds = np.repeat(np.arange(-50, 51), per); n = ds.size
    ep = RNG.integers(-8, 9, n)
    ea, eb = split_exp(ep)
    pos = RNG.integers(0, 16, n)
    A, B = product_at(n, pos, ea, eb)
    C = f32(RNG.integers(0, 2, n), ep - ds, RNG.integers(0, 1 << 23, n))
    c.add("single", A, B, C, {"per": per, "d0": -50})
    # the same with mantissas that make half-way cases likely: few set bits
    ma = 1 << RNG.integers(0, 10, n); mb = 1 << RNG.integers(0, 10, n)
    A, B = product_at(n, pos, ea, eb, ma=ma, mb=mb)
    C = f32(RNG.integers(0, 2, n), ep - ds, (1 << RNG.integers(0, 23, n)) | (RNG.integers(0, 2, n) << 22))
    c.add("single_sparse", A, B, C, {"per": per, "d0": -50})


def fam_pair(c):
    """two products at positions (i, j), exponent offset d, addend 0 / tiny / comparable: which positions share a group, group width"""
    per = 6
    ii, jj, dd = np.meshgrid(np.arange(16), np.arange(16), np.arange(0, 46), indexing="ij")
    keep = ii != jj
    ii, jj, dd = [np.repeat(x[keep], per) for x in (ii, jj, dd)]
    n = ii.size
    for name, cmode in (("pair_c0", 0), ("pair_csmall", 1), ("pair_cbig", 2), ("pair_cmid", 3)):
        ep = RNG.integers(18, 23, n)
        ea, eb = split_exp(ep)
        A, B = product_at(n, ii, ea, eb)
        ea2, eb2 = split_exp(ep - dd)
        A2, B2 = product_at(n, jj, ea2, eb2)
        A |= A2; B |= B2
        if cmode == 0: C = np.zeros(n, np.uint32)
        elif cmode == 1: C = f32(RNG.integers(0, 2, n), ep - 30 - RNG.integers(0, 20, n), RNG.integers(0, 1 << 23, n))
        elif cmode == 2: C = f32(RNG.integers(0, 2, n), ep + RNG.integers(1, 30, n), RNG.integers(0, 1 << 23, n))
        else: C = f32(RNG.integers(0, 2, n), ep - RNG.integers(0, 30, n), RNG.integers(0, 1 << 23, n))
        c.add(name, A, B, C, {"per": per})


def fam_cancel(c):
    """X at i, -X at j (exactly), tiny product at k with offset d below X, addend 0 or tiny: what survives the cancellation"""
    n = 120000
    i = RNG.integers(0, 16, n); j = (i + RNG.integers(1, 16, n)) % 16
    k = (i + RNG.integers(1, 16, n)) % 16
    k = np.where(k == j, (k + 1) % 16, k); k = np.where(k == i, (k + 1) % 16, k)
    assert ((k != i) & (k != j) & (i != j)).all()
    d = RNG.integers(0, 50, n)
    ep = RNG.integers(14, 23, n)
    ea, eb = split_exp(ep)
    ma, mb = rmant(n), rmant(n)
    A, B = product_at(n, i, ea, eb, sign=0, ma=ma, mb=mb)
    A2, B2 = product_at(n, j, ea, eb, sign=1, ma=ma, mb=mb)
    ea3, eb3 = split_exp(np.maximum(ep - d, -28))
    A3, B3 = product_at(n, k, ea3, eb3)
    A |= A2 | A3; B |= B2 | B3
    C = np.where(RNG.integers(0, 2, n) == 0, 0, f32(RNG.integers(0, 2, n), ep - d - RNG.integers(-3, 30, n), RNG.integers(0, 1 << 23, n))).astype(np.uint32)
    c.add("cancel", A, B, C)


def fam_manytiny(c):
    """one large product and 15 small ones at a common offset d: per-term truncation vs an exact group sum"""
    n = 100000
    A = np.zeros((n, 16), np.uint16); B = np.zeros((n, 16), np.uint16)
    big = RNG.integers(0, 16, n); d = RNG.integers(10, 40, n); ep = RNG.integers(10, 20, n)
    same_sign = RNG.integers(0, 2, n)
    for p in range(16):
        isbig = big == p
        e = np.where(isbig, ep, np.maximum(ep - d, -28))
        ea, eb = split_exp(e)
        s = np.where(isbig, 0, np.where(same_sign == 1, 0, RNG.integers(0, 2, n)))
        A[:, p] = h(s, ea, rmant(n)); B[:, p] = h(0, eb, rmant(n))
    cm = RNG.integers(0, 3, n)
    C = np.where(cm == 0, 0, f32(RNG.integers(0, 2, n), np.where(cm == 1, ep - d, ep + RNG.integers(-2, 3, n)), RNG.integers(0, 1 << 23, n))).astype(np.uint32)
    c.add("manytiny", A, B, C)


def fam_random(c):
    """random operands with a +-s exponent spread per operand; addend comparable / much larger / much smaller / zero"""
    for s in (0, 1, 2, 4, 6, 7):
        n = 150000
        ea = RNG.integers(-s, s + 1, (n, 16)); eb = RNG.integers(-s, s + 1, (n, 16))
        A = h(RNG.integers(0, 2, (n, 16)), ea, rmant((n, 16))); B = h(RNG.integers(0, 2, (n, 16)), eb, rmant((n, 16)))
        cm = RNG.integers(0, 4, n)
        ec = np.where(cm == 0, RNG.integers(-2, 4, n), np.where(cm == 1, RNG.integers(4, 16, n), RNG.integers(-20, -2, n)))
        C = np.where(cm == 3, 0, f32(RNG.integers(0, 2, n), ec, RNG.integers(0, 1 << 23, n))).astype(np.uint32)
        c.add(f"random_s{s}", A, B, C)
    # sparse: a random subset of positions is zero
    n = 150000
    A = h(RNG.integers(0, 2, (n, 16)), RNG.integers(-6, 7, (n, 16)), rmant((n, 16))); B = h(RNG.integers(0, 2, (n, 16)), RNG.integers(-6, 7, (n, 16)), rmant((n, 16)))
    A = np.where(RNG.integers(0, 3, (n, 16)) == 0, A, 0).astype(np.uint16)
    C = f32(RNG.integers(0, 2, n), RNG.integers(-16, 8, n), RNG.integers(0, 1 << 23, n))
    c.add("random_sparse", A, B, C)


def fam_f16x3(c):
    """operands shaped like CALD_PRECISION_F16X3's: hi / lo halves of scaled fp32 values, the addend a running sum much larger than a product"""
    n = 300000
    x = (RNG.standard_normal((n, 16)) * np.exp2(RNG.integers(0, 8, (n, 1)))).astype(np.float32)
    w = (RNG.standard_normal((n, 16)) * np.exp2(RNG.integers(4, 12, (n, 1)))).astype(np.float32)
    xh = x.astype(np.float16); xl = (x - xh.astype(np.float32)).astype(np.float16)
    wh = w.astype(np.float16); wl = (w - wh.astype(np.float32)).astype(np.float16)
    which = RNG.integers(0, 3, n)[:, None]
    A = np.where(which == 0, xl, xh).view(np.uint16); B = np.where(which == 1, wl, wh).view(np.uint16)
    acc = (RNG.standard_normal(n) * np.exp2(RNG.integers(8, 24, n))).astype(np.float32)
    acc[RNG.integers(0, 8, n) == 0] = 0.0
    c.add("f16x3_like", A, B, acc.view(np.uint32))


def fam_special(c):
    """fp16 subnormal operands, fp32 subnormal addends, signed zeros, results in the fp32 subnormal range"""
    n = 60000
    A = np.zeros((n, 16), np.uint16); B = np.zeros((n, 16), np.uint16)
    # subnormal a (e field 0) times normal b, a few positions
    npos = RNG.integers(1, 5, n)
    for t in range(4):
        p = RNG.integers(0, 16, n); on = t < npos
        sub = RNG.integers(0, 2, n) == 0
        av = np.where(sub, (RNG.integers(0, 2, n) << 15) | rmant(n), h(RNG.integers(0, 2, n), RNG.integers(-14, -8, n), rmant(n)))
        bv = h(RNG.integers(0, 2, n), RNG.integers(-14, 4, n), rmant(n))
        r = np.arange(n)
        A[r[on], p[on]] = av[on].astype(np.uint16); B[r[on], p[on]] = bv[on]
    cm = RNG.integers(0, 4, n)
    C = np.where(cm == 0, 0, np.where(cm == 1, RNG.integers(0, 1 << 23, n) | (RNG.integers(0, 2, n) << 31),      # fp32 subnormal
                 np.where(cm == 2, f32(RNG.integers(0, 2, n), RNG.integers(-126, -100, n), RNG.integers(0, 1 << 23, n)),
                          f32(RNG.integers(0, 2, n), RNG.integers(-40, -10, n), RNG.integers(0, 1 << 23, n))))).astype(np.uint32)
    c.add("special_small", A, B, C)
    # signed zeros: all products zero with chosen signs, addend +-0
    n = 4096
    A = h(RNG.integers(0, 2, (n, 16)), 0, 0); A[:, :] = np.where(RNG.integers(0, 2, (n, 16)) == 0, A, A & 0x8000)
    B = (RNG.integers(0, 2, (n, 16)) << 15).astype(np.uint16)
    C = (RNG.integers(0, 2, n) << 31).astype(np.uint32)
    c.add("special_zero", A, B, C)


def main():
    outdir = sys.argv[1]
    c = Cases()
    fam_single(c); fam_pair(c); fam_cancel(c); fam_manytiny(c); fam_random(c); fam_f16x3(c); fam_special(c)
    c.write(outdir)
    print({k: v["n"] for k, v in c.index.items()}, "total", c.n)


if __name__ == "__main__":
    main()
