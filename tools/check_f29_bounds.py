#!/usr/bin/env python3
"""Machine check of the F29 bound discipline (zk_field29.h) over every formula of zk_curve29.h.

Each value carries (LB, VB): LB bounds every limb below the top one, VB * p bounds the integer.  The operations
mirror the C++ one to one and assert what the C++ relies on:
  mul : 9 LB(a) LB(b) + 9 * 2^58 + 2^36 < 2^64                      (64-bit column accumulator)
  sub : LB(b) <= k (2^29 - 1) + slack(N+)   and   VB(b) + 1 <= multiple   (no negative limb, top limb included)
        LB(a) + max bias limb < 2^32                                 (u32 words)
Run: python tools/check_f29_bounds.py   (also executed by tests/test_f29_bounds.py)."""
import sys

# set per shape by configure(): 9 x 29 bits (254/255-bit fields) or 14 x 28 bits (BLS12-381 Fq)
W = L = STRICT = NPLUS = P_OVER_R = TOP_UNIT = None


def configure(w, l, log2p_max, log2p_min):
    global W, L, STRICT, NPLUS, P_OVER_R, TOP_UNIT
    W, L = w, l
    STRICT = (1 << W) - 1
    NPLUS = STRICT + ((1 << (32 - W)) - 1)
    P_OVER_R = 2.0 ** (log2p_max - W * L)                      # upper bound of p / R'
    TOP_UNIT = int(2.0 ** (log2p_min - W * (L - 1)))           # lower bound of p >> (W (L-1)): the top limb of one p


class V:
    def __init__(self, lb, vb, name=""):
        self.lb, self.vb, self.name = lb, vb, name

    def __repr__(self):
        return "%s(LB=2^%.2f, VB=%.2f)" % (self.name, __import__("math").log2(max(self.lb, 1)), self.vb)


def mul(a, b, name="mul"):
    col = L * a.lb * b.lb + L * (1 << (2 * W)) + (1 << (64 - W + 1))
    assert col < 1 << 64, ("column overflow", name, a, b)
    r = V(STRICT, a.vb * b.vb * P_OVER_R + 1.0, name)
    assert r.vb < 20, ("value too large for KP table / top limb", name, r)
    return r


def mulacc(a, b, c, d, name="mulacc"):
    """(a b + c d) / R' with one reduction: both products in the same columns"""
    col = L * (a.lb * b.lb + c.lb * d.lb) + L * (1 << (2 * W)) + (1 << (64 - W + 1))
    assert col < 1 << 64, ("column overflow", name, a, b, c, d)
    r = V(STRICT, (a.vb * b.vb + c.vb * d.vb) * P_OVER_R + 1.0, name)
    assert r.vb < 20, ("value too large", name, r)
    return r


def add(a, b, name="add"):
    r = V(a.lb + b.lb, a.vb + b.vb, name)
    assert r.lb < 1 << 32
    return r


BIAS = {"4K1": (4, 1), "4K2": (4, 2), "16K2": (16, 2), "8K3": (8, 3), "8K2": (8, 2)}


def sub(a, subtrahends, bias, name="sub"):
    mult, k = BIAS[bias]
    tot_lb = sum(s.lb for s in subtrahends)
    tot_vb = sum(s.vb for s in subtrahends)
    # every bias limb below the top is >= k * 2^29 - k
    assert tot_lb <= k * (1 << W) - k, ("limb could go negative", name, tot_lb, k)
    # top limb: bias_8 = floor(mult p / 2^232) - k >= total top limbs of the subtrahends  <=  (mult - tot_vb) p / 2^232 >= k + 1
    assert (mult - tot_vb) * TOP_UNIT >= k + 1, ("top limb could go negative", name, mult, tot_vb)
    max_bias_limb = (1 << W) - 1 + k * (1 << W)
    r = V(a.lb + max_bias_limb, a.vb + mult, name)
    assert r.lb < 1 << 32, ("u32 overflow", name, r)
    return r


def norm(a, name="norm"):
    assert a.lb < 1 << 32
    return V(STRICT + (a.lb >> W), a.vb, name)


def stored(x, y, zz, zzz):
    assert x.lb <= NPLUS and x.vb < 12, x
    assert y.lb <= NPLUS and y.vb < 8, y
    assert zz.lb <= STRICT and zz.vb < 2 and zzz.lb <= STRICT and zzz.vb < 2


def check_shape(w, l, log2p_max, log2p_min):
    configure(w, l, log2p_max, log2p_min)
    print("== %d x %d-bit limbs, 2^%.2f < p < 2^%.2f" % (l, w, log2p_min, log2p_max))
    X1, Y1, ZZ1, ZZZ1 = V(NPLUS, 12, "X1"), V(NPLUS, 8, "Y1"), V(STRICT, 2, "ZZ1"), V(STRICT, 2, "ZZZ1")
    qx, qy = V(STRICT, 2, "qx"), V(NPLUS, 4, "qy")     # converted base; y possibly negated

    # aff_neg_if: 0 - y + 4p with y strict < 2p
    ny = norm(sub(V(0, 0), [V(STRICT, 2)], "4K1"))
    assert ny.lb <= NPLUS and ny.vb <= 4

    # first add into an empty accumulator
    stored(qx, qy, V(STRICT, 1.01), V(STRICT, 1.01))

    # ---- madd-2008-s
    u2, s2 = mul(qx, ZZ1), mul(qy, ZZZ1)
    p = sub(u2, [X1], "16K2")
    r = sub(s2, [Y1], "16K2")
    assert u2.vb + 16 <= 18 and 16 - X1.vb >= 4          # zero-filter range (4p, 18p) -> k in 5..17
    assert 16 - Y1.vb >= 8                                # (8p, 18p) -> k in 9..17
    p, r = norm(p), norm(r)
    pp = mul(p, p)
    ppp = mul(p, pp)
    qq = mul(X1, pp)
    rr = mul(r, r)
    x3 = norm(sub(rr, [ppp, qq, qq], "8K3"))
    t = sub(qq, [x3], "16K2")
    y3 = mulacc(r, t, norm(sub(V(0, 0), [Y1], "16K2")), ppp)
    stored(x3, y3, mul(ZZ1, pp), mul(ZZZ1, ppp))
    print("madd ok:", x3, y3)

    # ---- add-2008-s
    X2, Y2, ZZ2, ZZZ2 = V(NPLUS, 12), V(NPLUS, 8), V(STRICT, 2), V(STRICT, 2)
    u1, u2, s1, s2 = mul(X1, ZZ2), mul(X2, ZZ1), mul(Y1, ZZZ2), mul(Y2, ZZZ1)
    p, r = sub(u2, [u1], "4K1"), sub(s2, [s1], "4K1")
    assert u2.vb + 4 <= 6 and 4 - u1.vb >= 2              # (2p, 6p) -> k in 3..5
    p, r = norm(p), norm(r)
    pp = mul(p, p)
    ppp = mul(p, pp)
    qq = mul(u1, pp)
    rr = mul(r, r)
    x3 = norm(sub(rr, [ppp, qq, qq], "8K3"))
    t = sub(qq, [x3], "16K2")
    y3 = mulacc(r, t, norm(sub(V(0, 0), [s1], "4K1")), ppp)
    stored(x3, y3, mul(mul(ZZ1, ZZ2), pp), mul(mul(ZZZ1, ZZZ2), ppp))
    print("add ok: ", x3, y3)

    # ---- dbl-2008-s-1 (stored point) and mdbl-2008-s-1 (affine base)
    for tag, (x, y, zz, zzz) in (("dbl", (X1, Y1, ZZ1, ZZZ1)), ("mdbl", (qx, qy, None, None))):
        u = add(y, y)
        v = mul(u, u)
        w = mul(u, v)
        s = mul(x, v)
        t = mul(x, x)
        m = norm(add(add(t, t), t))
        x3 = norm(sub(mul(m, m), [s, s], "4K2"))
        t = sub(s, [x3], "16K2")
        y3 = mulacc(m, t, norm(sub(V(0, 0), [w], "4K1")), y)
        stored(x3, y3, mul(v, zz) if zz else v, mul(w, zzz) if zzz else w)
        print("%s ok: " % tag, x3, y3)

    # ---- conversion back: norm, multiply by FROM29 (strict, < p), canon handles < 20 p
    for c in (X1, Y1, ZZ1):
        assert mul(norm(c), V(STRICT, 1)).vb < 2


# ---------------------------------------------------------------------------------------------------------------
# Fq2 (G2 twists): values are pairs; the operations mirror Fe29x2 in zk_field29.h and the C29x2 formulas of zk_curve29.h
# ---------------------------------------------------------------------------------------------------------------
def sub2(a, subs, bias, name="sub2"):
    return tuple(sub(a[i], [x[i] for x in subs], bias, name) for i in (0, 1))


def norm2(a):
    return (norm(a[0]), norm(a[1]))


def add2(a, b):
    return (add(a[0], b[0]), add(a[1], b[1]))


def mul2(a, b, negbias, name="mul2"):
    nb1 = norm(sub(V(0, 0), [b[1]], negbias, name + ".neg"))
    return (mulacc(a[0], b[0], a[1], nb1, name), mulacc(a[0], b[1], a[1], b[0], name))


def sqr2(a, dbias, name="sqr2"):
    s = add(a[0], a[1])
    d = norm(sub(a[0], [a[1]], dbias, name + ".d"))
    return (mul(s, d, name), mul(add(a[0], a[0]), a[1], name))


def refresh2(a):
    one = V(STRICT, 1)
    return (mul(a[0], one, "refresh"), mul(a[1], one, "refresh"))


def krange(lo_sub, hi, bias):
    """integer range (in units of p, exclusive) of hi - lo_sub + bias, as (min k, max k) of a multiple of p inside it"""
    mult = BIAS[bias][0]
    lo = min(mult - c.vb for c in lo_sub)
    hi_ = max(c.vb + mult for c in hi)
    return int(lo) + 1, int(hi_)


def check_shape_ext2(w, l, log2p_max, log2p_min):
    configure(w, l, log2p_max, log2p_min)
    print("== Fq2 over %d x %d-bit limbs" % (l, w))
    SX, SY, SZ = 2.0, 6.9, 2.0

    def P2(lb, vb):
        return (V(lb, vb), V(lb, vb))

    def stored(x, y, zz, zzz, tag):
        # (vb is an exclusive bound on value / p, so "<=" keeps value < S p)
        assert all(c.lb <= STRICT and c.vb <= SX for c in x), (tag, x)
        assert all(c.lb <= NPLUS and c.vb <= SY for c in y), (tag, y)
        assert all(c.lb <= STRICT and c.vb <= SZ for c in zz + zzz), (tag, zz, zzz)
        print("%s ok: y < %.2f p" % (tag, max(c.vb for c in y)))

    X1, Y1, ZZ1, ZZZ1 = P2(STRICT, SX), P2(NPLUS, SY), P2(STRICT, SZ), P2(STRICT, SZ)
    qx, qy = P2(STRICT, 2), P2(NPLUS, 4)
    ny = norm2(sub2(P2(0, 0), [P2(STRICT, 2)], "4K1"))          # aff_neg_if
    assert all(c.lb <= NPLUS and c.vb <= 4 for c in ny)
    stored(qx, qy, P2(STRICT, 1.01), P2(STRICT, 1.01), "first add")

    def add_core(p, r, x1u, y1s, rbias, tag):
        pp = sqr2(p, "8K2")
        ppp = mul2(p, pp, "4K1")
        qq = mul2(x1u, pp, "4K1")
        rr = sqr2(r, rbias)
        x3 = refresh2(norm2(sub2(rr, [ppp, qq, qq], "8K3")))
        t = norm2(sub2(qq, [x3], "4K1"))
        m1, m2 = mul2(t, r, rbias), mul2(y1s, ppp, "4K1")
        y3 = norm2(sub2(m1, [m2], "4K1"))
        return x3, y3, pp, ppp

    # madd-2008-s
    u2, s2 = mul2(qx, ZZ1, "4K1"), mul2(qy, ZZZ1, "4K1")
    p, r = sub2(u2, [X1], "4K1"), sub2(s2, [Y1], "8K2")
    assert krange(X1, u2, "4K1") == (3, 5), krange(X1, u2, "4K1")
    assert krange(Y1, s2, "8K2") == (2, 9), krange(Y1, s2, "8K2")
    x3, y3, pp, ppp = add_core(norm2(p), norm2(r), X1, Y1, "16K2", "madd")
    stored(x3, y3, mul2(ZZ1, pp, "4K1"), mul2(ZZZ1, ppp, "4K1"), "madd")

    # add-2008-s
    X2, Y2, ZZ2, ZZZ2 = P2(STRICT, SX), P2(NPLUS, SY), P2(STRICT, SZ), P2(STRICT, SZ)
    u1, u2, s1, s2 = mul2(X1, ZZ2, "4K1"), mul2(X2, ZZ1, "4K1"), mul2(Y1, ZZZ2, "4K1"), mul2(Y2, ZZZ1, "4K1")
    p, r = sub2(u2, [u1], "4K1"), sub2(s2, [s1], "4K1")
    assert krange(u1, u2, "4K1") == (3, 5) and krange(s1, s2, "4K1") == (3, 5)
    x3, y3, pp, ppp = add_core(norm2(p), norm2(r), u1, s1, "8K2", "add")
    stored(x3, y3, mul2(mul2(ZZ1, ZZ2, "4K1"), pp, "4K1"), mul2(mul2(ZZZ1, ZZZ2, "4K1"), ppp, "4K1"), "add")

    # dbl-2008-s-1 / mdbl-2008-s-1
    for tag, (x, y, zz, zzz) in (("dbl", (X1, Y1, ZZ1, ZZZ1)), ("mdbl", (qx, qy, None, None))):
        u = norm2(add2(y, y))
        v = sqr2(u, "16K2")
        w_ = mul2(u, v, "4K1")
        s = mul2(x, v, "4K1")
        t = sqr2(x, "4K1")
        m = norm2(add2(add2(t, t), t))
        x3 = refresh2(norm2(sub2(sqr2(m, "4K2"), [s, s], "4K2")))
        t = norm2(sub2(s, [x3], "4K1"))
        m1, m2 = mul2(m, t, "8K2"), mul2(w_, y, "8K2")
        y3 = norm2(sub2(m1, [m2], "4K1"))
        # (mdbl: v = (2y)^2 of a possibly negated base reaches 4p on 9 x 29 limbs -- the rare same-point path refreshes it)
        stored(x3, y3, mul2(v, zz, "4K1") if zz else refresh2(v), mul2(w_, zzz, "4K1") if zzz else w_, tag)

    # conversion back: norm, multiply by FROM29, canon handles < 20 p
    for c in (X1[0], Y1[0], ZZ1[0]):
        assert mul(norm(c), V(STRICT, 1)).vb < 2


def check_ntt_tile(w, l, log2p_max, log2p_min, max_log_r=10):
    """The lazy-limb NTT tile (zk_ntt29_kernels.h): decimation-in-time butterflies  t = w tw ; o0 = u + t ; o1 = u - t + 4p
    with one parallel carry step after every third stage.  Every element of the tile is bounded by the same (LB, VB) state;
    elements enter below 2p with strict limbs and must leave, after one more product, below 2p again (they are stored in
    256 bits)."""
    configure(w, l, log2p_max, log2p_min)
    tw = V(STRICT, 1.0, "twiddle")                     # canonical, < p
    worst = 0.0
    for log_r in range(1, max_log_r + 1):
        st = V(STRICT, 2.0, "loaded")                  # < 2p: the caller's canonical value or what an earlier pass stored
        for lg in range(log_r):
            t = mul(st, tw, "w*tw") if lg > 0 else st  # stage 0 has no twiddle: the operand is a fresh load
            assert t.lb <= STRICT and t.vb <= 2.0, t
            o0 = add(st, t, "u+t")
            o1 = sub(st, [t], "4K1", "u-t+4p")
            st = V(max(o0.lb, o1.lb), max(o0.vb, o1.vb), "stage %d" % lg)
            if lg % 3 == 2:
                st = norm(st, "carry")
        out = mul(st, tw, "store")                     # inter-pass twiddle / scale factor / R'
        assert out.vb <= 2.0, out
        worst = max(worst, st.vb)
    assert worst * 2.0 ** log2p_max < 2.0 ** (W * (L - 1) + 32), "top limb overflow"
    return worst


def main():
    for shape in ((29, 9, 255.0, 253.5),):
        vb = check_ntt_tile(*shape)
        print("NTT tile %d x %d: bounds hold for tiles up to 2^10 points (largest value bound %.0f p)" % (shape[1], shape[0], vb))
    check_shape(29, 9, 254.001, 253.5)     # Pasta Fp / Fq (p = 2^254 (1 + 2^-128)), BN254 Fq (2^253.6)
    check_shape(28, 14, 380.8, 380.6)      # BLS12-381 Fq (2^380.7)
    check_shape_ext2(29, 9, 254.001, 253.5)    # BN254 G2
    check_shape_ext2(28, 14, 380.8, 380.6)     # BLS12-381 G2
    print("all F29 bounds hold")
    return 0


if __name__ == "__main__":
    sys.exit(main())
