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


def add(a, b, name="add"):
    r = V(a.lb + b.lb, a.vb + b.vb, name)
    assert r.lb < 1 << 32
    return r


BIAS = {"4K1": (4, 1), "4K2": (4, 2), "16K2": (16, 2), "8K3": (8, 3)}


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
    m1, m2 = mul(r, t), mul(Y1, ppp)
    y3 = norm(sub(m1, [m2], "4K1"))
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
    m1, m2 = mul(r, t), mul(s1, ppp)
    y3 = norm(sub(m1, [m2], "4K1"))
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
        m1, m2 = mul(m, t), mul(w, y)
        y3 = norm(sub(m1, [m2], "4K1"))
        stored(x3, y3, mul(v, zz) if zz else v, mul(w, zzz) if zzz else w)
        print("%s ok: " % tag, x3, y3)

    # ---- conversion back: norm, multiply by FROM29 (strict, < p), canon handles < 20 p
    for c in (X1, Y1, ZZ1):
        assert mul(norm(c), V(STRICT, 1)).vb < 2


def main():
    check_shape(29, 9, 254.001, 253.5)     # Pasta Fp / Fq (p = 2^254 (1 + 2^-128)), BN254 Fq (2^253.6)
    check_shape(28, 14, 380.8, 380.6)      # BLS12-381 Fq (2^380.7)
    print("all F29 bounds hold")
    return 0


if __name__ == "__main__":
    sys.exit(main())
