"""CPU: the F29 field primitives (9 x 29-bit lazy limbs, zk_field29.h) against Python integers, at random AND at the
extreme limb values the bound discipline permits (where a wrong bound would overflow a column or a word), plus the
machine check of the bounds of every curve formula (tools/check_f29_bounds.py)."""
import os
import random
import subprocess
import sys

import pytest

from oracle import pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "emu", "f29_check")
FIELDS = ["PallasFp", "PallasFq", "Bn254Fq", "Bls381Fq"]
SHAPE = {"PallasFp": (29, 9, 8), "PallasFq": (29, 9, 8), "Bn254Fq": (29, 9, 8), "Bls381Fq": (28, 14, 12)}   # W, L, 32-bit words
W = L = MASK = NW = None


def shape(field):
    global W, L, MASK, NW
    W, L, NW = SHAPE[field]
    MASK = (1 << W) - 1


def build():
    src = os.path.join(ROOT, "tests", "emu", "f29_check.cc")
    csrc = os.path.join(ROOT, "contangle-zkcp_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("zk_field29.h", "zk_curve29.h", "zk_params29.h", "zk_field.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + csrc, src, "-o", EXE])


def val(limbs):
    return sum(x << (W * i) for i, x in enumerate(limbs))


def limbs_of(x, strict_top=False):
    out = [(x >> (W * i)) & MASK for i in range(L - 1)]
    out.append(x >> (W * (L - 1)))
    return out


def spread_random(rng, x, lb):
    """random lazy representation of integer x with limbs below the top in [0, lb]"""
    out = limbs_of(x)
    for i in range(L - 1):
        # move a multiple of 2^W from limb i+1 into limb i when possible
        room = (lb - out[i]) >> W
        take = min(room, out[i + 1])
        if take > 0:
            t = rng.randint(0, take)
            out[i] += t << W
            out[i + 1] -= t
    assert val(out) == x and all(0 <= v < 1 << 32 for v in out)
    return out


def run(lines):
    build()
    inp = "\n".join("%s %s %s %s" % (f, op, " ".join(map(str, a)), " ".join(map(str, b))) for f, op, a, b in lines) + "\n"
    out = subprocess.run([EXE], input=inp, stdout=subprocess.PIPE, text=True, check=True).stdout.strip().split("\n")
    assert len(out) == len(lines)
    return [list(map(int, l.split())) for l in out]


def test_bound_checker():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_f29_bounds.py")], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "all F29 bounds hold" in r.stdout, r.stdout


@pytest.mark.parametrize("field", FIELDS)
def test_f29_mul_extremes(field):
    shape(field)
    p = pyref.FIELDS[field][0]
    Rp = 1 << (W * L)
    top = p >> (W * (L - 1))
    rng = random.Random(29)
    cases = []
    nplus = MASK + (1 << (32 - W)) - 1
    wide = MASK + 1 + 2 * (MASK + 1) + nplus                 # q - x3 + BIAS16K2: the widest form a product ever sees
    max_n = [nplus] * (L - 1)                # N+ limbs at their maximum
    max_s = [wide] * (L - 1)
    for k in range(200):
        if k == 0:
            a, b = max_n + [18 * top], max_s + [17 * top]
        elif k == 1:
            a, b = [2 * nplus] * (L - 1) + [16 * top], [2 * nplus] * (L - 1) + [16 * top]   # u = 2y squared in dbl
        elif k == 2:
            a, b = [0] * L, max_s + [1]
        elif k < 100:
            a = spread_random(rng, rng.randrange(18 * p), nplus)
            b = spread_random(rng, rng.randrange(17 * p), wide)
        else:
            a, b = limbs_of(rng.randrange(2 * p)), limbs_of(rng.randrange(2 * p))
        cases.append((a, b))
    res = run([(field, "mul", a, b) for a, b in cases])
    for (a, b), r in zip(cases, res):
        assert all(x <= MASK for x in r[:L - 1])                               # strict limbs
        assert val(r) * Rp % p == val(a) * val(b) % p                          # Montgomery relation
        assert val(r) < val(a) * val(b) // Rp + p + 1                          # value bound


@pytest.mark.parametrize("field", FIELDS)
def test_f29_sub_norm_canon(field):
    shape(field)
    p = pyref.FIELDS[field][0]
    rng = random.Random(7)
    nplus = MASK + (1 << (32 - W)) - 1
    lines, exp = [], []
    for k in range(100):
        a = spread_random(rng, rng.randrange(2 * p), MASK)
        b12 = spread_random(rng, rng.randrange(12 * p), nplus)
        b2 = limbs_of(rng.randrange(2 * p))
        if k == 0:
            a, b12 = [0] * L, limbs_of(12 * p - 1)
        lines.append((field, "sub16k2", a, b12)); exp.append(val(a) - val(b12) + 16 * p)
        lines.append((field, "sub4k1", a, b2)); exp.append(val(a) - val(b2) + 4 * p)
        lines.append((field, "sub3", a, b2)); exp.append(val(a) - 3 * val(b2) + 8 * p)
        lines.append((field, "sub2x", a, b2)); exp.append(val(a) - 2 * val(b2) + 4 * p)
    res = run(lines)
    for r, e in zip(res, exp):
        assert val(r) == e and all(0 <= x < 1 << 32 for x in r)
    # norm / carry / canon
    lines, chk = [], []
    for k in range(100):
        x = rng.randrange(19 * p)
        lazy = spread_random(rng, x, (1 << 32) - 1 - (((1 << (32 - W)) - 1) << W)) if k else [0xFFFFFFF0] * (L - 1) + [1]   # carry-in must not wrap a word
        x = val(lazy)
        if x >= 20 * p:
            continue
        lines.append((field, "norm", lazy, [0] * L)); chk.append(("norm", x))
        lines.append((field, "carry", lazy, [0] * L)); chk.append(("carry", x))
        lines.append((field, "canon", lazy, [0] * L)); chk.append(("canon", x))
    res = run(lines)
    for r, (op, x) in zip(res, chk):
        if op == "norm":
            assert val(r) == x and all(v <= MASK + (1 << (32 - W)) - 1 for v in r[:L - 1])
        elif op == "carry":
            assert val(r) == x and all(v <= MASK for v in r[:L - 1])
        else:
            assert val(r) == x % p and all(v <= MASK for v in r[:L - 1])


@pytest.mark.parametrize("field", FIELDS)
def test_f29_conversion_and_zero_filter(field):
    shape(field)
    p = pyref.FIELDS[field][0]
    rng = random.Random(3)
    R, Rp = 1 << (32 * NW), 1 << (W * L)
    lines, xs = [], []
    for k in range(50):
        x = [0, 1, p - 1][k] if k < 3 else rng.randrange(p)
        std = x * R % p
        words = [(std >> (32 * i)) & 0xFFFFFFFF for i in range(NW)] + [0] * (L - NW)
        lines.append((field, "fromstd", words, [0] * L)); xs.append(x)
    res = run(lines)
    back = []
    for r, x in zip(res, xs):
        assert val(r) % p == x * Rp % p and val(r) < 2 * p and all(v <= MASK for v in r[:L - 1])
        if x == 0:
            assert val(r) == 0
        # lazy variant of the same value goes back to the canonical standard form
        lazy = spread_random(rng, val(r) + rng.randrange(10) * p, (1 << 31))
        back.append((field, "tostd", lazy, [0] * L))
    res = run(back)
    for r, x in zip(res, xs):
        got = sum(w << (32 * i) for i, w in enumerate(r[:NW]))
        assert got == x * R % p
    # zero filter: multiples of p in range are accepted exactly, everything else rejected
    lines, exp = [], []
    for k in range(3, 18):
        for delta in (0, 1, p // 3):
            x = k * p + delta
            lazy = spread_random(rng, x, 1 << 31)
            lines.append((field, "filter", lazy, [3, 17] + [0] * (L - 2))); exp.append(delta == 0)
    for _ in range(200):
        x = rng.randrange(3 * p, 18 * p)
        lines.append((field, "filter", spread_random(rng, x, 1 << 31), [3, 17] + [0] * (L - 2))); exp.append(x % p == 0)
    res = run(lines)
    for r, e in zip(res, exp):
        assert bool(r[2]) == e


@pytest.mark.parametrize("field", ["Bn254Fq", "Bls381Fq"])
def test_f29_fq2_mul_sqr_extremes(field):
    """Fe29x2 (the G2 coordinates): product with a negated operand and one reduction per component, complex square,
    refresh, zero test -- at the limb / value bounds tools/check_f29_bounds.py allows at their call sites"""
    shape(field)
    p = pyref.FIELDS[field][0]
    Rp = 1 << (W * L)
    rng = random.Random(58)
    nplus = MASK + (1 << (32 - W)) - 1
    top = p >> (W * (L - 1))

    def lazy(vb, lb):
        return spread_random(rng, rng.randrange(int(vb * p)), lb)

    lines, exp = [], []
    for op, bvb, blb, kmul in (("x2mul4k1", 2.9, MASK, 4), ("x2mul8k2", 6.9, nplus, 8), ("x2mul16k2", 14.9, nplus, 16)):
        for k in range(60):
            if k == 0:       # every limb at its maximum
                a0 = a1 = b0 = [nplus] * (L - 1) + [10 * top]
                b1 = [blb] * (L - 1) + [int((bvb - 1) * top)]
            else:
                a0, a1, b0, b1 = lazy(10, nplus), lazy(10, nplus), lazy(10, nplus), lazy(bvb, blb)
            lines.append((field, op, a0 + a1, b0 + b1))
            exp.append(("mul", a0, a1, b0, b1, kmul))
    for op, avb, kb in (("x2sqr8k2", 6.9, 8), ("x2sqr16k2", 14.9, 16)):
        for k in range(60):
            a0, a1 = ([nplus] * (L - 1) + [int((avb - 1) * top)],) * 2 if k == 0 else (lazy(avb, nplus), lazy(avb, nplus))
            lines.append((field, op, a0 + a1, [0] * (2 * L)))
            exp.append(("sqr", a0, a1, None, None, kb))
    for k in range(40):
        a0, a1 = lazy(19, nplus), lazy(19, nplus)
        lines.append((field, "x2refresh", a0 + a1, [0] * (2 * L)))
        exp.append(("refresh", a0, a1, None, None, 0))
    res = run(lines)
    for r, (kind, a0, a1, b0, b1, kb) in zip(res, exp):
        c0, c1 = r[:L], r[L:]
        assert all(x <= MASK for x in c0[:L - 1] + c1[:L - 1])            # strict limbs
        A0, A1 = val(a0), val(a1)
        if kind == "mul":
            B0, B1 = val(b0), val(b1)
            assert val(c0) * Rp % p == (A0 * B0 - A1 * B1) % p and val(c1) * Rp % p == (A0 * B1 + A1 * B0) % p
            assert val(c0) < (A0 * B0 + A1 * (kb * p - B1)) // Rp + p + 1 and val(c1) < (A0 * B1 + A1 * B0) // Rp + p + 1
        elif kind == "sqr":
            assert val(c0) * Rp % p == (A0 * A0 - A1 * A1) % p and val(c1) * Rp % p == 2 * A0 * A1 % p
            assert val(c0) < (A0 + A1) * (A0 - A1 + kb * p) // Rp + p + 1
        else:
            assert val(c0) % p == A0 % p and val(c1) % p == A1 % p and val(c0) < A0 // (1 << 6) + p + 1 and val(c0) < 2 * p
    # zero test: both components must be multiples of p inside [kmin p, kmax p]
    lines, exp = [], []
    for k0 in (2, 5, 9):
        for k1 in (2, 9):
            for d0, d1 in ((0, 0), (1, 0), (0, 1), (p // 5, 0)):
                x0, x1 = spread_random(rng, k0 * p + d0, 1 << 31), spread_random(rng, k1 * p + d1, 1 << 31)
                lines.append((field, "x2iszero", x0 + x1, [2, 9] + [0] * (2 * L - 2)))
                exp.append(d0 == 0 and d1 == 0)
    res = run(lines)
    for r, e in zip(res, exp):
        assert bool(r[0]) == e


@pytest.mark.parametrize("field", FIELDS)
def test_f29_sqr_and_mulacc(field):
    """the dedicated square (doubled operand, symmetric half of the products) and the two-product multiply with one
    reduction, at the widest operands their call sites produce"""
    shape(field)
    p = pyref.FIELDS[field][0]
    Rp = 1 << (W * L)
    top = p >> (W * (L - 1))
    rng = random.Random(31)
    nplus = MASK + (1 << (32 - W)) - 1
    wide = MASK + 1 + 2 * (MASK + 1) + nplus
    cases = []
    for k in range(120):
        if k == 0:
            a = [2 * nplus] * (L - 1) + [16 * top]           # u = 2y in dbl: the widest operand a square sees
        elif k == 1:
            a = [nplus] * (L - 1) + [18 * top]
        else:
            a = spread_random(rng, rng.randrange(16 * p), 2 * nplus)
        cases.append(a)
    res = run([(field, "sqr", a, [0] * L) for a in cases])
    for a, r in zip(cases, res):
        assert all(x <= MASK for x in r[:L - 1])
        assert val(r) * Rp % p == val(a) * val(a) % p and val(r) < val(a) * val(a) // Rp + p + 1
    cases = []
    for k in range(120):
        if k == 0:
            a, b = [nplus] * (L - 1) + [18 * top], [wide] * (L - 1) + [17 * top]     # r and t = q - x3 + 16p
        else:
            a, b = spread_random(rng, rng.randrange(18 * p), nplus), spread_random(rng, rng.randrange(17 * p), wide)
        cases.append((a, b))
    res = run([(field, "mulacc", a, b) for a, b in cases])
    for (a, b), r in zip(cases, res):
        c, d = [x >> 1 for x in b], [x >> 1 for x in a]
        tot = val(a) * val(b) + val(c) * val(d)
        assert all(x <= MASK for x in r[:L - 1])
        assert val(r) * Rp % p == tot % p and val(r) < tot // Rp + p + 1
