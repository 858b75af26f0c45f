// "F29": a base field as L unsaturated W-bit limbs in u32 words -- 9 x 29 bits (R' = 2^261) for the 254/255-bit fields
// (Pasta Fp/Fq, BN254 Fq), 14 x 28 bits (R' = 2^392) for BLS12-381 Fq -- Montgomery radix R' = 2^(W L), with LAZY reduction.  This is the arithmetic of the MSM bucket kernels on gfx950, where every VALU
// instruction costs the same issue slot:
//   * a column of partial products never overflows a 64-bit accumulator, so a product is ONE v_mad_u64_u32 (the
//     saturated 32-bit form needs a MAD and a carry add), ~225 instead of ~290 instructions per multiplication;
//   * additions are 9 independent limb adds, subtractions 9 limb subtract-and-bias pairs; no carry chains, no conditional
//     subtraction of p.  Values are only ever reduced by the next Montgomery product.
// The price is a bound discipline, stated here and machine-checked by tools/check_f29_bounds.py for every formula of
// zk_curve29.h:
//   limb bound  LB : every limb below the top one is <= LB            (u32 words; LB < 2^32)
//   value bound VB : the integer  sum v[i] 2^(29 i)  is  < VB * p
//   * fe29_mul(a, b):      requires 9 * LB(a) * LB(b) + 9 * 2^58 + 2^36 < 2^64;  result limbs < 2^29 ("strict"), value < VB(a) VB(b) p / 2^7 + p
//   * fe29_add(a, b):      LB = LB(a) + LB(b), VB = VB(a) + VB(b)
//   * fe29_sub<BIAS>(a,b): a - b + BIAS limb-wise; requires LB(b) <= k (2^29 - 1) [the k of the bias] and VB(b) + 2 <= its multiple of p;
//                          LB = LB(a) + max limb of BIAS, VB = VB(a) + multiple
//   * fe29_norm(a):        one parallel carry step: LB <= 2^29 + 6 ("N+"), value unchanged
// Exact comparisons (is this value == 0 mod p?) never look at a lazy value directly: fe29_zero_filter() rejects all but
// ~2^-25 of the non-zero cases from the low limb alone, and fe29_is_kp() decides the survivors on the fully carried integer.
//
// Data never changes Montgomery domain behind the caller's back: bases are converted once when they are uploaded
// (x R  ->  x R'), results are converted back on the host (zk_msm.inl).
#pragma once
#include "zk_field.h"
#include "zk_params29.h"

namespace zk {

template <class P>
struct alignas(4) Fe29 {
    uint32_t v[F29<P>::L];
};
template <class P>
constexpr uint32_t f29_mask() {
    return (1u << F29<P>::W) - 1;
}

template <class P>
ZK_HD void fe29_zero(Fe29<P>& r) {
    ZK_UNROLL
    for (int i = 0; i < F29<P>::L; i++) r.v[i] = 0;
}
template <class P>
ZK_HD void fe29_one(Fe29<P>& r) {
    ZK_UNROLL
    for (int i = 0; i < F29<P>::L; i++) r.v[i] = F29<P>::ONE[i];
}
// all limbs zero: only the literal zero written by fe29_zero / a converted (0,0) identity satisfies this
template <class P>
ZK_HD bool fe29_is_literal_zero(const Fe29<P>& a) {
    uint32_t o = 0;
    ZK_UNROLL
    for (int i = 0; i < F29<P>::L; i++) o |= a.v[i];
    return o == 0;
}
template <class P>
ZK_HD void fe29_cmov(Fe29<P>& r, const Fe29<P>& a, bool sel) {
    ZK_UNROLL
    for (int i = 0; i < F29<P>::L; i++) r.v[i] = sel ? a.v[i] : r.v[i];
}

// Montgomery product, R' = 2^261, carry-free columns (finely integrated product scanning).
// TWO: r = (a b + c d) / R' -- both products summed in the same columns, one reduction (the Fq2 product below); requires
// L (LB(a) LB(b) + LB(c) LB(d)) + L 2^(2W) + 2^36 < 2^64.
template <class P, bool TWO>
ZK_HD void fe29_mul_impl(Fe29<P>& r, const Fe29<P>& a, const Fe29<P>& b, const Fe29<P>& c, const Fe29<P>& d) {
    using K = F29<P>;
    constexpr int L = K::L, W = K::W;
    constexpr uint32_t MASK = (1u << W) - 1;
    uint64_t acc = 0;
    uint32_t m[L];
    uint32_t out[L];
    ZK_UNROLL
    for (int k = 0; k < 2 * L - 1; k++) {
        if constexpr (L <= 9) {
            // two independent accumulation chains per column (products / reduction terms): the MADs of one chain are
            // serially dependent, and at 3 waves per SIMD a second chain fills issue slots between them (measured:
            // accumulate -2% on the 9-limb fields, +2% on the 14-limb one, which therefore keeps the single chain)
            uint64_t red = 0;
            ZK_UNROLL
            for (int i = 0; i < L; i++) {
                const int j = k - i;
                // m_i is known for i < k (first half) and for every i once k >= L
                if (j >= 1 && j < L && (i < k) && K::P[j] != 0) red += (uint64_t)m[i] * K::P[j];
            }
            ZK_UNROLL
            for (int i = 0; i < L; i++) {
                const int j = k - i;
                if (j >= 0 && j < L) {
                    acc += (uint64_t)a.v[i] * b.v[j];
                    if (TWO) red += (uint64_t)c.v[i] * d.v[j];
                }
            }
            acc += red;
        } else {
            ZK_UNROLL
            for (int i = 0; i < L; i++) {
                const int j = k - i;
                if (j >= 0 && j < L) {
                    acc += (uint64_t)a.v[i] * b.v[j];
                    if (TWO) acc += (uint64_t)c.v[i] * d.v[j];
                }
            }
            ZK_UNROLL
            for (int i = 0; i < L; i++) {
                const int j = k - i;
                if (j >= 1 && j < L && (i < k) && K::P[j] != 0) acc += (uint64_t)m[i] * K::P[j];
            }
        }
        if (k < L) {
            m[k] = ((uint32_t)acc * K::INV) & MASK;
            acc += (uint64_t)m[k] * K::P[0];
            acc >>= W;
        } else {
            out[k - L] = (uint32_t)acc & MASK;
            acc >>= W;
        }
    }
    out[L - 1] = (uint32_t)acc;
    ZK_UNROLL
    for (int i = 0; i < L; i++) r.v[i] = out[i];
}
template <class P>
ZK_HD void fe29_mul(Fe29<P>& r, const Fe29<P>& a, const Fe29<P>& b) {
    fe29_mul_impl<P, false>(r, a, b, a, b);
}
template <class P>
ZK_HD void fe29_mulacc(Fe29<P>& r, const Fe29<P>& a, const Fe29<P>& b, const Fe29<P>& c, const Fe29<P>& d) {
    fe29_mul_impl<P, true>(r, a, b, c, d);
}
// (a dedicated square -- symmetric half of the partial products against the doubled operand, L (L + 1) / 2 MADs -- was
// measured SLOWER in the bucket kernel: +2% on 9 limbs, +8% on 14; the kernel is bound by issue efficiency at 3 waves
// per SIMD and by its dependent chains, not by the MAD count)
template <class P>
ZK_HD void fe29_sqr(Fe29<P>& r, const Fe29<P>& a) {
    fe29_mul(r, a, a);
}

template <class P>
ZK_HD void fe29_add(Fe29<P>& r, const Fe29<P>& a, const Fe29<P>& b) {
    ZK_UNROLL
    for (int i = 0; i < F29<P>::L; i++) r.v[i] = a.v[i] + b.v[i];
}
// r = a - b + BIAS (BIAS = one of the F29<P>::BIAS* tables)
template <class P>
ZK_HD void fe29_sub(Fe29<P>& r, const Fe29<P>& a, const Fe29<P>& b, const uint32_t (&bias)[F29<P>::L]) {
    ZK_UNROLL
    for (int i = 0; i < F29<P>::L; i++) r.v[i] = a.v[i] + bias[i] - b.v[i];
}
// r = a - b - c - c + BIAS8K3
template <class P>
ZK_HD void fe29_sub3(Fe29<P>& r, const Fe29<P>& a, const Fe29<P>& b, const Fe29<P>& c) {
    ZK_UNROLL
    for (int i = 0; i < F29<P>::L; i++) r.v[i] = a.v[i] + F29<P>::BIAS8K3[i] - b.v[i] - c.v[i] - c.v[i];
}
// r = a - c - c + BIAS4K2
template <class P>
ZK_HD void fe29_sub2x(Fe29<P>& r, const Fe29<P>& a, const Fe29<P>& c) {
    ZK_UNROLL
    for (int i = 0; i < F29<P>::L; i++) r.v[i] = a.v[i] + F29<P>::BIAS4K2[i] - c.v[i] - c.v[i];
}
// one parallel carry step: limbs below the top become <= 2^W - 1 + (2^(32-W) - 1)
template <class P>
ZK_HD void fe29_norm(Fe29<P>& r, const Fe29<P>& a) {
    constexpr int L = F29<P>::L, W = F29<P>::W;
    constexpr uint32_t MASK = (1u << W) - 1;
    uint32_t c[L - 1];
    ZK_UNROLL
    for (int i = 0; i < L - 1; i++) c[i] = a.v[i] >> W;
    r.v[L - 1] = a.v[L - 1] + c[L - 2];
    ZK_UNROLL
    for (int i = L - 2; i >= 1; i--) r.v[i] = (a.v[i] & MASK) + c[i - 1];
    r.v[0] = a.v[0] & MASK;
}
// full serial carry: every limb below the top strictly < 2^W (unique digits of the integer)
template <class P>
ZK_HD void fe29_carry(Fe29<P>& r, const Fe29<P>& a) {
    constexpr int L = F29<P>::L, W = F29<P>::W;
    constexpr uint32_t MASK = (1u << W) - 1;
    uint32_t c = 0;
    ZK_UNROLL
    for (int i = 0; i < L - 1; i++) {
        const uint32_t t = a.v[i] + c;
        r.v[i] = t & MASK;
        c = t >> W;
    }
    r.v[L - 1] = a.v[L - 1] + c;
}
// cheap necessary condition for "integer(a) == k p for some kmin <= k <= kmax": the low 29 bits of the integer decide k
template <class P>
ZK_HD bool fe29_zero_filter(const Fe29<P>& a, uint32_t kmin, uint32_t kmax, uint32_t& k) {
    k = ((a.v[0] & f29_mask<P>()) * F29<P>::P0INV) & f29_mask<P>();
    return k - kmin <= kmax - kmin;
}
// exact: integer(a) == k p   (k < 20)
template <class P>
ZK_HD bool fe29_is_kp(const Fe29<P>& a, uint32_t k) {
    Fe29<P> t;
    fe29_carry(t, a);
    uint32_t o = 0;
    for (int i = 0; i < F29<P>::L; i++) o |= t.v[i] ^ F29<P>::KP[k][i];
    return o == 0;
}
// canonical representative in [0, p): strict limbs.  Conversion / rare paths only; value must be < 20 p.
// Branch-free: t -= 16p, 8p, 4p, 2p, p in turn, each kept only if it did not borrow out of the top limb (compile-time rows of
// KP, no data-dependent control flow).  Round 2's form -- "for k = 19 .. 1: lexicographic compare with KP[k], early exits,
// subtract and break" -- is correct C++ (host and emulator agree with Python) but, inlined behind the LDS scan of
// msm_axis_weighted_kernel with one active lane, gfx950 code generation returned a difference whose middle limbs were
// 2^29 - 1 for BN254's Fq (the borrow chain ran against stale operands: profiles/r03_a_bn254_device_partials.txt shows the
// product stage equal and this stage differing on the device only).  tests: test_msm_device_side_partial_conversion.
template <class P>
ZK_HD void fe29_canon(Fe29<P>& r, const Fe29<P>& a) {
    constexpr int L = F29<P>::L, W = F29<P>::W;
    Fe29<P> t;
    fe29_carry(t, a);
    ZK_UNROLL
    for (int s = 4; s >= 0; s--) {
        const int k = 1 << s;
        uint32_t d[L];
        int32_t br = 0;
        ZK_UNROLL
        for (int i = 0; i < L - 1; i++) {
            const int32_t x = (int32_t)t.v[i] - (int32_t)F29<P>::KP[k][i] - br;
            br = (int32_t)((uint32_t)x >> 31);
            d[i] = (uint32_t)(x + (br << W));
        }
        const int64_t top = (int64_t)t.v[L - 1] - (int64_t)F29<P>::KP[k][L - 1] - br;
        d[L - 1] = (uint32_t)top;
        const uint32_t keep = top < 0 ? 0u : 0xffffffffu;      // t >= k p: take the difference
        ZK_UNROLL
        for (int i = 0; i < L; i++) t.v[i] = (d[i] & keep) | (t.v[i] & ~keep);
    }
    r = t;
}

// ---- conversions between the caller's form (8 x u32, x R mod p, R = 2^256) and F29 (x R' mod p) ----
// repack the caller's 32-bit words into W-bit limbs (no arithmetic)
template <class P>
ZK_HD void fe29_unpack(Fe29<P>& r, const Fe<P>& a) {
    constexpr int L = F29<P>::L, W = F29<P>::W, N = P::N;
    ZK_UNROLL
    for (int i = 0; i < L; i++) {
        const int bit = W * i, w = bit >> 5, off = bit & 31;
        uint64_t x = 0;
        if (w < N) x = a.v[w];
        if (w + 1 < N) x |= (uint64_t)a.v[w + 1] << 32;
        r.v[i] = (uint32_t)(x >> off) & (i == L - 1 ? 0xffffffffu : (1u << W) - 1);
    }
}
// strict canonical limbs (< p) -> the caller's 32-bit words
template <class P>
ZK_HD void fe29_pack(Fe<P>& r, const Fe29<P>& a) {
    constexpr int L = F29<P>::L, W = F29<P>::W, N = P::N;
    ZK_UNROLL
    for (int w = 0; w < N; w++) {
        uint64_t x = 0;
        ZK_UNROLL
        for (int i = 0; i < L; i++) {
            const int lo = W * i - 32 * w;  // position of limb i relative to word w
            if (lo > -W && lo < 32) x |= lo >= 0 ? (uint64_t)a.v[i] << lo : (uint64_t)a.v[i] >> (-lo);
        }
        r.v[w] = (uint32_t)x;
    }
}
// caller's Montgomery form -> F29 Montgomery form (strict limbs, value < 2p); 0 -> literal 0
template <class P>
ZK_HD void fe29_from_std(Fe29<P>& r, const Fe<P>& a) {
    Fe29<P> t, c;
    fe29_unpack(t, a);
    ZK_UNROLL
    for (int i = 0; i < F29<P>::L; i++) c.v[i] = F29<P>::TO29[i];
    fe29_mul(r, t, c);
}
// F29 (any lazy value < 20p after one multiplication: the product below brings it under 2p) -> caller's canonical form
template <class P>
ZK_HD void fe29_to_std(Fe<P>& r, const Fe29<P>& a) {
    Fe29<P> t, c;
    ZK_UNROLL
    for (int i = 0; i < F29<P>::L; i++) c.v[i] = F29<P>::FROM29[i];
    fe29_norm(t, a);
    fe29_mul(t, t, c);
    fe29_canon(t, t);
    fe29_pack(r, t);
}

// ------------------------------------------------------------------------------------------------------------------
// Fq2 = Fq[u] / (u^2 + 1) over the lazy limbs (the G2 twists of BN254 and BLS12-381): a pair of Fe29.  The product is
//   c0 = (a0 b0 + a1 (k p - b1)) / R'      c1 = (a0 b1 + a1 b0) / R'
// two double products with ONE reduction each (2 x (2 L^2 + reduction) MADs instead of the 3 full products of Karatsuba
// plus its biased recombination), and its outputs are strict limbs with a value of a few p at most -- which is what lets
// the G2 formulas of zk_curve29.h keep the bias constants of the G1 ones.  `negbias` is the BIAS table used to negate b.c1:
// it must dominate b.c1 (see fe29_sub).  The square is the complex one, (a0 + a1)(a0 - a1 + k p) and 2 a0 a1.
// ------------------------------------------------------------------------------------------------------------------
template <class P>
struct alignas(4) Fe29x2 {
    Fe29<P> c0, c1;
};
template <class P>
ZK_HD void fe29_zero(Fe29x2<P>& r) {
    fe29_zero(r.c0);
    fe29_zero(r.c1);
}
template <class P>
ZK_HD void fe29_one(Fe29x2<P>& r) {
    fe29_one(r.c0);
    fe29_zero(r.c1);
}
template <class P>
ZK_HD bool fe29_is_literal_zero(const Fe29x2<P>& a) {
    return fe29_is_literal_zero(a.c0) && fe29_is_literal_zero(a.c1);
}
template <class P>
ZK_HD void fe29_cmov(Fe29x2<P>& r, const Fe29x2<P>& a, bool sel) {
    fe29_cmov(r.c0, a.c0, sel);
    fe29_cmov(r.c1, a.c1, sel);
}
template <class P>
ZK_HD void fe29_add(Fe29x2<P>& r, const Fe29x2<P>& a, const Fe29x2<P>& b) {
    fe29_add(r.c0, a.c0, b.c0);
    fe29_add(r.c1, a.c1, b.c1);
}
template <class P>
ZK_HD void fe29_sub(Fe29x2<P>& r, const Fe29x2<P>& a, const Fe29x2<P>& b, const uint32_t (&bias)[F29<P>::L]) {
    fe29_sub(r.c0, a.c0, b.c0, bias);
    fe29_sub(r.c1, a.c1, b.c1, bias);
}
template <class P>
ZK_HD void fe29_sub3(Fe29x2<P>& r, const Fe29x2<P>& a, const Fe29x2<P>& b, const Fe29x2<P>& c) {
    fe29_sub3(r.c0, a.c0, b.c0, c.c0);
    fe29_sub3(r.c1, a.c1, b.c1, c.c1);
}
template <class P>
ZK_HD void fe29_sub2x(Fe29x2<P>& r, const Fe29x2<P>& a, const Fe29x2<P>& c) {
    fe29_sub2x(r.c0, a.c0, c.c0);
    fe29_sub2x(r.c1, a.c1, c.c1);
}
template <class P>
ZK_HD void fe29_norm(Fe29x2<P>& r, const Fe29x2<P>& a) {
    fe29_norm(r.c0, a.c0);
    fe29_norm(r.c1, a.c1);
}
template <class P>
ZK_HD void fe29_mul(Fe29x2<P>& r, const Fe29x2<P>& a, const Fe29x2<P>& b, const uint32_t (&negbias)[F29<P>::L]) {
    Fe29<P> z, nb1, t0, t1;
    fe29_zero(z);
    fe29_sub(nb1, z, b.c1, negbias);
    fe29_norm(nb1, nb1);
    fe29_mulacc(t0, a.c0, b.c0, a.c1, nb1);
    fe29_mulacc(t1, a.c0, b.c1, a.c1, b.c0);
    r.c0 = t0;
    r.c1 = t1;
}
template <class P>
ZK_HD void fe29_sqr(Fe29x2<P>& r, const Fe29x2<P>& a, const uint32_t (&dbias)[F29<P>::L]) {
    Fe29<P> s, d, e, t0, t1;
    fe29_add(s, a.c0, a.c1);
    fe29_sub(d, a.c0, a.c1, dbias);
    fe29_norm(d, d);
    fe29_add(e, a.c0, a.c0);
    fe29_mul(t0, s, d);
    fe29_mul(t1, e, a.c1);
    r.c0 = t0;
    r.c1 = t1;
}
// value-preserving contraction: a * (R' mod p) / R' = a, strict limbs, value < VB(a) / 2^7 + 1 times p
template <class P>
ZK_HD void fe29_refresh(Fe29x2<P>& r, const Fe29x2<P>& a) {
    Fe29<P> one;
    fe29_one(one);
    fe29_mul(r.c0, a.c0, one);
    fe29_mul(r.c1, a.c1, one);
}
// is the Fq2 value zero, given that both components are integers in [kmin p, kmax p]?
template <class P>
ZK_HD bool fe29_is_zero_mod_p(const Fe29x2<P>& a, uint32_t kmin, uint32_t kmax) {
    uint32_t k0, k1;
    if (!fe29_zero_filter(a.c0, kmin, kmax, k0)) return false;
    if (!fe29_zero_filter(a.c1, kmin, kmax, k1)) return false;
    return fe29_is_kp(a.c0, k0) && fe29_is_kp(a.c1, k1);
}
template <class P>
ZK_HD void fe29_from_std(Fe29x2<P>& r, const Fe2<P>& a) {
    fe29_from_std(r.c0, a.c0);
    fe29_from_std(r.c1, a.c1);
}
template <class P>
ZK_HD void fe29_to_std(Fe2<P>& r, const Fe29x2<P>& a) {
    fe29_to_std(r.c0, a.c0);
    fe29_to_std(r.c1, a.c1);
}

}  // namespace zk
