// XYZZ point arithmetic over the F29 view (zk_field29.h) of Pallas, Vesta and BN254 G1: the bucket arithmetic of the MSM
// kernels.  `C29<C>` is the curve C seen through 9 x 29-bit lazy limbs; the MSM kernels are instantiated with it and pick
// up the overloads below instead of the generic formulas of zk_curve.h.
//
// Stored-point invariant (checked for every formula by tools/check_f29_bounds.py; "N+" = limbs <= 2^29 + 6):
//     X: N+, value < 12p      Y: N+, value < 8p      ZZ, ZZZ: strict limbs, value < 2p      identity: ZZ = literal 0
// Affine inputs come from fe29_from_std (strict limbs, value < 2p) with y possibly negated (N+, value < 4p).
// The formulas are the same EFD madd-2008-s / add-2008-s / dbl-2008-s-1 / mdbl-2008-s-1 as zk_curve.h; what differs is
// where a parallel carry (fe29_norm) is inserted and which bias each subtraction uses.
#pragma once
#include "zk_curve.h"
#include "zk_field29.h"

namespace zk {

template <class C>
struct C29 {
    using Base = C;
    using Fq = typename C::Fq;
    using Fr = typename C::Fr;
    static constexpr int EXT = 29;
};

template <class C>
ZK_HD bool aff_is_inf(const Affine<C29<C>>& p) {
    return fe29_is_literal_zero(p.x) && fe29_is_literal_zero(p.y);
}
template <class C>
ZK_HD void xyzz_set_inf(XYZZ<C29<C>>& p) {
    fe29_zero(p.x);
    fe29_zero(p.y);
    fe29_zero(p.zz);
    fe29_zero(p.zzz);
}
template <class C>
ZK_HD bool xyzz_is_inf(const XYZZ<C29<C>>& p) {
    return fe29_is_literal_zero(p.zz);
}
// y -> -y (as 4p - y, one carry step): y strict, value < 2p  ->  N+, value < 4p
template <class C>
ZK_HD void aff_neg_if(Affine<C29<C>>& p, bool neg) {
    using F = typename C::Fq;
    Fe29<F> z, ny;
    fe29_zero(z);
    fe29_sub(ny, z, p.y, F29<F>::BIAS4K1);
    fe29_norm(ny, ny);
    // the identity (0, 0) must stay literal zero: 4p - 0 is not
    fe29_cmov(p.y, ny, neg && !fe29_is_literal_zero(p.x));
}

// r = 2 q for affine q (strict limbs, value < 4p on y): mdbl-2008-s-1
template <class C>
ZK_HD void xyzz_dbl_affine(XYZZ<C29<C>>& r, const Affine<C29<C>>& q) {
    using F = typename C::Fq;
    using K = F29<F>;
    Fe29<F> u, v, w, s, m, t, x3, m1, m2;
    fe29_add(u, q.y, q.y);   // LB 2^30+12, VB 8
    fe29_sqr(v, u);          // strict, < 1.5p
    fe29_mul(w, u, v);       // < 1.1p
    fe29_mul(s, q.x, v);     // < 1.1p
    fe29_sqr(t, q.x);        // < 1.1p
    fe29_add(m, t, t);
    fe29_add(m, m, t);       // 3 x^2: LB 3 * 2^29, VB 3.3
    fe29_norm(m, m);         // N+
    fe29_sqr(x3, m);         // < 1.1p
    fe29_sub2x(x3, x3, s);   // - 2s + 4p: VB 5.1
    fe29_norm(x3, x3);
    fe29_sub(t, s, x3, K::BIAS16K2);  // VB 17.1, LB < 2^31.2
    fe29_mul(m1, m, t);
    fe29_mul(m2, w, q.y);
    fe29_sub(t, m1, m2, K::BIAS4K1);
    fe29_norm(r.y, t);
    r.x = x3;
    r.zz = v;
    r.zzz = w;
}

// p = 2p: dbl-2008-s-1
template <class C>
ZK_HD void xyzz_dbl(XYZZ<C29<C>>& p) {
    using F = typename C::Fq;
    using K = F29<F>;
    if (xyzz_is_inf(p)) return;
    Fe29<F> u, v, w, s, m, t, x3, m1, m2;
    fe29_add(u, p.y, p.y);   // LB 2^30+12, VB 16
    fe29_sqr(v, u);          // < 3p
    fe29_mul(w, u, v);       // < 1.4p
    fe29_mul(s, p.x, v);     // < 1.3p
    fe29_sqr(t, p.x);        // < 2.2p
    fe29_add(m, t, t);
    fe29_add(m, m, t);       // LB 3 * 2^29, VB 6.6
    fe29_norm(m, m);
    fe29_sqr(x3, m);         // < 1.4p
    fe29_sub2x(x3, x3, s);   // VB 5.4
    fe29_norm(x3, x3);
    fe29_sub(t, s, x3, K::BIAS16K2);
    fe29_mul(m1, m, t);      // < 1.9p
    fe29_mul(m2, w, p.y);    // < 1.1p
    fe29_sub(t, m1, m2, K::BIAS4K1);
    fe29_norm(p.y, t);       // VB 5.9
    p.x = x3;
    fe29_mul(p.zz, v, p.zz);
    fe29_mul(p.zzz, w, p.zzz);
}

// acc += q (affine): madd-2008-s
template <class C>
ZK_HD void xyzz_add_mixed(XYZZ<C29<C>>& acc, const Affine<C29<C>>& q) {
    using F = typename C::Fq;
    using K = F29<F>;
    if (aff_is_inf(q)) return;
    if (xyzz_is_inf(acc)) {
        acc.x = q.x;
        acc.y = q.y;
        fe29_one(acc.zz);
        fe29_one(acc.zzz);
        return;
    }
    Fe29<F> u2, s2, p, r, pp, ppp, qq, rr, t, m1, m2;
    fe29_mul(u2, q.x, acc.zz);
    fe29_mul(s2, q.y, acc.zzz);
    fe29_sub(p, u2, acc.x, K::BIAS16K2);   // integer in (4p, 18p)
    fe29_sub(r, s2, acc.y, K::BIAS16K2);   // integer in (8p, 18p)
    uint32_t k;
    if (fe29_zero_filter(p, 5, 17, k) && fe29_is_kp(p, k)) {
        // same x: either the same point (double it) or its negative (sum is the identity)
        uint32_t kr;
        if (fe29_zero_filter(r, 9, 17, kr) && fe29_is_kp(r, kr)) {
            xyzz_dbl_affine(acc, q);
        } else {
            xyzz_set_inf(acc);
        }
        return;
    }
    fe29_norm(p, p);
    fe29_norm(r, r);
    fe29_sqr(pp, p);           // < 3.6p
    fe29_mul(ppp, p, pp);      // < 1.6p
    fe29_mul(qq, acc.x, pp);   // < 1.4p
    fe29_sqr(rr, r);           // < 3.6p
    fe29_sub3(t, rr, ppp, qq); // rr - ppp - 2 qq + 8p: VB 11.6
    fe29_norm(acc.x, t);
    fe29_sub(t, qq, acc.x, K::BIAS16K2);   // VB 17.4, LB < 2^31.2
    fe29_mul(m1, r, t);        // < 3.5p
    fe29_mul(m2, acc.y, ppp);  // < 1.1p
    fe29_sub(t, m1, m2, K::BIAS4K1);
    fe29_norm(acc.y, t);       // VB 7.5
    fe29_mul(acc.zz, acc.zz, pp);
    fe29_mul(acc.zzz, acc.zzz, ppp);
}

// acc += q (both XYZZ) unless acc == q (then acc is untouched and true is returned: the caller doubles): add-2008-s
template <class C>
ZK_HD bool xyzz_add_nodbl(XYZZ<C29<C>>& acc, const XYZZ<C29<C>>& q) {
    using F = typename C::Fq;
    using K = F29<F>;
    if (xyzz_is_inf(q)) return false;
    if (xyzz_is_inf(acc)) {
        acc = q;
        return false;
    }
    Fe29<F> u1, u2, s1, s2, p, r, pp, ppp, qq, rr, t, m1, m2;
    fe29_mul(u1, acc.x, q.zz);
    fe29_mul(u2, q.x, acc.zz);
    fe29_mul(s1, acc.y, q.zzz);
    fe29_mul(s2, q.y, acc.zzz);
    fe29_sub(p, u2, u1, K::BIAS4K1);   // integer in (2p, 6p)
    fe29_sub(r, s2, s1, K::BIAS4K1);
    uint32_t k;
    if (fe29_zero_filter(p, 3, 5, k) && fe29_is_kp(p, k)) {
        uint32_t kr;
        if (fe29_zero_filter(r, 3, 5, kr) && fe29_is_kp(r, kr)) return true;
        xyzz_set_inf(acc);
        return false;
    }
    fe29_norm(p, p);
    fe29_norm(r, r);
    fe29_sqr(pp, p);
    fe29_mul(ppp, p, pp);
    fe29_mul(qq, u1, pp);
    fe29_sqr(rr, r);
    fe29_sub3(t, rr, ppp, qq);
    fe29_norm(acc.x, t);
    fe29_sub(t, qq, acc.x, K::BIAS16K2);
    fe29_mul(m1, r, t);
    fe29_mul(m2, s1, ppp);
    fe29_sub(t, m1, m2, K::BIAS4K1);
    fe29_norm(acc.y, t);
    fe29_mul(acc.zz, acc.zz, q.zz);
    fe29_mul(acc.zz, acc.zz, pp);
    fe29_mul(acc.zzz, acc.zzz, q.zzz);
    fe29_mul(acc.zzz, acc.zzz, ppp);
    return false;
}
template <class C>
ZK_HD void xyzz_add(XYZZ<C29<C>>& acc, const XYZZ<C29<C>>& q) {
    if (xyzz_add_nodbl(acc, q)) xyzz_dbl(acc);
}

// ---- conversions (bases on upload; partial sums on the host) ----
template <class C>
ZK_HD void aff29_from_std(Affine<C29<C>>& r, const Affine<C>& p) {
    fe29_from_std(r.x, p.x);
    fe29_from_std(r.y, p.y);
}
template <class C>
ZK_HD void xyzz29_to_std(XYZZ<C>& r, const XYZZ<C29<C>>& p) {
    if (xyzz_is_inf(p)) {
        xyzz_set_inf(r);
        return;
    }
    fe29_to_std(r.x, p.x);
    fe29_to_std(r.y, p.y);
    fe29_to_std(r.zz, p.zz);
    fe29_to_std(r.zzz, p.zzz);
}

}  // namespace zk
