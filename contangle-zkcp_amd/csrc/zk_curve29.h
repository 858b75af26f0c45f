// XYZZ point arithmetic over the F29 view (zk_field29.h) of Pallas, Vesta and BN254 G1: the bucket arithmetic of the MSM
// kernels.  `C29<C>` is the curve C seen through 9 x 29-bit lazy limbs; the MSM kernels are instantiated with it and pick
// up the overloads below instead of the generic formulas of zk_curve.h.
//
// Stored-point invariant (checked for every formula by tools/check_f29_bounds.py; "N+" = limbs <= 2^29 + 6):
//     X: N+, value < 12p      Y: N+, value < 8p (strict, < 4p, once it has been through a formula)      ZZ, ZZZ: strict limbs, value < 2p      identity: ZZ = literal 0
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
    fe29_zero(m1);
    fe29_sub(m2, m1, w, K::BIAS4K1);  // 4p - w
    fe29_norm(m2, m2);
    fe29_mulacc(r.y, m, t, m2, q.y);  // m t - w y: one reduction for both products; strict, < 2p
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
    fe29_zero(m1);
    fe29_sub(m2, m1, w, K::BIAS4K1);   // 4p - w
    fe29_norm(m2, m2);
    fe29_mulacc(p.y, m, t, m2, p.y);   // m t - w y: strict, < 2.2p
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
    fe29_zero(m1);
    fe29_sub(m2, m1, acc.y, K::BIAS16K2);  // 16p - Y1
    fe29_norm(m2, m2);
    fe29_mulacc(acc.y, r, t, m2, ppp);     // R (Q - X3) - Y1 PPP with one reduction: strict, < 3.7p
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
    fe29_zero(m1);
    fe29_sub(m2, m1, s1, K::BIAS4K1);      // 4p - S1
    fe29_norm(m2, m2);
    fe29_mulacc(acc.y, r, t, m2, ppp);     // R (Q - X3) - S1 PPP: strict
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

// ------------------------------------------------------------------------------------------------------------------
// G2: the same four formulas over Fe29x2 (zk_field29.h).  `C29x2<C>` is the twist curve C (coordinates in Fq2) seen
// through lazy limbs.  Stored-point invariant (tools/check_f29_bounds.py, "ext2" shapes), per Fq2 component:
//     X, ZZ, ZZZ: strict limbs, value < 2p      Y: N+, value < 6.9p      identity: ZZ = literal 0
// X is passed through fe29_refresh (one multiplication by R' mod p per component) before it is stored: the Fq2 product
// sums two double products and negates an operand, so an X left at ~13p as on G1 would push every later bias up.
// Each fe29_mul names the bias that negates its SECOND operand's c1; each fe29_sqr the bias of its a0 - a1.
// ------------------------------------------------------------------------------------------------------------------
template <class C>
struct C29x2 {
    using Base = C;
    using Fq = typename C::Fq;
    using Fr = typename C::Fr;
    static constexpr int EXT = 58;
};

template <class C>
ZK_HD bool aff_is_inf(const Affine<C29x2<C>>& p) {
    return fe29_is_literal_zero(p.x) && fe29_is_literal_zero(p.y);
}
template <class C>
ZK_HD void xyzz_set_inf(XYZZ<C29x2<C>>& p) {
    fe29_zero(p.x);
    fe29_zero(p.y);
    fe29_zero(p.zz);
    fe29_zero(p.zzz);
}
template <class C>
ZK_HD bool xyzz_is_inf(const XYZZ<C29x2<C>>& p) {
    return fe29_is_literal_zero(p.zz);
}
template <class C>
ZK_HD void aff_neg_if(Affine<C29x2<C>>& p, bool neg) {
    using F = typename C::Fq;
    Fe29x2<F> z, ny;
    fe29_zero(z);
    fe29_sub(ny, z, p.y, F29<F>::BIAS4K1);
    fe29_norm(ny, ny);
    fe29_cmov(p.y, ny, neg && !fe29_is_literal_zero(p.x));
}

// shared tail of dbl-2008-s-1 / mdbl-2008-s-1: from (x, y) to (x3, y3, v, w)
template <class F>
ZK_HD void xyzz29x2_dbl_core(Fe29x2<F>& x3, Fe29x2<F>& y3, Fe29x2<F>& v, Fe29x2<F>& w, const Fe29x2<F>& x, const Fe29x2<F>& y) {
    using K = F29<F>;
    Fe29x2<F> u, s, m, t, m1, m2;
    fe29_add(u, y, y);
    fe29_norm(u, u);
    fe29_sqr(v, u, K::BIAS16K2);
    fe29_mul(w, u, v, K::BIAS4K1);
    fe29_mul(s, x, v, K::BIAS4K1);
    fe29_sqr(t, x, K::BIAS4K1);
    fe29_add(m, t, t);
    fe29_add(m, m, t);
    fe29_norm(m, m);
    fe29_sqr(x3, m, K::BIAS4K2);
    fe29_sub2x(x3, x3, s);
    fe29_norm(x3, x3);
    fe29_refresh(x3, x3);
    fe29_sub(t, s, x3, K::BIAS4K1);
    fe29_norm(t, t);
    fe29_mul(m1, m, t, K::BIAS8K2);
    fe29_mul(m2, w, y, K::BIAS8K2);
    fe29_sub(t, m1, m2, K::BIAS4K1);
    fe29_norm(y3, t);
}
template <class C>
ZK_HD void xyzz_dbl_affine(XYZZ<C29x2<C>>& r, const Affine<C29x2<C>>& q) {
    using F = typename C::Fq;
    Fe29x2<F> x3, y3, v, w;
    xyzz29x2_dbl_core<F>(x3, y3, v, w, q.x, q.y);
    r.x = x3;
    r.y = y3;
    fe29_refresh(r.zz, v);   // (2y)^2 of a negated base reaches 4p: bring it under the stored 2p (rare path)
    r.zzz = w;
}
template <class C>
ZK_HD void xyzz_dbl(XYZZ<C29x2<C>>& p) {
    using F = typename C::Fq;
    using K = F29<F>;
    if (xyzz_is_inf(p)) return;
    Fe29x2<F> x3, y3, v, w;
    xyzz29x2_dbl_core<F>(x3, y3, v, w, p.x, p.y);
    p.x = x3;
    p.y = y3;
    fe29_mul(p.zz, v, p.zz, K::BIAS4K1);
    fe29_mul(p.zzz, w, p.zzz, K::BIAS4K1);
}

// shared tail of madd-2008-s / add-2008-s once P = U2 - U1 and R = S2 - S1 are known (normalised) and non-zero:
//   x1u = X1 (or U1), y1s = Y1 (or S1); rbias = the bias that dominates R's components (fe29_sqr / fe29_mul negation)
template <class F>
ZK_HD void xyzz29x2_add_core(Fe29x2<F>& x3, Fe29x2<F>& y3, Fe29x2<F>& pp, Fe29x2<F>& ppp, const Fe29x2<F>& p, const Fe29x2<F>& r,
                             const Fe29x2<F>& x1u, const Fe29x2<F>& y1s, const uint32_t (&rbias)[F29<F>::L]) {
    using K = F29<F>;
    Fe29x2<F> qq, rr, t, m1, m2;
    fe29_sqr(pp, p, K::BIAS8K2);
    fe29_mul(ppp, p, pp, K::BIAS4K1);
    fe29_mul(qq, x1u, pp, K::BIAS4K1);
    fe29_sqr(rr, r, rbias);
    fe29_sub3(t, rr, ppp, qq);
    fe29_norm(t, t);
    fe29_refresh(x3, t);
    fe29_sub(t, qq, x3, K::BIAS4K1);
    fe29_norm(t, t);
    fe29_mul(m1, t, r, rbias);
    fe29_mul(m2, y1s, ppp, K::BIAS4K1);
    fe29_sub(t, m1, m2, K::BIAS4K1);
    fe29_norm(y3, t);
}

// acc += q (affine): madd-2008-s
template <class C>
ZK_HD void xyzz_add_mixed(XYZZ<C29x2<C>>& acc, const Affine<C29x2<C>>& q) {
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
    Fe29x2<F> u2, s2, p, r, pp, ppp, x3, y3;
    fe29_mul(u2, q.x, acc.zz, K::BIAS4K1);
    fe29_mul(s2, q.y, acc.zzz, K::BIAS4K1);
    fe29_sub(p, u2, acc.x, K::BIAS4K1);    // components: integers in (2p, 5.1p)
    fe29_sub(r, s2, acc.y, K::BIAS8K2);    // (1.1p, 9.1p)
    if (fe29_is_zero_mod_p(p, 3, 5)) {
        if (fe29_is_zero_mod_p(r, 2, 9)) {
            xyzz_dbl_affine(acc, q);
        } else {
            xyzz_set_inf(acc);
        }
        return;
    }
    fe29_norm(p, p);
    fe29_norm(r, r);
    xyzz29x2_add_core<F>(x3, y3, pp, ppp, p, r, acc.x, acc.y, K::BIAS16K2);
    acc.x = x3;
    acc.y = y3;
    fe29_mul(acc.zz, acc.zz, pp, K::BIAS4K1);
    fe29_mul(acc.zzz, acc.zzz, ppp, K::BIAS4K1);
}

// acc += q (both XYZZ) unless acc == q (then acc is untouched and true is returned: the caller doubles): add-2008-s
template <class C>
ZK_HD bool xyzz_add_nodbl(XYZZ<C29x2<C>>& acc, const XYZZ<C29x2<C>>& q) {
    using F = typename C::Fq;
    using K = F29<F>;
    if (xyzz_is_inf(q)) return false;
    if (xyzz_is_inf(acc)) {
        acc = q;
        return false;
    }
    Fe29x2<F> u1, u2, s1, s2, p, r, pp, ppp, x3, y3;
    fe29_mul(u1, acc.x, q.zz, K::BIAS4K1);
    fe29_mul(u2, q.x, acc.zz, K::BIAS4K1);
    fe29_mul(s1, acc.y, q.zzz, K::BIAS4K1);
    fe29_mul(s2, q.y, acc.zzz, K::BIAS4K1);
    fe29_sub(p, u2, u1, K::BIAS4K1);   // components: integers in (2.9p, 5.1p)
    fe29_sub(r, s2, s1, K::BIAS4K1);
    if (fe29_is_zero_mod_p(p, 3, 5)) {
        if (fe29_is_zero_mod_p(r, 3, 5)) return true;
        xyzz_set_inf(acc);
        return false;
    }
    fe29_norm(p, p);
    fe29_norm(r, r);
    xyzz29x2_add_core<F>(x3, y3, pp, ppp, p, r, u1, s1, K::BIAS8K2);
    acc.x = x3;
    acc.y = y3;
    fe29_mul(acc.zz, acc.zz, q.zz, K::BIAS4K1);
    fe29_mul(acc.zz, acc.zz, pp, K::BIAS4K1);
    fe29_mul(acc.zzz, acc.zzz, q.zzz, K::BIAS4K1);
    fe29_mul(acc.zzz, acc.zzz, ppp, K::BIAS4K1);
    return false;
}
template <class C>
ZK_HD void xyzz_add(XYZZ<C29x2<C>>& acc, const XYZZ<C29x2<C>>& q) {
    if (xyzz_add_nodbl(acc, q)) xyzz_dbl(acc);
}

template <class C>
ZK_HD void aff29_from_std(Affine<C29x2<C>>& r, const Affine<C>& p) {
    fe29_from_std(r.x, p.x);
    fe29_from_std(r.y, p.y);
}
template <class C>
ZK_HD void xyzz29_to_std(XYZZ<C>& r, const XYZZ<C29x2<C>>& p) {
    if (xyzz_is_inf(p)) {
        xyzz_set_inf(r);
        return;
    }
    fe29_to_std(r.x, p.x);
    fe29_to_std(r.y, p.y);
    fe29_to_std(r.zz, p.zz);
    fe29_to_std(r.zzz, p.zzz);
}

// the lazy-limb view of curve C: C29 for G1 (coordinates in Fq), C29x2 for the G2 twists (Fq2)
template <class C>
using F29View = std::conditional_t<C::EXT == 2, C29x2<C>, C29<C>>;

}  // namespace zk
