// Short-Weierstrass (a = 0) group arithmetic for the MSM path: Pallas, Vesta, BN254 G1,
// BLS12-381 G1.  Bucket accumulators use extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): a mixed add is 8M+2S with no inversion and, unlike
// the Jacobian form ark-ec 0.3 / pasta use on the CPU, needs no squaring of Z.  Results
// leave the library as Jacobian (X, Y, Z) -- the representation of ark-ec 0.3
// `GroupProjective` and pasta_curves 0.4 `Ep/Eq` -- so a Rust shim can rebuild the
// caller's type limb for limb.
//
// Affine points are (x, y) Montgomery; infinity is encoded as (0, 0), which is never on
// y^2 = x^3 + b for b != 0.
//
// The same formulas serve G1 (coordinates in Fq) and the G2 twists of BN254 / BLS12-381
// (coordinates in Fq2): `Coord<C>` is Fe<Fq> or Fe2<Fq> and every fe_* below is an overload.
#pragma once
#include <type_traits>

#include "zk_field.h"
#include "zk_field29.h"

namespace zk {

// EXT: 1 = Fq, 2 = Fq2; 29 / 58 = the lazy-limb views of Fq / Fq2 used by the MSM bucket kernels (zk_curve29.h)
template <class C>
using Coord = std::conditional_t<C::EXT == 29, Fe29<typename C::Fq>,
                                 std::conditional_t<C::EXT == 58, Fe29x2<typename C::Fq>,
                                                    std::conditional_t<C::EXT == 2, Fe2<typename C::Fq>, Fe<typename C::Fq>>>>;
template <class C>
constexpr int coord_words() {
    return C::EXT * C::Fq::N;
}

template <class C>
struct alignas(16) Affine {
    Coord<C> x, y;
};
template <class C>
struct alignas(16) XYZZ {
    Coord<C> x, y, zz, zzz;
};
template <class C>
struct Jacobian {
    Coord<C> x, y, z;
};

template <class C>
ZK_HD bool aff_is_inf(const Affine<C>& p) {
    return fe_is_zero(p.x) && fe_is_zero(p.y);
}
template <class C>
ZK_HD void curve_generator(Affine<C>& g) {
    fe_from_words(g.x, C::GX);
    fe_from_words(g.y, C::GY);
}
template <class C>
ZK_HD void xyzz_set_inf(XYZZ<C>& p) {
    fe_zero(p.x);
    fe_zero(p.y);
    fe_zero(p.zz);
    fe_zero(p.zzz);
}
template <class C>
ZK_HD bool xyzz_is_inf(const XYZZ<C>& p) {
    return fe_is_zero(p.zz);
}
template <class C>
ZK_HD void xyzz_from_affine(XYZZ<C>& r, const Affine<C>& p) {
    if (aff_is_inf(p)) {
        xyzz_set_inf(r);
        return;
    }
    r.x = p.x;
    r.y = p.y;
    fe_one(r.zz);
    fe_one(r.zzz);
}

// mdbl-2008-s-1: r = 2*(affine p), p != inf
template <class C>
ZK_HD void xyzz_dbl_affine(XYZZ<C>& r, const Affine<C>& p) {
    Coord<C> u, v, w, s, m, t;
    fe_dbl(u, p.y);
    fe_sqr(v, u);
    fe_mul(w, u, v);
    fe_mul(s, p.x, v);
    fe_sqr(t, p.x);
    fe_dbl(m, t);
    fe_add(m, m, t);  // 3 x^2
    fe_sqr(r.x, m);
    fe_sub(r.x, r.x, s);
    fe_sub(r.x, r.x, s);
    fe_sub(t, s, r.x);
    fe_mul(t, m, t);
    fe_mul(u, w, p.y);
    fe_sub(r.y, t, u);
    r.zz = v;
    r.zzz = w;
}

// dbl-2008-s-1: p = 2p
template <class C>
ZK_HD void xyzz_dbl(XYZZ<C>& p) {
    if (xyzz_is_inf(p)) return;
    Coord<C> u, v, w, s, m, t, x3;
    fe_dbl(u, p.y);
    fe_sqr(v, u);
    fe_mul(w, u, v);
    fe_mul(s, p.x, v);
    fe_sqr(t, p.x);
    fe_dbl(m, t);
    fe_add(m, m, t);
    fe_sqr(x3, m);
    fe_sub(x3, x3, s);
    fe_sub(x3, x3, s);
    fe_sub(t, s, x3);
    fe_mul(t, m, t);
    fe_mul(u, w, p.y);
    fe_sub(p.y, t, u);
    p.x = x3;
    fe_mul(p.zz, v, p.zz);
    fe_mul(p.zzz, w, p.zzz);
}

// madd-2008-s: acc += (affine q), all special cases handled
template <class C>
ZK_HD void xyzz_add_mixed(XYZZ<C>& acc, const Affine<C>& q) {
    if (aff_is_inf(q)) return;
    if (xyzz_is_inf(acc)) {
        acc.x = q.x;
        acc.y = q.y;
        fe_one(acc.zz);
        fe_one(acc.zzz);
        return;
    }
    Coord<C> p, r, pp, ppp, qq, t;
    fe_mul(p, q.x, acc.zz);
    fe_mul(r, q.y, acc.zzz);
    fe_sub(p, p, acc.x);
    fe_sub(r, r, acc.y);
    if (fe_is_zero(p)) {
        if (fe_is_zero(r)) {
            xyzz_dbl_affine(acc, q);  // same point
        } else {
            xyzz_set_inf(acc);  // opposite points
        }
        return;
    }
    fe_sqr(pp, p);
    fe_mul(ppp, p, pp);
    fe_mul(qq, acc.x, pp);
    fe_sqr(acc.x, r);
    fe_sub(acc.x, acc.x, ppp);
    fe_sub(acc.x, acc.x, qq);
    fe_sub(acc.x, acc.x, qq);
    fe_sub(t, qq, acc.x);
    fe_mul(t, r, t);
    fe_mul(acc.y, acc.y, ppp);
    fe_sub(acc.y, t, acc.y);
    fe_mul(acc.zz, acc.zz, pp);
    fe_mul(acc.zzz, acc.zzz, ppp);
}

// add-2008-s without the doubling branch: acc += q unless acc == q, in which case acc is left alone and true is returned
// (the caller doubles).  Split this way so that a kernel can keep ONE inlined copy of the addition and ONE of the doubling:
// an inlined XYZZ addition is 25-40 KB of code and the instruction cache (shared by two CUs) holds 64 KB.
template <class C>
ZK_HD bool xyzz_add_nodbl(XYZZ<C>& acc, const XYZZ<C>& q) {
    if (xyzz_is_inf(q)) return false;
    if (xyzz_is_inf(acc)) {
        acc = q;
        return false;
    }
    Coord<C> u1, u2, s1, s2, p, r, pp, ppp, qq, t;
    fe_mul(u1, acc.x, q.zz);
    fe_mul(u2, q.x, acc.zz);
    fe_mul(s1, acc.y, q.zzz);
    fe_mul(s2, q.y, acc.zzz);
    fe_sub(p, u2, u1);
    fe_sub(r, s2, s1);
    if (fe_is_zero(p)) {
        if (fe_is_zero(r)) return true;   // same point
        xyzz_set_inf(acc);                // opposite points
        return false;
    }
    fe_sqr(pp, p);
    fe_mul(ppp, p, pp);
    fe_mul(qq, u1, pp);
    fe_sqr(acc.x, r);
    fe_sub(acc.x, acc.x, ppp);
    fe_sub(acc.x, acc.x, qq);
    fe_sub(acc.x, acc.x, qq);
    fe_sub(t, qq, acc.x);
    fe_mul(t, r, t);
    fe_mul(s1, s1, ppp);
    fe_sub(acc.y, t, s1);
    fe_mul(acc.zz, acc.zz, q.zz);
    fe_mul(acc.zz, acc.zz, pp);
    fe_mul(acc.zzz, acc.zzz, q.zzz);
    fe_mul(acc.zzz, acc.zzz, ppp);
    return false;
}

// add-2008-s: acc += q (both XYZZ), all special cases handled
template <class C>
ZK_HD void xyzz_add(XYZZ<C>& acc, const XYZZ<C>& q) {
    if (xyzz_add_nodbl(acc, q)) xyzz_dbl(acc);
}

// XYZZ -> Jacobian: (X*ZZ, Y*ZZZ, ZZ) satisfies x = X'/Z'^2, y = Y'/Z'^3.  Identity -> (0, R, 0)
// (ark-ec 0.3 `GroupProjective::zero()` = (0, 1, 0)).
template <class C>
ZK_HD void xyzz_to_jacobian(Jacobian<C>& r, const XYZZ<C>& p) {
    if (xyzz_is_inf(p)) {
        fe_zero(r.x);
        fe_one(r.y);
        fe_zero(r.z);
        return;
    }
    fe_mul(r.x, p.x, p.zz);
    fe_mul(r.y, p.y, p.zzz);
    r.z = p.zz;
}

// XYZZ -> affine (host-side: one Fermat inversion).  Identity -> (0, 0)
template <class C>
ZK_HD void xyzz_to_affine(Affine<C>& r, const XYZZ<C>& p) {
    if (xyzz_is_inf(p)) {
        fe_zero(r.x);
        fe_zero(r.y);
        return;
    }
    Coord<C> izzz, izz, t;
    fe_inv(izzz, p.zzz);       // 1/z^3
    fe_mul(t, izzz, p.zz);     // 1/z
    fe_sqr(izz, t);            // 1/z^2
    fe_mul(r.x, p.x, izz);
    fe_mul(r.y, p.y, izzz);
}

template <class C>
ZK_HD void aff_neg_if(Affine<C>& p, bool neg) {
    // y -> -y when neg; (0,0) stays (0,0)
    Coord<C> ny;
    fe_neg(ny, p.y);
    fe_cmov(p.y, ny, neg);
}

}  // namespace zk
