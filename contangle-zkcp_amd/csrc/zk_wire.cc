// ark-serialize 0.3 wire formats (include/zkcp_amd_prover.h): host-side codecs between the bytes the reference writes
// (lib/src/utils.rs:85-118, circuits-ark/src/utils.rs:12-22) and the library's Montgomery limb layout.  Host arithmetic
// only (the portable field code of zk_field.h); large vectors are converted on all host threads.
#include "zk_internal.h"
#include "zk_host64.h"
#include "zkcp_amd_prover.h"

#include <thread>

using namespace zk;

namespace {

// canonical (non-Montgomery) words of a field element
template <class P>
void fe_canon(uint32_t* w, const Fe<P>& a) {
    Fe<P> t;
    fe_from_mont(t, a);
    for (int i = 0; i < P::N; i++) w[i] = t.v[i];
}
template <class P>
int cmp_words(const uint32_t* a, const uint32_t* b) {
    for (int i = P::N - 1; i >= 0; i--)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
template <class P>
constexpr int fq_bytes() {
    return (P::BITS + 2 + 7) / 8;
}
template <class P>
void put_fp(uint8_t* out, const Fe<P>& a, uint8_t flags) {
    uint32_t w[P::N];
    fe_canon<P>(w, a);
    constexpr int NB = fq_bytes<P>();
    for (int i = 0; i < NB; i++) out[i] = (uint8_t)(w[i / 4] >> (8 * (i % 4)));
    out[NB - 1] |= flags;
}
// -> false when the value is not canonical (>= p)
template <class P>
bool get_fp(Fe<P>& r, const uint8_t* in, bool with_flags, uint8_t* flags) {
    constexpr int NB = fq_bytes<P>();
    Fe<P> t;
    fe_zero(t);
    for (int i = 0; i < NB; i++) {
        uint8_t b = in[i];
        if (i == NB - 1 && with_flags) {
            *flags = b & 0xC0;
            b &= 0x3F;
        }
        t.v[i / 4] |= (uint32_t)b << (8 * (i % 4));
    }
    if (cmp_words<P>(t.v, P::P) >= 0) return false;
    fe_to_mont(r, t);
    return true;
}
// coordinate = Fe or Fe2 (c0 then c1, flags on c1)
template <class P>
void put_coord(uint8_t* out, const Fe<P>& a, uint8_t flags) {
    put_fp<P>(out, a, flags);
}
template <class P>
void put_coord(uint8_t* out, const Fe2<P>& a, uint8_t flags) {
    put_fp<P>(out, a.c0, 0);
    put_fp<P>(out + fq_bytes<P>(), a.c1, flags);
}
template <class P>
bool get_coord(Fe<P>& r, const uint8_t* in, bool with_flags, uint8_t* flags) {
    return get_fp<P>(r, in, with_flags, flags);
}
template <class P>
bool get_coord(Fe2<P>& r, const uint8_t* in, bool with_flags, uint8_t* flags) {
    uint8_t none = 0;
    return get_fp<P>(r.c0, in, false, &none) && get_fp<P>(r.c1, in + fq_bytes<P>(), with_flags, flags);
}
// ark-ff ordering: Fp by canonical integer; Fp2 by c1, then c0
template <class P>
bool larger_than_neg(const Fe<P>& y) {
    Fe<P> ny;
    fe_neg(ny, y);
    uint32_t a[P::N], b[P::N];
    fe_canon<P>(a, y);
    fe_canon<P>(b, ny);
    return cmp_words<P>(a, b) > 0;
}
template <class P>
bool larger_than_neg(const Fe2<P>& y) {
    Fe2<P> ny;
    fe_neg(ny, y);
    uint32_t a[P::N], b[P::N];
    fe_canon<P>(a, y.c1);
    fe_canon<P>(b, ny.c1);
    int c = cmp_words<P>(a, b);
    if (c == 0) {
        fe_canon<P>(a, y.c0);
        fe_canon<P>(b, ny.c0);
        c = cmp_words<P>(a, b);
    }
    return c > 0;
}

// a^e for a little-endian word exponent
template <class P>
void fe_pow_words(Fe<P>& r, const Fe<P>& a, const uint32_t* e, int nwords) {
    Fe<P> acc, base = a;
    fe_one(acc);
    for (int i = 0; i < 32 * nwords; i++) {
        if ((e[i / 32] >> (i % 32)) & 1) fe_mul(acc, acc, base);
        fe_sqr(base, base);
    }
    r = acc;
}
// square root for p = 3 mod 4 (both base fields): a^((p+1)/4); false when a is a non-residue
template <class P>
bool fe_sqrt(Fe<P>& r, const Fe<P>& a) {
    uint32_t e[P::N];
    uint64_t c = 1;   // p + 1
    for (int i = 0; i < P::N; i++) {
        c += P::P[i];
        e[i] = (uint32_t)c;
        c >>= 32;
    }
    for (int i = 0; i < P::N; i++) e[i] = (e[i] >> 2) | (i + 1 < P::N ? e[i + 1] << 30 : (uint32_t)c << 30);
    Fe<P> s, t;
    fe_pow_words(s, a, e, P::N);
    fe_sqr(t, s);
    if (!fe_eq(t, a)) return false;
    r = s;
    return true;
}
// Fq2 = Fq[u]/(u^2+1): complex method
template <class P>
bool fe_sqrt(Fe2<P>& r, const Fe2<P>& a) {
    if (fe_is_zero(a.c1)) {
        Fe<P> s;
        if (fe_sqrt(s, a.c0)) {
            r.c0 = s;
            fe_zero(r.c1);
            return true;
        }
        Fe<P> n;
        fe_neg(n, a.c0);
        if (!fe_sqrt(s, n)) return false;
        fe_zero(r.c0);
        r.c1 = s;
        return true;
    }
    Fe<P> n0, n1, alpha, two, inv2, delta, c0, t;
    fe_sqr(n0, a.c0);
    fe_sqr(n1, a.c1);
    fe_add(n0, n0, n1);
    if (!fe_sqrt(alpha, n0)) return false;
    fe_one(two);
    fe_add(two, two, two);
    fe_inv(inv2, two);
    fe_add(delta, a.c0, alpha);
    fe_mul(delta, delta, inv2);
    if (!fe_sqrt(c0, delta)) {
        fe_sub(delta, a.c0, alpha);
        fe_mul(delta, delta, inv2);
        if (!fe_sqrt(c0, delta)) return false;
    }
    fe_add(t, c0, c0);
    fe_inv(t, t);
    r.c0 = c0;
    fe_mul(r.c1, a.c1, t);
    return true;
}

template <class C>
void curve_b(Coord<C>& b) {
    fe_from_words(b, C::B);
}
template <class C>
bool on_curve(const Affine<C>& p) {
    Coord<C> l, r, b;
    fe_sqr(l, p.y);
    fe_sqr(r, p.x);
    fe_mul(r, r, p.x);
    curve_b<C>(b);
    fe_add(r, r, b);
    return fe_eq(l, r);
}
template <class C>
constexpr int coord_bytes() {
    return fq_bytes<typename C::Fq>() * C::EXT;
}
template <class C>
constexpr int point_bytes(int compressed) {
    return coord_bytes<C>() * (compressed ? 1 : 2);
}

template <class C>
void encode_one(uint8_t* out, const Affine<C>& p, int compressed) {
    constexpr int CB = coord_bytes<C>();
    memset(out, 0, (size_t)point_bytes<C>(compressed));
    const bool inf = aff_is_inf(p);
    if (compressed) {
        if (inf) {
            out[CB - 1] |= 1 << 6;
            return;
        }
        put_coord(out, p.x, larger_than_neg(p.y) ? (uint8_t)(1 << 7) : (uint8_t)0);
        return;
    }
    if (inf) {   // GroupAffine::zero() = (0, 1, infinity)
        Coord<C> one;
        fe_one(one);
        put_coord(out + CB, one, 1 << 6);
        return;
    }
    put_coord(out, p.x, 0);
    put_coord(out + CB, p.y, 0);
}
template <class C>
int decode_one(Affine<C>& p, const uint8_t* in, int compressed, int check) {
    constexpr int CB = coord_bytes<C>();
    uint8_t flags = 0;
    if (compressed) {
        if (!get_coord(p.x, in, true, &flags)) return ZK_ERR_INVALID_ARG;
        if (flags == 0xC0) return ZK_ERR_INVALID_ARG;
        if (flags & 0x40) {
            fe_zero(p.x);
            fe_zero(p.y);
            return ZK_OK;
        }
        Coord<C> rhs, b, y;
        fe_sqr(rhs, p.x);
        fe_mul(rhs, rhs, p.x);
        curve_b<C>(b);
        fe_add(rhs, rhs, b);
        if (!fe_sqrt(y, rhs)) return ZK_ERR_INVALID_ARG;
        if (larger_than_neg(y) != ((flags & 0x80) != 0)) fe_neg(y, y);
        p.y = y;
        return ZK_OK;
    }
    uint8_t none = 0;
    if (!get_coord(p.x, in, false, &none)) return ZK_ERR_INVALID_ARG;
    if (!get_coord(p.y, in + CB, true, &flags)) return ZK_ERR_INVALID_ARG;
    if (flags == 0xC0) return ZK_ERR_INVALID_ARG;
    if (flags & 0x40) {
        fe_zero(p.x);
        fe_zero(p.y);
        return ZK_OK;
    }
    if (check && !on_curve<C>(p)) return ZK_ERR_INVALID_ARG;
    return ZK_OK;
}

// run fn(i) for i in [0, n) on the host threads; the first non-zero status wins
template <class Fn>
int parallel_for(uint64_t n, Fn fn) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 64) nt = 64;
    if (n < 4096 || nt == 1) {
        for (uint64_t i = 0; i < n; i++) {
            int s = fn(i);
            if (s != ZK_OK) return s;
        }
        return ZK_OK;
    }
    std::vector<int> status(nt, ZK_OK);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t]() {
            const uint64_t lo = n * t / nt, hi = n * (t + 1) / nt;
            for (uint64_t i = lo; i < hi; i++) {
                int s = fn(i);
                if (s != ZK_OK) {
                    status[t] = s;
                    return;
                }
            }
        });
    for (auto& t : th) t.join();
    for (int s : status)
        if (s != ZK_OK) return s;
    return ZK_OK;
}

template <class C>
int encode_points(const void* aff, uint64_t n, int compressed, uint8_t* out) {
    const Affine<C>* p = (const Affine<C>*)aff;
    const size_t pb = (size_t)point_bytes<C>(compressed);
    return parallel_for(n, [&](uint64_t i) {
        Affine<C> q;
        memcpy(&q, (const unsigned char*)p + i * 2 * 4 * coord_words<C>(), 2 * 4 * coord_words<C>());
        encode_one<C>(out + i * pb, q, compressed);
        return (int)ZK_OK;
    });
}
template <class C>
int decode_points(const uint8_t* in, uint64_t n, int compressed, int check, void* aff_out) {
    const size_t pb = (size_t)point_bytes<C>(compressed);
    return parallel_for(n, [&](uint64_t i) {
        Affine<C> q;
        int s = decode_one<C>(q, in + i * pb, compressed, check);
        if (s != ZK_OK) return s;
        memcpy((unsigned char*)aff_out + i * 2 * 4 * coord_words<C>(), &q, 2 * 4 * coord_words<C>());
        return (int)ZK_OK;
    });
}

// ---- Groth16 proof assembly on 64-bit host limbs (the same view the MSM tail uses)
template <class C>
void load_affine_h(HostXYZZ<C>& r, const void* aff) {
    Affine<C> a;
    memcpy(&a, aff, 2 * 4 * coord_words<C>());
    XYZZ<C> x;
    xyzz_from_affine(x, a);
    to_host<C>(r, x);
}
template <class C>
void load_jacobian_h(HostXYZZ<C>& r, const void* jac) {
    Jacobian<C> j;
    memcpy(&j, jac, 3 * 4 * coord_words<C>());
    XYZZ<C> x;
    if (fe_is_zero(j.z)) {
        xyzz_set_inf(x);
    } else {
        x.x = j.x;
        x.y = j.y;
        fe_sqr(x.zz, j.z);
        fe_mul(x.zzz, x.zz, j.z);
    }
    to_host<C>(r, x);
}
// [k] P for a canonical scalar k (little-endian words), double-and-add, MSB first
template <class C>
void scalar_mul_h(HostXYZZ<C>& r, const HostXYZZ<C>& p, const uint32_t* k, int nwords) {
    HostXYZZ<C> acc;
    xyzz_set_inf(acc);
    for (int i = 32 * nwords - 1; i >= 0; i--) {
        xyzz_dbl(acc);
        if ((k[i / 32] >> (i % 32)) & 1) xyzz_add(acc, p);
    }
    r = acc;
}
template <class C>
void store_affine_h(void* out, const HostXYZZ<C>& p) {
    XYZZ<C> x;
    from_host<C>(x, p);
    Affine<C> a;
    xyzz_to_affine(a, x);
    memcpy(out, &a, 2 * 4 * coord_words<C>());
}
template <class G1, class G2>
int assemble(const zk_groth16_assembly* in, void* oa, void* ob, void* oc) {
    using Fr = typename G1::Fr;
    Fe<Fr> r, s, rs;
    memcpy(&r, in->r, sizeof r);
    memcpy(&s, in->s, sizeof s);
    fe_mul(rs, r, s);
    uint32_t rw[Fr::N], sw[Fr::N], rsw[Fr::N];
    fe_canon<Fr>(rw, r);
    fe_canon<Fr>(sw, s);
    fe_canon<Fr>(rsw, rs);
    HostXYZZ<G1> alpha, beta1, delta1, a0, b10, a_acc, b1_acc, l_acc, h_acc, t, ga, gb1, gc;
    load_affine_h<G1>(alpha, in->alpha_g1);
    load_affine_h<G1>(beta1, in->beta_g1);
    load_affine_h<G1>(delta1, in->delta_g1);
    load_affine_h<G1>(a0, in->a_query0);
    load_affine_h<G1>(b10, in->b_g1_query0);
    load_jacobian_h<G1>(a_acc, in->a_acc);
    load_jacobian_h<G1>(b1_acc, in->b_g1_acc);
    load_jacobian_h<G1>(l_acc, in->l_acc);
    load_jacobian_h<G1>(h_acc, in->h_acc);
    // A = r delta + a_query[0] + a_acc + alpha          (calculate_coeff(r delta_g1, a_query, alpha_g1, assignment))
    scalar_mul_h<G1>(ga, delta1, rw, Fr::N);
    xyzz_add(ga, a0);
    xyzz_add(ga, a_acc);
    xyzz_add(ga, alpha);
    // B1 = s delta + b_g1_query[0] + b_g1_acc + beta_g1
    scalar_mul_h<G1>(gb1, delta1, sw, Fr::N);
    xyzz_add(gb1, b10);
    xyzz_add(gb1, b1_acc);
    xyzz_add(gb1, beta1);
    // C = s A + r B1 - r s delta + l_acc + h_acc
    scalar_mul_h<G1>(gc, ga, sw, Fr::N);
    scalar_mul_h<G1>(t, gb1, rw, Fr::N);
    xyzz_add(gc, t);
    scalar_mul_h<G1>(t, delta1, rsw, Fr::N);
    {   // subtract: negate y
        XYZZ<G1> tt;
        from_host<G1>(tt, t);
        if (!xyzz_is_inf(tt)) fe_neg(tt.y, tt.y);
        to_host<G1>(t, tt);
    }
    xyzz_add(gc, t);
    xyzz_add(gc, l_acc);
    xyzz_add(gc, h_acc);
    // B = s delta_g2 + b_g2_query[0] + b_g2_acc + beta_g2
    HostXYZZ<G2> beta2, delta2, b20, b2_acc, gb2;
    load_affine_h<G2>(beta2, in->beta_g2);
    load_affine_h<G2>(delta2, in->delta_g2);
    load_affine_h<G2>(b20, in->b_g2_query0);
    load_jacobian_h<G2>(b2_acc, in->b_g2_acc);
    scalar_mul_h<G2>(gb2, delta2, sw, Fr::N);
    xyzz_add(gb2, b20);
    xyzz_add(gb2, b2_acc);
    xyzz_add(gb2, beta2);
    store_affine_h<G1>(oa, ga);
    store_affine_h<G2>(ob, gb2);
    store_affine_h<G1>(oc, gc);
    return ZK_OK;
}

bool pairing_curves(zk_pairing_t p, zk_curve_t* g1, zk_curve_t* g2) {
    if (p == ZK_PAIRING_BN254) {
        *g1 = ZK_BN254_G1;
        *g2 = ZK_BN254_G2;
        return true;
    }
    if (p == ZK_PAIRING_BLS12_381) {
        *g1 = ZK_BLS12_381_G1;
        *g2 = ZK_BLS12_381_G2;
        return true;
    }
    return false;
}
}  // namespace

extern "C" {
#define API __attribute__((visibility("default")))

API int zk_ark_point_size(zk_curve_t c, int compressed) {
    switch (c) {
        case ZK_BN254_G1: return point_bytes<Bn254G1>(compressed);
        case ZK_BN254_G2: return point_bytes<Bn254G2>(compressed);
        case ZK_BLS12_381_G1: return point_bytes<Bls381G1>(compressed);
        case ZK_BLS12_381_G2: return point_bytes<Bls381G2>(compressed);
        default: return ZK_ERR_UNSUPPORTED;   // the Pasta curves are not arkworks types
    }
}
API int zk_ark_points_encode(zk_curve_t c, const void* aff, uint64_t n, int compressed, uint8_t* out) {
    if (n && (!aff || !out)) return ZK_ERR_INVALID_ARG;
    switch (c) {
        case ZK_BN254_G1: return encode_points<Bn254G1>(aff, n, compressed, out);
        case ZK_BN254_G2: return encode_points<Bn254G2>(aff, n, compressed, out);
        case ZK_BLS12_381_G1: return encode_points<Bls381G1>(aff, n, compressed, out);
        case ZK_BLS12_381_G2: return encode_points<Bls381G2>(aff, n, compressed, out);
        default: return ZK_ERR_UNSUPPORTED;
    }
}
API int zk_ark_points_decode(zk_curve_t c, const uint8_t* in, uint64_t n, int compressed, int check, void* aff_out) {
    if (n && (!in || !aff_out)) return ZK_ERR_INVALID_ARG;
    switch (c) {
        case ZK_BN254_G1: return decode_points<Bn254G1>(in, n, compressed, check, aff_out);
        case ZK_BN254_G2: return decode_points<Bn254G2>(in, n, compressed, check, aff_out);
        case ZK_BLS12_381_G1: return decode_points<Bls381G1>(in, n, compressed, check, aff_out);
        case ZK_BLS12_381_G2: return decode_points<Bls381G2>(in, n, compressed, check, aff_out);
        default: return ZK_ERR_UNSUPPORTED;
    }
}

API int zk_ark_scalars_encode(zk_field_t f, const void* mont, uint64_t n, uint8_t* out) {
    if (n && (!mont || !out)) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, {
        constexpr int NB = (F::BITS + 7) / 8;
        const Fe<F>* a = (const Fe<F>*)mont;
        return parallel_for(n, [&](uint64_t i) {
            Fe<F> x;
            memcpy(&x, (const unsigned char*)a + i * 4 * F::N, 4 * F::N);
            uint32_t w[F::N];
            fe_canon<F>(w, x);
            for (int k = 0; k < NB; k++) out[i * NB + k] = (uint8_t)(w[k / 4] >> (8 * (k % 4)));
            return (int)ZK_OK;
        });
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_ark_scalars_decode(zk_field_t f, const uint8_t* in, uint64_t n, void* mont_out) {
    if (n && (!in || !mont_out)) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, {
        constexpr int NB = (F::BITS + 7) / 8;
        return parallel_for(n, [&](uint64_t i) {
            Fe<F> t, r;
            fe_zero(t);
            for (int k = 0; k < NB; k++) t.v[k / 4] |= (uint32_t)in[i * NB + k] << (8 * (k % 4));
            if (cmp_words<F>(t.v, F::P) >= 0) return (int)ZK_ERR_INVALID_ARG;
            fe_to_mont(r, t);
            memcpy((unsigned char*)mont_out + i * 4 * F::N, &r, 4 * F::N);
            return (int)ZK_OK;
        });
    });
    return ZK_ERR_INVALID_ARG;
}

API int zk_ark_proving_key_index(zk_pairing_t p, const uint8_t* buf, uint64_t len, zk_ark_pk_index* out) {
    zk_curve_t g1, g2;
    if (!buf || !out || !pairing_curves(p, &g1, &g2)) return ZK_ERR_INVALID_ARG;
    const uint64_t s1 = (uint64_t)zk_ark_point_size(g1, 0), s2 = (uint64_t)zk_ark_point_size(g2, 0);
    uint64_t off = 0;
    bool ok = true;
    auto one = [&](zk_ark_span& sp, uint64_t sz) {
        sp.offset = off;
        sp.count = 1;
        off += sz;
        if (off > len) ok = false;
    };
    auto vec = [&](zk_ark_span& sp, uint64_t sz) {
        if (!ok || off + 8 > len) {
            ok = false;
            return;
        }
        uint64_t n = 0;
        for (int i = 0; i < 8; i++) n |= (uint64_t)buf[off + i] << (8 * i);
        off += 8;
        if (n > (len - off) / sz) {
            ok = false;
            return;
        }
        sp.offset = off;
        sp.count = n;
        off += n * sz;
    };
    memset(out, 0, sizeof *out);
    one(out->alpha_g1, s1);
    one(out->beta_g2, s2);
    one(out->gamma_g2, s2);
    one(out->delta_g2, s2);
    vec(out->gamma_abc_g1, s1);
    if (ok) one(out->beta_g1, s1);
    if (ok) one(out->delta_g1, s1);
    vec(out->a_query, s1);
    vec(out->b_g1_query, s1);
    vec(out->b_g2_query, s2);
    vec(out->h_query, s1);
    vec(out->l_query, s1);
    if (!ok) return ZK_ERR_INVALID_ARG;
    out->total_bytes = off;
    return ZK_OK;
}

API int zk_bases_upload_ark(zk_curve_t c, const uint8_t* in, uint64_t n, uint64_t* handle_out) {
    if (!handle_out || (n && !in)) return ZK_ERR_INVALID_ARG;
    size_t esz = 0;
    CURVE_SWITCH(c, esz = sizeof(Affine<C>));
    std::vector<unsigned char> host(esz * (n ? n : 1));
    ZK_TRY(zk_ark_points_decode(c, in, n, 0, 0, host.data()));
    return zk_bases_upload(c, host.data(), n, handle_out);
}

API int zk_ark_proof_size(zk_pairing_t p) {
    zk_curve_t g1, g2;
    if (!pairing_curves(p, &g1, &g2)) return ZK_ERR_INVALID_ARG;
    return 2 * zk_ark_point_size(g1, 1) + zk_ark_point_size(g2, 1);
}
API int zk_ark_proof_encode(zk_pairing_t p, const void* a, const void* b, const void* c, uint8_t* out) {
    zk_curve_t g1, g2;
    if (!a || !b || !c || !out || !pairing_curves(p, &g1, &g2)) return ZK_ERR_INVALID_ARG;
    const int s1 = zk_ark_point_size(g1, 1), s2 = zk_ark_point_size(g2, 1);
    ZK_TRY(zk_ark_points_encode(g1, a, 1, 1, out));
    ZK_TRY(zk_ark_points_encode(g2, b, 1, 1, out + s1));
    return zk_ark_points_encode(g1, c, 1, 1, out + s1 + s2);
}
API int zk_ark_proof_decode(zk_pairing_t p, const uint8_t* in, void* a, void* b, void* c) {
    zk_curve_t g1, g2;
    if (!a || !b || !c || !in || !pairing_curves(p, &g1, &g2)) return ZK_ERR_INVALID_ARG;
    const int s1 = zk_ark_point_size(g1, 1), s2 = zk_ark_point_size(g2, 1);
    ZK_TRY(zk_ark_points_decode(g1, in, 1, 1, 0, a));
    ZK_TRY(zk_ark_points_decode(g2, in + s1, 1, 1, 0, b));
    return zk_ark_points_decode(g1, in + s1 + s2, 1, 1, 0, c);
}

API int zk_groth16_assemble_proof(zk_pairing_t p, const zk_groth16_assembly* in, void* a, void* b, void* c) {
    if (!in || !a || !b || !c) return ZK_ERR_INVALID_ARG;
    const void* const* fields = (const void* const*)in;
    for (size_t i = 0; i < sizeof(zk_groth16_assembly) / sizeof(void*); i++)
        if (!fields[i]) return ZK_ERR_INVALID_ARG;
    if (p == ZK_PAIRING_BN254) return assemble<Bn254G1, Bn254G2>(in, a, b, c);
    if (p == ZK_PAIRING_BLS12_381) return assemble<Bls381G1, Bls381G2>(in, a, b, c);
    return ZK_ERR_INVALID_ARG;
}

}  // extern "C"
