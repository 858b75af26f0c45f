/*
 * TEST INFRASTRUCTURE -- CPU oracle, limb-count-generic body.  Included twice by
 * zk_oracle.c with NL = 4 (Pasta, BN254, BLS12-381 Fr) and NL = 6 (BLS12-381 Fq).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.  The product library never links or calls this.
 *
 * PARITY UNPINNED: the algorithms restated here live in third-party crates that
 * are NOT in /root/reference:
 *   ark-ff / ark-ec / ark-poly ^0.3.0   (circuits-ark/Cargo.toml:10-17, lib/Cargo.toml:33-40)
 *   halo2_proofs 0.2, pasta_curves 0.4  (circuits-halo2/Cargo.toml:10-14)
 * reached from the reference only through Groth16::prove call sites
 *   lib/src/zk/verifiable_encryption.rs:92, lib/src/zk/encryption.rs:76,
 *   lib/src/zk/sample_entries.rs:86, lib/src/zk/property.rs:133
 * and the reference's tests hold no MSM/NTT vectors.  The restatement follows the
 * published algorithms of those versions and is pinned by mathematics instead:
 * pure-Python big-integer fixtures (oracle/pyref.py -> tests/golden/) and
 * known-answer identities (tests/test_oracle.py).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define NM(x) CAT(CAT(x, _), NL)

typedef struct { uint64_t v[NL]; } NM(fe);

typedef struct {
    const orc_field_consts *c;
} NM(fctx);

/* ---------------------------------------------------------------- field: ark-ff 0.3 Fp256/Fp384 Montgomery (R = 2^(64 NL)) */
static inline int NM(fe_is_zero)(const NM(fe) *a) {
    uint64_t o = 0;
    for (int i = 0; i < NL; i++) o |= a->v[i];
    return o == 0;
}
static inline int NM(fe_eq)(const NM(fe) *a, const NM(fe) *b) {
    uint64_t o = 0;
    for (int i = 0; i < NL; i++) o |= a->v[i] ^ b->v[i];
    return o == 0;
}
static inline int NM(geq_p)(const uint64_t *a, const uint64_t *p) {
    for (int i = NL - 1; i >= 0; i--) {
        if (a[i] > p[i]) return 1;
        if (a[i] < p[i]) return 0;
    }
    return 1;
}
static inline void NM(sub_p)(uint64_t *a, const uint64_t *p) {
    unsigned __int128 br = 0;
    for (int i = 0; i < NL; i++) {
        unsigned __int128 d = (unsigned __int128)a[i] - p[i] - (uint64_t)br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static inline void NM(fe_add)(const NM(fctx) *f, NM(fe) *r, const NM(fe) *a, const NM(fe) *b) {
    unsigned __int128 c = 0;
    uint64_t t[NL];
    for (int i = 0; i < NL; i++) {
        c += (unsigned __int128)a->v[i] + b->v[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    /* all supported moduli leave at least one spare top bit, so c == 0 here */
    if (NM(geq_p)(t, f->c->p)) NM(sub_p)(t, f->c->p);
    for (int i = 0; i < NL; i++) r->v[i] = t[i];
}
static inline void NM(fe_sub)(const NM(fctx) *f, NM(fe) *r, const NM(fe) *a, const NM(fe) *b) {
    unsigned __int128 br = 0;
    uint64_t t[NL];
    for (int i = 0; i < NL; i++) {
        unsigned __int128 d = (unsigned __int128)a->v[i] - b->v[i] - (uint64_t)br;
        t[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        unsigned __int128 c = 0;
        for (int i = 0; i < NL; i++) {
            c += (unsigned __int128)t[i] + f->c->p[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    for (int i = 0; i < NL; i++) r->v[i] = t[i];
}
static inline void NM(fe_neg)(const NM(fctx) *f, NM(fe) *r, const NM(fe) *a) {
    NM(fe) z;
    memset(&z, 0, sizeof z);
    NM(fe_sub)(f, r, &z, a);
}
static inline void NM(fe_dbl)(const NM(fctx) *f, NM(fe) *r, const NM(fe) *a) { NM(fe_add)(f, r, a, a); }

/* CIOS Montgomery product: r = a*b*R^-1 mod p */
static inline void NM(fe_mul)(const NM(fctx) *f, NM(fe) *r, const NM(fe) *a, const NM(fe) *b) {
    const uint64_t *p = f->c->p;
    const uint64_t inv = f->c->inv64;
    uint64_t t[NL + 2];
    for (int i = 0; i < NL + 2; i++) t[i] = 0;
    for (int i = 0; i < NL; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < NL; j++) {
            c += (unsigned __int128)a->v[j] * b->v[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[NL];
        t[NL] = (uint64_t)c;
        t[NL + 1] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * inv;
        c = (unsigned __int128)m * p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < NL; j++) {
            c += (unsigned __int128)m * p[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[NL];
        t[NL - 1] = (uint64_t)c;
        t[NL] = t[NL + 1] + (uint64_t)(c >> 64);
    }
    if (t[NL] || NM(geq_p)(t, p)) NM(sub_p)(t, p);
    for (int i = 0; i < NL; i++) r->v[i] = t[i];
}
static inline void NM(fe_sqr)(const NM(fctx) *f, NM(fe) *r, const NM(fe) *a) { NM(fe_mul)(f, r, a, a); }

static inline void NM(fe_one)(const NM(fctx) *f, NM(fe) *r) {
    for (int i = 0; i < NL; i++) r->v[i] = f->c->r[i];
}
static inline void NM(fe_zero)(NM(fe) *r) { memset(r, 0, sizeof *r); }
/* canonical integer -> Montgomery (mul by R^2), and back (mul by 1): ark-ff from_repr / into_repr */
static inline void NM(fe_to_mont)(const NM(fctx) *f, NM(fe) *r, const NM(fe) *a) {
    NM(fe) r2;
    for (int i = 0; i < NL; i++) r2.v[i] = f->c->r2[i];
    NM(fe_mul)(f, r, a, &r2);
}
static inline void NM(fe_from_mont)(const NM(fctx) *f, NM(fe) *r, const NM(fe) *a) {
    NM(fe) one;
    memset(&one, 0, sizeof one);
    one.v[0] = 1;
    NM(fe_mul)(f, r, a, &one);
}
/* r = a^e, e little-endian u64 limbs */
static void NM(fe_pow)(const NM(fctx) *f, NM(fe) *r, const NM(fe) *a, const uint64_t *e, int elimbs) {
    NM(fe) acc, base = *a;
    NM(fe_one)(f, &acc);
    for (int i = 0; i < elimbs * 64; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) NM(fe_mul)(f, &acc, &acc, &base);
        NM(fe_sqr)(f, &base, &base);
    }
    *r = acc;
}
static void NM(fe_pow_u64)(const NM(fctx) *f, NM(fe) *r, const NM(fe) *a, uint64_t e) { NM(fe_pow)(f, r, a, &e, 1); }
/* Fermat inverse a^(p-2); 0 -> 0 */
static void NM(fe_inv)(const NM(fctx) *f, NM(fe) *r, const NM(fe) *a) {
    uint64_t e[NL];
    for (int i = 0; i < NL; i++) e[i] = f->c->p[i];
    e[0] -= 2; /* p odd and > 2: no borrow */
    NM(fe_pow)(f, r, a, e, NL);
}

/* ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1): ark-ff 0.3 Fp2, NONRESIDUE = -1 */
typedef struct { NM(fe) c0, c1; } NM(fe2);
static inline int NM(fe2_is_zero)(const NM(fe2) *a) { return NM(fe_is_zero)(&a->c0) && NM(fe_is_zero)(&a->c1); }
static inline int NM(fe2_eq)(const NM(fe2) *a, const NM(fe2) *b) { return NM(fe_eq)(&a->c0, &b->c0) && NM(fe_eq)(&a->c1, &b->c1); }
static inline void NM(fe2_zero)(NM(fe2) *r) { memset(r, 0, sizeof *r); }
static inline void NM(fe2_one)(const NM(fctx) *f, NM(fe2) *r) { NM(fe_one)(f, &r->c0); NM(fe_zero)(&r->c1); }
static inline void NM(fe2_add)(const NM(fctx) *f, NM(fe2) *r, const NM(fe2) *a, const NM(fe2) *b) {
    NM(fe_add)(f, &r->c0, &a->c0, &b->c0);
    NM(fe_add)(f, &r->c1, &a->c1, &b->c1);
}
static inline void NM(fe2_sub)(const NM(fctx) *f, NM(fe2) *r, const NM(fe2) *a, const NM(fe2) *b) {
    NM(fe_sub)(f, &r->c0, &a->c0, &b->c0);
    NM(fe_sub)(f, &r->c1, &a->c1, &b->c1);
}
static inline void NM(fe2_neg)(const NM(fctx) *f, NM(fe2) *r, const NM(fe2) *a) {
    NM(fe_neg)(f, &r->c0, &a->c0);
    NM(fe_neg)(f, &r->c1, &a->c1);
}
static inline void NM(fe2_dbl)(const NM(fctx) *f, NM(fe2) *r, const NM(fe2) *a) { NM(fe2_add)(f, r, a, a); }
/* schoolbook: (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u */
static inline void NM(fe2_mul)(const NM(fctx) *f, NM(fe2) *r, const NM(fe2) *a, const NM(fe2) *b) {
    NM(fe) t0, t1, t2, t3;
    NM(fe_mul)(f, &t0, &a->c0, &b->c0);
    NM(fe_mul)(f, &t1, &a->c1, &b->c1);
    NM(fe_mul)(f, &t2, &a->c0, &b->c1);
    NM(fe_mul)(f, &t3, &a->c1, &b->c0);
    NM(fe_sub)(f, &r->c0, &t0, &t1);
    NM(fe_add)(f, &r->c1, &t2, &t3);
}
static inline void NM(fe2_sqr)(const NM(fctx) *f, NM(fe2) *r, const NM(fe2) *a) { NM(fe2_mul)(f, r, a, a); }
static void NM(fe2_inv)(const NM(fctx) *f, NM(fe2) *r, const NM(fe2) *a) {
    NM(fe) n, t;
    NM(fe_sqr)(f, &n, &a->c0);
    NM(fe_sqr)(f, &t, &a->c1);
    NM(fe_add)(f, &n, &n, &t);
    NM(fe_inv)(f, &n, &n);
    NM(fe_neg)(f, &t, &a->c1);
    NM(fe_mul)(f, &r->c0, &a->c0, &n);
    NM(fe_mul)(f, &r->c1, &t, &n);
}

/* curve + MSM bodies: coordinates in Fq, then in Fq2 */
#define KT NM(fe)
#define KOP(op) NM(fe_##op)
#define CN(x) NM(x)
#include "zk_oracle_curve.h"
#undef KT
#undef KOP
#undef CN
#define KT NM(fe2)
#define KOP(op) NM(fe2_##op)
#define CN(x) NM(x##_g2)
#include "zk_oracle_curve.h"
#undef KT
#undef KOP
#undef CN

#undef NM
#undef CAT
#undef CAT_
