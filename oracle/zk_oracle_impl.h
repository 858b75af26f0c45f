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

/* ---------------------------------------------------------------- curve: short Weierstrass, a = 0, Jacobian (ark-ec 0.3 GroupProjective / pasta Ep) */
typedef struct { NM(fe) x, y; } NM(aff);   /* infinity encoded as (0,0) -- (0,0) is never on y^2 = x^3 + b, b != 0 */
typedef struct { NM(fe) x, y, z; } NM(jac); /* identity: z == 0 */

static inline int NM(aff_is_inf)(const NM(aff) *p) { return NM(fe_is_zero)(&p->x) && NM(fe_is_zero)(&p->y); }
static inline void NM(jac_set_inf)(const NM(fctx) *f, NM(jac) *p) {
    NM(fe_zero)(&p->x);
    NM(fe_one)(f, &p->y);
    NM(fe_zero)(&p->z);
}
static inline int NM(jac_is_inf)(const NM(jac) *p) { return NM(fe_is_zero)(&p->z); }

/* dbl-2009-l (a = 0), as ark-ec 0.3 short_weierstrass_jacobian::double_in_place */
static void NM(jac_double)(const NM(fctx) *f, NM(jac) *p) {
    if (NM(jac_is_inf)(p)) return;
    NM(fe) a, b, c, d, e, ff, t;
    NM(fe_sqr)(f, &a, &p->x);
    NM(fe_sqr)(f, &b, &p->y);
    NM(fe_sqr)(f, &c, &b);
    NM(fe_add)(f, &d, &p->x, &b);
    NM(fe_sqr)(f, &d, &d);
    NM(fe_sub)(f, &d, &d, &a);
    NM(fe_sub)(f, &d, &d, &c);
    NM(fe_dbl)(f, &d, &d);
    NM(fe_dbl)(f, &e, &a);
    NM(fe_add)(f, &e, &e, &a);
    NM(fe_sqr)(f, &ff, &e);
    NM(fe_mul)(f, &t, &p->y, &p->z);
    NM(fe_dbl)(f, &p->z, &t);
    NM(fe_dbl)(f, &t, &d);
    NM(fe_sub)(f, &p->x, &ff, &t);
    NM(fe_sub)(f, &t, &d, &p->x);
    NM(fe_mul)(f, &t, &t, &e);
    NM(fe_dbl)(f, &c, &c);
    NM(fe_dbl)(f, &c, &c);
    NM(fe_dbl)(f, &c, &c);
    NM(fe_sub)(f, &p->y, &t, &c);
}

/* madd-2007-bl, as ark-ec 0.3 add_assign_mixed */
static void NM(jac_add_mixed)(const NM(fctx) *f, NM(jac) *p, const NM(aff) *q) {
    if (NM(aff_is_inf)(q)) return;
    if (NM(jac_is_inf)(p)) {
        p->x = q->x;
        p->y = q->y;
        NM(fe_one)(f, &p->z);
        return;
    }
    NM(fe) z1z1, u2, s2, h, hh, i, j, r, v, t;
    NM(fe_sqr)(f, &z1z1, &p->z);
    NM(fe_mul)(f, &u2, &q->x, &z1z1);
    NM(fe_mul)(f, &s2, &p->z, &q->y);
    NM(fe_mul)(f, &s2, &s2, &z1z1);
    if (NM(fe_eq)(&p->x, &u2) && NM(fe_eq)(&p->y, &s2)) {
        NM(jac_double)(f, p);
        return;
    }
    NM(fe_sub)(f, &h, &u2, &p->x);
    NM(fe_sqr)(f, &hh, &h);
    NM(fe_dbl)(f, &i, &hh);
    NM(fe_dbl)(f, &i, &i);
    NM(fe_mul)(f, &j, &h, &i);
    NM(fe_sub)(f, &r, &s2, &p->y);
    NM(fe_dbl)(f, &r, &r);
    NM(fe_mul)(f, &v, &p->x, &i);
    /* X3 = r^2 - J - 2V */
    NM(fe) x3, y3, z3;
    NM(fe_sqr)(f, &x3, &r);
    NM(fe_sub)(f, &x3, &x3, &j);
    NM(fe_sub)(f, &x3, &x3, &v);
    NM(fe_sub)(f, &x3, &x3, &v);
    /* Y3 = r (V - X3) - 2 Y1 J */
    NM(fe_sub)(f, &t, &v, &x3);
    NM(fe_mul)(f, &y3, &r, &t);
    NM(fe_mul)(f, &t, &p->y, &j);
    NM(fe_dbl)(f, &t, &t);
    NM(fe_sub)(f, &y3, &y3, &t);
    /* Z3 = (Z1 + H)^2 - Z1Z1 - HH */
    NM(fe_add)(f, &z3, &p->z, &h);
    NM(fe_sqr)(f, &z3, &z3);
    NM(fe_sub)(f, &z3, &z3, &z1z1);
    NM(fe_sub)(f, &z3, &z3, &hh);
    p->x = x3;
    p->y = y3;
    p->z = z3;
}

/* add-2007-bl, as ark-ec 0.3 add_assign */
static void NM(jac_add)(const NM(fctx) *f, NM(jac) *p, const NM(jac) *q) {
    if (NM(jac_is_inf)(q)) return;
    if (NM(jac_is_inf)(p)) {
        *p = *q;
        return;
    }
    NM(fe) z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t;
    NM(fe_sqr)(f, &z1z1, &p->z);
    NM(fe_sqr)(f, &z2z2, &q->z);
    NM(fe_mul)(f, &u1, &p->x, &z2z2);
    NM(fe_mul)(f, &u2, &q->x, &z1z1);
    NM(fe_mul)(f, &s1, &p->y, &q->z);
    NM(fe_mul)(f, &s1, &s1, &z2z2);
    NM(fe_mul)(f, &s2, &q->y, &p->z);
    NM(fe_mul)(f, &s2, &s2, &z1z1);
    if (NM(fe_eq)(&u1, &u2) && NM(fe_eq)(&s1, &s2)) {
        NM(jac_double)(f, p);
        return;
    }
    NM(fe_sub)(f, &h, &u2, &u1);
    NM(fe_dbl)(f, &i, &h);
    NM(fe_sqr)(f, &i, &i);
    NM(fe_mul)(f, &j, &h, &i);
    NM(fe_sub)(f, &r, &s2, &s1);
    NM(fe_dbl)(f, &r, &r);
    NM(fe_mul)(f, &v, &u1, &i);
    NM(fe) x3, y3, z3;
    NM(fe_sqr)(f, &x3, &r);
    NM(fe_sub)(f, &x3, &x3, &j);
    NM(fe_sub)(f, &x3, &x3, &v);
    NM(fe_sub)(f, &x3, &x3, &v);
    NM(fe_sub)(f, &t, &v, &x3);
    NM(fe_mul)(f, &y3, &r, &t);
    NM(fe_mul)(f, &t, &s1, &j);
    NM(fe_dbl)(f, &t, &t);
    NM(fe_sub)(f, &y3, &y3, &t);
    NM(fe_add)(f, &z3, &p->z, &q->z);
    NM(fe_sqr)(f, &z3, &z3);
    NM(fe_sub)(f, &z3, &z3, &z1z1);
    NM(fe_sub)(f, &z3, &z3, &z2z2);
    NM(fe_mul)(f, &z3, &z3, &h);
    p->x = x3;
    p->y = y3;
    p->z = z3;
}

/* Jacobian -> affine (x = X/Z^2, y = Y/Z^3); identity -> (0,0) */
static void NM(jac_to_aff)(const NM(fctx) *f, NM(aff) *r, const NM(jac) *p) {
    if (NM(jac_is_inf)(p)) {
        NM(fe_zero)(&r->x);
        NM(fe_zero)(&r->y);
        return;
    }
    NM(fe) zi, zi2, zi3;
    NM(fe_inv)(f, &zi, &p->z);
    NM(fe_sqr)(f, &zi2, &zi);
    NM(fe_mul)(f, &zi3, &zi2, &zi);
    NM(fe_mul)(f, &r->x, &p->x, &zi2);
    NM(fe_mul)(f, &r->y, &p->y, &zi3);
}

static int NM(aff_on_curve)(const NM(fctx) *f, const NM(fe) *b, const NM(aff) *p) {
    if (NM(aff_is_inf)(p)) return 1;
    NM(fe) l, r;
    NM(fe_sqr)(f, &l, &p->y);
    NM(fe_sqr)(f, &r, &p->x);
    NM(fe_mul)(f, &r, &r, &p->x);
    NM(fe_add)(f, &r, &r, b);
    return NM(fe_eq)(&l, &r);
}

/* [k]P by left-to-right double-and-add; k canonical little-endian u64 limbs */
static void NM(jac_scalar_mul)(const NM(fctx) *f, NM(jac) *r, const NM(aff) *p, const uint64_t *k, int klimbs) {
    NM(jac) acc;
    NM(jac_set_inf)(f, &acc);
    for (int i = klimbs * 64 - 1; i >= 0; i--) {
        NM(jac_double)(f, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) NM(jac_add_mixed)(f, &acc, p);
    }
    *r = acc;
}

/* ---------------------------------------------------------------- MSM */
/* bits [start, start+c) of a little-endian 4x64 scalar (ark BigInteger::divn then % 2^c) */
static inline uint64_t NM(scalar_window)(const uint64_t *s, int slimbs, int start, int c) {
    int limb = start / 64, off = start % 64;
    if (limb >= slimbs) return 0;
    uint64_t v = s[limb] >> off;
    if (off + c > 64 && limb + 1 < slimbs) v |= s[limb + 1] << (64 - off);
    return v & ((1ull << c) - 1);
}

/* naive sum of [s_i]P_i */
static void NM(msm_naive)(const NM(fctx) *f, NM(jac) *out, const NM(aff) *bases, const uint64_t *scalars, int slimbs, size_t n) {
    NM(jac) acc, t;
    NM(jac_set_inf)(f, &acc);
    for (size_t i = 0; i < n; i++) {
        NM(jac_scalar_mul)(f, &t, &bases[i], scalars + i * slimbs, slimbs);
        NM(jac_add)(f, &acc, &t);
    }
    *out = acc;
}

/* ark-ec 0.3 msm/variable_base.rs  VariableBaseMSM::multi_scalar_mul  -- one window */
typedef struct {
    const NM(fctx) *f;
    const NM(aff) *bases;
    const uint64_t *scalars; /* canonical, slimbs u64 each */
    int slimbs;
    size_t n;
    int c;
    int num_windows;
    int *next_window; /* shared work counter (atomic) */
    NM(jac) *window_sums;
} NM(ark_job);

static void NM(ark_window)(const NM(ark_job) *j, int w) {
    const NM(fctx) *f = j->f;
    const int c = j->c, w_start = w * c;
    size_t nb = ((size_t)1 << c) - 1;
    NM(jac) res;
    NM(jac_set_inf)(f, &res);
    NM(jac) *buckets = (NM(jac) *)malloc(nb * sizeof(NM(jac)));
    for (size_t b = 0; b < nb; b++) NM(jac_set_inf)(f, &buckets[b]);
    for (size_t i = 0; i < j->n; i++) {
        const uint64_t *s = j->scalars + i * j->slimbs;
        int zero = 1, one = (s[0] == 1);
        for (int k = 0; k < j->slimbs; k++) {
            if (s[k]) zero = 0;
            if (k && s[k]) one = 0;
        }
        if (zero) continue; /* zero scalars are filtered out */
        if (one) {          /* unit scalars: added directly, only in the first window */
            if (w_start == 0) NM(jac_add_mixed)(f, &res, &j->bases[i]);
            continue;
        }
        uint64_t d = NM(scalar_window)(s, j->slimbs, w_start, c);
        if (d) NM(jac_add_mixed)(f, &buckets[d - 1], &j->bases[i]);
    }
    NM(jac) running;
    NM(jac_set_inf)(f, &running);
    for (size_t b = nb; b-- > 0;) {
        NM(jac_add)(f, &running, &buckets[b]);
        NM(jac_add)(f, &res, &running);
    }
    free(buckets);
    j->window_sums[w] = res;
}
static void *NM(ark_worker)(void *arg) {
    NM(ark_job) *j = (NM(ark_job) *)arg;
    for (;;) {
        int w = __atomic_fetch_add(j->next_window, 1, __ATOMIC_RELAXED);
        if (w >= j->num_windows) break;
        NM(ark_window)(j, w);
    }
    return NULL;
}
static int NM(ark_c)(size_t size) {
    if (size < 32) return 3;
    /* ln_without_floats(a) = ark_std::log2(a) * 69 / 100, log2 = ceil */
    int l = 0;
    while (((size_t)1 << l) < size) l++;
    return l * 69 / 100 + 2;
}
static void NM(msm_ark)(const NM(fctx) *f, NM(jac) *out, const NM(aff) *bases, const uint64_t *scalars, int slimbs,
                        size_t n, int scalar_bits, int threads) {
    int c = NM(ark_c)(n);
    int nw = (scalar_bits + c - 1) / c;
    NM(jac) *sums = (NM(jac) *)malloc(nw * sizeof(NM(jac)));
    int next = 0;
    NM(ark_job) job = {f, bases, scalars, slimbs, n, c, nw, &next, sums};
    if (threads < 1) threads = 1;
    if (threads > nw) threads = nw;
    pthread_t *th = (pthread_t *)malloc(threads * sizeof(pthread_t));
    for (int t = 1; t < threads; t++) pthread_create(&th[t], NULL, NM(ark_worker), &job);
    NM(ark_worker)(&job);
    for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    /* lowest + fold(windows[1..] high -> low: total += w; c doublings) */
    NM(jac) total;
    NM(jac_set_inf)(f, &total);
    for (int w = nw - 1; w >= 1; w--) {
        NM(jac_add)(f, &total, &sums[w]);
        for (int k = 0; k < c; k++) NM(jac_double)(f, &total);
    }
    NM(jac_add)(f, &total, &sums[0]);
    free(sums);
    *out = total;
}

/* halo2_proofs 0.2 arithmetic.rs  multiexp_serial / best_multiexp */
typedef struct {
    const NM(fctx) *f;      /* base field */
    const NM(aff) *bases;
    const uint8_t *repr;    /* to_repr(): 32 canonical little-endian bytes per coeff */
    size_t n;
    NM(jac) acc;
} NM(h2_job);

static inline size_t NM(h2_get_at)(size_t segment, size_t c, const uint8_t *bytes) {
    size_t skip_bits = segment * c, skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    uint8_t v[8] = {0};
    for (size_t k = 0; k < 8 && skip_bytes + k < 32; k++) v[k] = bytes[skip_bytes + k];
    uint64_t tmp = 0;
    for (int k = 7; k >= 0; k--) tmp = (tmp << 8) | v[k];
    tmp >>= skip_bits - skip_bytes * 8;
    tmp %= ((uint64_t)1 << c);
    return (size_t)tmp;
}
static void *NM(h2_serial)(void *arg) {
    NM(h2_job) *j = (NM(h2_job) *)arg;
    const NM(fctx) *f = j->f;
    size_t n = j->n, c;
    if (n < 4) c = 1;
    else if (n < 32) c = 3;
    else c = (size_t)ceil(log((double)(uint32_t)n));
    size_t segments = 256 / c + 1, nb = ((size_t)1 << c) - 1;
    NM(jac) *buckets = (NM(jac) *)malloc(nb * sizeof(NM(jac)));
    NM(jac) acc = j->acc;
    for (size_t seg = segments; seg-- > 0;) {
        for (size_t k = 0; k < c; k++) NM(jac_double)(f, &acc);
        for (size_t b = 0; b < nb; b++) NM(jac_set_inf)(f, &buckets[b]);
        for (size_t i = 0; i < n; i++) {
            size_t d = NM(h2_get_at)(seg, c, j->repr + 32 * i);
            /* Bucket::{None,Affine,Projective}: None+base -> Affine, Affine+base -> Projective; same group element as a mixed add */
            if (d) NM(jac_add_mixed)(f, &buckets[d - 1], &j->bases[i]);
        }
        NM(jac) running;
        NM(jac_set_inf)(f, &running);
        for (size_t b = nb; b-- > 0;) {
            NM(jac_add)(f, &running, &buckets[b]);
            NM(jac_add)(f, &acc, &running);
        }
    }
    free(buckets);
    j->acc = acc;
    return NULL;
}
static void NM(msm_halo2)(const NM(fctx) *f, NM(jac) *out, const NM(aff) *bases, const uint8_t *repr, size_t n, int threads) {
    if (threads < 1) threads = 1;
    if (n > (size_t)threads) {
        size_t chunk = n / threads;
        size_t nchunks = (n + chunk - 1) / chunk;
        NM(h2_job) *jobs = (NM(h2_job) *)malloc(nchunks * sizeof(NM(h2_job)));
        pthread_t *th = (pthread_t *)malloc(nchunks * sizeof(pthread_t));
        for (size_t k = 0; k < nchunks; k++) {
            size_t lo = k * chunk, hi = lo + chunk > n ? n : lo + chunk;
            jobs[k].f = f;
            jobs[k].bases = bases + lo;
            jobs[k].repr = repr + 32 * lo;
            jobs[k].n = hi - lo;
            NM(jac_set_inf)(f, &jobs[k].acc);
            pthread_create(&th[k], NULL, NM(h2_serial), &jobs[k]);
        }
        NM(jac) total;
        NM(jac_set_inf)(f, &total);
        for (size_t k = 0; k < nchunks; k++) {
            pthread_join(th[k], NULL);
            NM(jac_add)(f, &total, &jobs[k].acc);
        }
        free(th);
        free(jobs);
        *out = total;
    } else {
        NM(h2_job) job;
        job.f = f;
        job.bases = bases;
        job.repr = repr;
        job.n = n;
        NM(jac_set_inf)(f, &job.acc);
        NM(h2_serial)(&job);
        *out = job.acc;
    }
}

#undef NM
#undef CAT
#undef CAT_
