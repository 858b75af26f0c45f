/*
 * TEST INFRASTRUCTURE -- CPU oracle, curve / MSM body, generic over the coordinate field.
 * Included from zk_oracle_impl.h twice per limb count: once with coordinates in Fq (G1, Pallas, Vesta)
 * and once with coordinates in Fq2 = Fq[u]/(u^2+1) (the G2 twists of BN254 / BLS12-381,
 * ark-bn254 / ark-bls12-381 0.3 g2.rs).  KT = coordinate type, KOP(op) = its arithmetic, CN(x) = symbol suffixing.
 * PARITY UNPINNED -- see zk_oracle_impl.h.
 */
/* ---------------------------------------------------------------- curve: short Weierstrass, a = 0, Jacobian (ark-ec 0.3 GroupProjective / pasta Ep) */
typedef struct { KT x, y; } CN(aff);   /* infinity encoded as (0,0) -- (0,0) is never on y^2 = x^3 + b, b != 0 */
typedef struct { KT x, y, z; } CN(jac); /* identity: z == 0 */

static inline int CN(aff_is_inf)(const CN(aff) *p) { return KOP(is_zero)(&p->x) && KOP(is_zero)(&p->y); }
static inline void CN(jac_set_inf)(const NM(fctx) *f, CN(jac) *p) {
    KOP(zero)(&p->x);
    KOP(one)(f, &p->y);
    KOP(zero)(&p->z);
}
static inline int CN(jac_is_inf)(const CN(jac) *p) { return KOP(is_zero)(&p->z); }

/* dbl-2009-l (a = 0), as ark-ec 0.3 short_weierstrass_jacobian::double_in_place */
static void CN(jac_double)(const NM(fctx) *f, CN(jac) *p) {
    if (CN(jac_is_inf)(p)) return;
    KT a, b, c, d, e, ff, t;
    KOP(sqr)(f, &a, &p->x);
    KOP(sqr)(f, &b, &p->y);
    KOP(sqr)(f, &c, &b);
    KOP(add)(f, &d, &p->x, &b);
    KOP(sqr)(f, &d, &d);
    KOP(sub)(f, &d, &d, &a);
    KOP(sub)(f, &d, &d, &c);
    KOP(dbl)(f, &d, &d);
    KOP(dbl)(f, &e, &a);
    KOP(add)(f, &e, &e, &a);
    KOP(sqr)(f, &ff, &e);
    KOP(mul)(f, &t, &p->y, &p->z);
    KOP(dbl)(f, &p->z, &t);
    KOP(dbl)(f, &t, &d);
    KOP(sub)(f, &p->x, &ff, &t);
    KOP(sub)(f, &t, &d, &p->x);
    KOP(mul)(f, &t, &t, &e);
    KOP(dbl)(f, &c, &c);
    KOP(dbl)(f, &c, &c);
    KOP(dbl)(f, &c, &c);
    KOP(sub)(f, &p->y, &t, &c);
}

/* madd-2007-bl, as ark-ec 0.3 add_assign_mixed */
static void CN(jac_add_mixed)(const NM(fctx) *f, CN(jac) *p, const CN(aff) *q) {
    if (CN(aff_is_inf)(q)) return;
    if (CN(jac_is_inf)(p)) {
        p->x = q->x;
        p->y = q->y;
        KOP(one)(f, &p->z);
        return;
    }
    KT z1z1, u2, s2, h, hh, i, j, r, v, t;
    KOP(sqr)(f, &z1z1, &p->z);
    KOP(mul)(f, &u2, &q->x, &z1z1);
    KOP(mul)(f, &s2, &p->z, &q->y);
    KOP(mul)(f, &s2, &s2, &z1z1);
    if (KOP(eq)(&p->x, &u2) && KOP(eq)(&p->y, &s2)) {
        CN(jac_double)(f, p);
        return;
    }
    KOP(sub)(f, &h, &u2, &p->x);
    KOP(sqr)(f, &hh, &h);
    KOP(dbl)(f, &i, &hh);
    KOP(dbl)(f, &i, &i);
    KOP(mul)(f, &j, &h, &i);
    KOP(sub)(f, &r, &s2, &p->y);
    KOP(dbl)(f, &r, &r);
    KOP(mul)(f, &v, &p->x, &i);
    /* X3 = r^2 - J - 2V */
    KT x3, y3, z3;
    KOP(sqr)(f, &x3, &r);
    KOP(sub)(f, &x3, &x3, &j);
    KOP(sub)(f, &x3, &x3, &v);
    KOP(sub)(f, &x3, &x3, &v);
    /* Y3 = r (V - X3) - 2 Y1 J */
    KOP(sub)(f, &t, &v, &x3);
    KOP(mul)(f, &y3, &r, &t);
    KOP(mul)(f, &t, &p->y, &j);
    KOP(dbl)(f, &t, &t);
    KOP(sub)(f, &y3, &y3, &t);
    /* Z3 = (Z1 + H)^2 - Z1Z1 - HH */
    KOP(add)(f, &z3, &p->z, &h);
    KOP(sqr)(f, &z3, &z3);
    KOP(sub)(f, &z3, &z3, &z1z1);
    KOP(sub)(f, &z3, &z3, &hh);
    p->x = x3;
    p->y = y3;
    p->z = z3;
}

/* add-2007-bl, as ark-ec 0.3 add_assign */
static void CN(jac_add)(const NM(fctx) *f, CN(jac) *p, const CN(jac) *q) {
    if (CN(jac_is_inf)(q)) return;
    if (CN(jac_is_inf)(p)) {
        *p = *q;
        return;
    }
    KT z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t;
    KOP(sqr)(f, &z1z1, &p->z);
    KOP(sqr)(f, &z2z2, &q->z);
    KOP(mul)(f, &u1, &p->x, &z2z2);
    KOP(mul)(f, &u2, &q->x, &z1z1);
    KOP(mul)(f, &s1, &p->y, &q->z);
    KOP(mul)(f, &s1, &s1, &z2z2);
    KOP(mul)(f, &s2, &q->y, &p->z);
    KOP(mul)(f, &s2, &s2, &z1z1);
    if (KOP(eq)(&u1, &u2) && KOP(eq)(&s1, &s2)) {
        CN(jac_double)(f, p);
        return;
    }
    KOP(sub)(f, &h, &u2, &u1);
    KOP(dbl)(f, &i, &h);
    KOP(sqr)(f, &i, &i);
    KOP(mul)(f, &j, &h, &i);
    KOP(sub)(f, &r, &s2, &s1);
    KOP(dbl)(f, &r, &r);
    KOP(mul)(f, &v, &u1, &i);
    KT x3, y3, z3;
    KOP(sqr)(f, &x3, &r);
    KOP(sub)(f, &x3, &x3, &j);
    KOP(sub)(f, &x3, &x3, &v);
    KOP(sub)(f, &x3, &x3, &v);
    KOP(sub)(f, &t, &v, &x3);
    KOP(mul)(f, &y3, &r, &t);
    KOP(mul)(f, &t, &s1, &j);
    KOP(dbl)(f, &t, &t);
    KOP(sub)(f, &y3, &y3, &t);
    KOP(add)(f, &z3, &p->z, &q->z);
    KOP(sqr)(f, &z3, &z3);
    KOP(sub)(f, &z3, &z3, &z1z1);
    KOP(sub)(f, &z3, &z3, &z2z2);
    KOP(mul)(f, &z3, &z3, &h);
    p->x = x3;
    p->y = y3;
    p->z = z3;
}

/* Jacobian -> affine (x = X/Z^2, y = Y/Z^3); identity -> (0,0) */
static void CN(jac_to_aff)(const NM(fctx) *f, CN(aff) *r, const CN(jac) *p) {
    if (CN(jac_is_inf)(p)) {
        KOP(zero)(&r->x);
        KOP(zero)(&r->y);
        return;
    }
    KT zi, zi2, zi3;
    KOP(inv)(f, &zi, &p->z);
    KOP(sqr)(f, &zi2, &zi);
    KOP(mul)(f, &zi3, &zi2, &zi);
    KOP(mul)(f, &r->x, &p->x, &zi2);
    KOP(mul)(f, &r->y, &p->y, &zi3);
}

static int CN(aff_on_curve)(const NM(fctx) *f, const KT *b, const CN(aff) *p) {
    if (CN(aff_is_inf)(p)) return 1;
    KT l, r;
    KOP(sqr)(f, &l, &p->y);
    KOP(sqr)(f, &r, &p->x);
    KOP(mul)(f, &r, &r, &p->x);
    KOP(add)(f, &r, &r, b);
    return KOP(eq)(&l, &r);
}

/* [k]P by left-to-right double-and-add; k canonical little-endian u64 limbs */
static void CN(jac_scalar_mul)(const NM(fctx) *f, CN(jac) *r, const CN(aff) *p, const uint64_t *k, int klimbs) {
    CN(jac) acc;
    CN(jac_set_inf)(f, &acc);
    for (int i = klimbs * 64 - 1; i >= 0; i--) {
        CN(jac_double)(f, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) CN(jac_add_mixed)(f, &acc, p);
    }
    *r = acc;
}

/* ---------------------------------------------------------------- MSM */
/* bits [start, start+c) of a little-endian 4x64 scalar (ark BigInteger::divn then % 2^c) */
static inline uint64_t CN(scalar_window)(const uint64_t *s, int slimbs, int start, int c) {
    int limb = start / 64, off = start % 64;
    if (limb >= slimbs) return 0;
    uint64_t v = s[limb] >> off;
    if (off + c > 64 && limb + 1 < slimbs) v |= s[limb + 1] << (64 - off);
    return v & ((1ull << c) - 1);
}

/* naive sum of [s_i]P_i */
static void CN(msm_naive)(const NM(fctx) *f, CN(jac) *out, const CN(aff) *bases, const uint64_t *scalars, int slimbs, size_t n) {
    CN(jac) acc, t;
    CN(jac_set_inf)(f, &acc);
    for (size_t i = 0; i < n; i++) {
        CN(jac_scalar_mul)(f, &t, &bases[i], scalars + i * slimbs, slimbs);
        CN(jac_add)(f, &acc, &t);
    }
    *out = acc;
}

/* ark-ec 0.3 msm/variable_base.rs  VariableBaseMSM::multi_scalar_mul  -- one window */
typedef struct {
    const NM(fctx) *f;
    const CN(aff) *bases;
    const uint64_t *scalars; /* canonical, slimbs u64 each */
    int slimbs;
    size_t n;
    int c;
    int num_windows;
    int *next_window; /* shared work counter (atomic) */
    CN(jac) *window_sums;
} CN(ark_job);

static void CN(ark_window)(const CN(ark_job) *j, int w) {
    const NM(fctx) *f = j->f;
    const int c = j->c, w_start = w * c;
    size_t nb = ((size_t)1 << c) - 1;
    CN(jac) res;
    CN(jac_set_inf)(f, &res);
    CN(jac) *buckets = (CN(jac) *)malloc(nb * sizeof(CN(jac)));
    for (size_t b = 0; b < nb; b++) CN(jac_set_inf)(f, &buckets[b]);
    for (size_t i = 0; i < j->n; i++) {
        const uint64_t *s = j->scalars + i * j->slimbs;
        int zero = 1, one = (s[0] == 1);
        for (int k = 0; k < j->slimbs; k++) {
            if (s[k]) zero = 0;
            if (k && s[k]) one = 0;
        }
        if (zero) continue; /* zero scalars are filtered out */
        if (one) {          /* unit scalars: added directly, only in the first window */
            if (w_start == 0) CN(jac_add_mixed)(f, &res, &j->bases[i]);
            continue;
        }
        uint64_t d = CN(scalar_window)(s, j->slimbs, w_start, c);
        if (d) CN(jac_add_mixed)(f, &buckets[d - 1], &j->bases[i]);
    }
    CN(jac) running;
    CN(jac_set_inf)(f, &running);
    for (size_t b = nb; b-- > 0;) {
        CN(jac_add)(f, &running, &buckets[b]);
        CN(jac_add)(f, &res, &running);
    }
    free(buckets);
    j->window_sums[w] = res;
}
static void *CN(ark_worker)(void *arg) {
    CN(ark_job) *j = (CN(ark_job) *)arg;
    for (;;) {
        int w = __atomic_fetch_add(j->next_window, 1, __ATOMIC_RELAXED);
        if (w >= j->num_windows) break;
        CN(ark_window)(j, w);
    }
    return NULL;
}
static int CN(ark_c)(size_t size) {
    if (size < 32) return 3;
    /* ln_without_floats(a) = ark_std::log2(a) * 69 / 100, log2 = ceil */
    int l = 0;
    while (((size_t)1 << l) < size) l++;
    return l * 69 / 100 + 2;
}
static void CN(msm_ark)(const NM(fctx) *f, CN(jac) *out, const CN(aff) *bases, const uint64_t *scalars, int slimbs,
                        size_t n, int scalar_bits, int threads) {
    int c = CN(ark_c)(n);
    int nw = (scalar_bits + c - 1) / c;
    CN(jac) *sums = (CN(jac) *)malloc(nw * sizeof(CN(jac)));
    int next = 0;
    CN(ark_job) job = {f, bases, scalars, slimbs, n, c, nw, &next, sums};
    if (threads < 1) threads = 1;
    if (threads > nw) threads = nw;
    pthread_t *th = (pthread_t *)malloc(threads * sizeof(pthread_t));
    for (int t = 1; t < threads; t++) pthread_create(&th[t], NULL, CN(ark_worker), &job);
    CN(ark_worker)(&job);
    for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    /* lowest + fold(windows[1..] high -> low: total += w; c doublings) */
    CN(jac) total;
    CN(jac_set_inf)(f, &total);
    for (int w = nw - 1; w >= 1; w--) {
        CN(jac_add)(f, &total, &sums[w]);
        for (int k = 0; k < c; k++) CN(jac_double)(f, &total);
    }
    CN(jac_add)(f, &total, &sums[0]);
    free(sums);
    *out = total;
}

/* halo2_proofs 0.2 arithmetic.rs  multiexp_serial / best_multiexp */
typedef struct {
    const NM(fctx) *f;      /* base field */
    const CN(aff) *bases;
    const uint8_t *repr;    /* to_repr(): 32 canonical little-endian bytes per coeff */
    size_t n;
    CN(jac) acc;
} CN(h2_job);

static inline size_t CN(h2_get_at)(size_t segment, size_t c, const uint8_t *bytes) {
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
static void *CN(h2_serial)(void *arg) {
    CN(h2_job) *j = (CN(h2_job) *)arg;
    const NM(fctx) *f = j->f;
    size_t n = j->n, c;
    if (n < 4) c = 1;
    else if (n < 32) c = 3;
    else c = (size_t)ceil(log((double)(uint32_t)n));
    size_t segments = 256 / c + 1, nb = ((size_t)1 << c) - 1;
    CN(jac) *buckets = (CN(jac) *)malloc(nb * sizeof(CN(jac)));
    CN(jac) acc = j->acc;
    for (size_t seg = segments; seg-- > 0;) {
        for (size_t k = 0; k < c; k++) CN(jac_double)(f, &acc);
        for (size_t b = 0; b < nb; b++) CN(jac_set_inf)(f, &buckets[b]);
        for (size_t i = 0; i < n; i++) {
            size_t d = CN(h2_get_at)(seg, c, j->repr + 32 * i);
            /* Bucket::{None,Affine,Projective}: None+base -> Affine, Affine+base -> Projective; same group element as a mixed add */
            if (d) CN(jac_add_mixed)(f, &buckets[d - 1], &j->bases[i]);
        }
        CN(jac) running;
        CN(jac_set_inf)(f, &running);
        for (size_t b = nb; b-- > 0;) {
            CN(jac_add)(f, &running, &buckets[b]);
            CN(jac_add)(f, &acc, &running);
        }
    }
    free(buckets);
    j->acc = acc;
    return NULL;
}
static void CN(msm_halo2)(const NM(fctx) *f, CN(jac) *out, const CN(aff) *bases, const uint8_t *repr, size_t n, int threads) {
    if (threads < 1) threads = 1;
    if (n > (size_t)threads) {
        size_t chunk = n / threads;
        size_t nchunks = (n + chunk - 1) / chunk;
        CN(h2_job) *jobs = (CN(h2_job) *)malloc(nchunks * sizeof(CN(h2_job)));
        pthread_t *th = (pthread_t *)malloc(nchunks * sizeof(pthread_t));
        for (size_t k = 0; k < nchunks; k++) {
            size_t lo = k * chunk, hi = lo + chunk > n ? n : lo + chunk;
            jobs[k].f = f;
            jobs[k].bases = bases + lo;
            jobs[k].repr = repr + 32 * lo;
            jobs[k].n = hi - lo;
            CN(jac_set_inf)(f, &jobs[k].acc);
            pthread_create(&th[k], NULL, CN(h2_serial), &jobs[k]);
        }
        CN(jac) total;
        CN(jac_set_inf)(f, &total);
        for (size_t k = 0; k < nchunks; k++) {
            pthread_join(th[k], NULL);
            CN(jac_add)(f, &total, &jobs[k].acc);
        }
        free(th);
        free(jobs);
        *out = total;
    } else {
        CN(h2_job) job;
        job.f = f;
        job.bases = bases;
        job.repr = repr;
        job.n = n;
        CN(jac_set_inf)(f, &job.acc);
        CN(h2_serial)(&job);
        *out = job.acc;
    }
}

