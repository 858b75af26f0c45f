/*
 * TEST INFRASTRUCTURE -- CPU oracle for the MSM / NTT hot path (plain C, gcc).
 *
 * Build:  make -C oracle      ->  oracle/libzkoracle.so
 * Users:  tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  Nothing else.
 *
 * PARITY UNPINNED (see zk_oracle_impl.h header and DESIGN.md): the restated
 * algorithms are those of ark-ec/ark-poly ^0.3.0 and halo2_proofs 0.2, which are
 * not present under /root/reference; they are pinned by pure-Python big-integer
 * fixtures (oracle/pyref.py) and known-answer identities, not by reference outputs.
 *
 * Conventions: little-endian u64 limbs; field elements in Montgomery form unless
 * a name says "canonical"; affine points (x, y) with infinity = (0, 0); every
 * result point is returned in affine form so that comparison is canonical.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "zk_oracle_consts.h"

#define NL 4
#include "zk_oracle_impl.h"
#undef NL
#define NL 6
#include "zk_oracle_impl.h"
#undef NL

#define EXPORT __attribute__((visibility("default")))

static int field_nl(int field) { return ORC_FIELDS[field].n64; }

EXPORT int orc_field_count(void) { return ORC_F_COUNT; }
EXPORT int orc_curve_count(void) { return ORC_C_COUNT; }
EXPORT const char *orc_field_name(int f) { return ORC_FIELDS[f].name; }
EXPORT const char *orc_curve_name(int c) { return ORC_CURVES[c].name; }
EXPORT int orc_field_nlimbs(int f) { return field_nl(f); }
EXPORT int orc_field_bits(int f) { return ORC_FIELDS[f].nbits; }
EXPORT int orc_field_two_adicity(int f) { return ORC_FIELDS[f].two_adicity; }
EXPORT int orc_curve_base_field(int c) { return ORC_CURVES[c].base_field; }
EXPORT int orc_curve_scalar_field(int c) { return ORC_CURVES[c].scalar_field; }
EXPORT void orc_field_modulus(int f, uint64_t *out) { memcpy(out, ORC_FIELDS[f].p, 8 * field_nl(f)); }
EXPORT void orc_field_root(int f, uint64_t *out) { memcpy(out, ORC_FIELDS[f].root_mont, 8 * field_nl(f)); }
EXPORT void orc_field_generator(int f, uint64_t *out) { memcpy(out, ORC_FIELDS[f].gen_mont, 8 * field_nl(f)); }

/* ---- field element ops: op 0 add, 1 sub, 2 mul, 3 inv(a), 4 to_mont(a), 5 from_mont(a), 6 neg(a) ---- */
EXPORT void orc_fe_op(int field, int op, const uint64_t *a, const uint64_t *b, uint64_t *r) {
    if (field_nl(field) == 4) {
        fctx_4 f = {&ORC_FIELDS[field]};
        fe_4 x, y, z;
        memcpy(&x, a, 32);
        if (b) memcpy(&y, b, 32);
        switch (op) {
            case 0: fe_add_4(&f, &z, &x, &y); break;
            case 1: fe_sub_4(&f, &z, &x, &y); break;
            case 2: fe_mul_4(&f, &z, &x, &y); break;
            case 3: fe_inv_4(&f, &z, &x); break;
            case 4: fe_to_mont_4(&f, &z, &x); break;
            case 5: fe_from_mont_4(&f, &z, &x); break;
            default: fe_neg_4(&f, &z, &x); break;
        }
        memcpy(r, &z, 32);
    } else {
        fctx_6 f = {&ORC_FIELDS[field]};
        fe_6 x, y, z;
        memcpy(&x, a, 48);
        if (b) memcpy(&y, b, 48);
        switch (op) {
            case 0: fe_add_6(&f, &z, &x, &y); break;
            case 1: fe_sub_6(&f, &z, &x, &y); break;
            case 2: fe_mul_6(&f, &z, &x, &y); break;
            case 3: fe_inv_6(&f, &z, &x); break;
            case 4: fe_to_mont_6(&f, &z, &x); break;
            case 5: fe_from_mont_6(&f, &z, &x); break;
            default: fe_neg_6(&f, &z, &x); break;
        }
        memcpy(r, &z, 48);
    }
}
/* batch Montgomery <-> canonical over n elements */
EXPORT void orc_fe_batch_convert(int field, int to_mont, const uint64_t *in, uint64_t *out, size_t n) {
    int nl = field_nl(field);
    for (size_t i = 0; i < n; i++) orc_fe_op(field, to_mont ? 4 : 5, in + i * nl, NULL, out + i * nl);
}

/* ---- curve ops: 4-way dispatch over (base-field limbs, coordinate extension degree) ---- */
#define PASTE3_(a, b, c) a##b##c
#define PASTE3(a, b, c) PASTE3_(a, b, c)
#define CURVE_CALL(c, BODY)                                              \
    do {                                                                 \
        const int nl_ = field_nl(ORC_CURVES[c].base_field);              \
        const int ext_ = ORC_CURVES[c].ext;                              \
        if (ext_ == 1 && nl_ == 4) { BODY(_4, 4) }                       \
        else if (ext_ == 1) { BODY(_6, 6) }                              \
        else if (nl_ == 4) { BODY(_g2_4, 4) }                            \
        else { BODY(_g2_6, 6) }                                          \
    } while (0)
/* u64 limbs per coordinate */
static int coord_nl(int c) { return field_nl(ORC_CURVES[c].base_field) * ORC_CURVES[c].ext; }
EXPORT int orc_curve_coord_limbs(int c) { return coord_nl(c); }
EXPORT void orc_curve_generator(int c, uint64_t *out) {
    int nl = coord_nl(c);
    memcpy(out, ORC_CURVES[c].gx_mont, 8 * nl);
    memcpy(out + nl, ORC_CURVES[c].gy_mont, 8 * nl);
}

EXPORT int orc_on_curve(int c, const uint64_t *aff) {
    int ok = 0;
#define BODY(SUF, NLV)                                                         \
    PASTE3(fctx_, NLV, ) f = {&ORC_FIELDS[ORC_CURVES[c].base_field]};          \
    ok = aff_on_curve##SUF(&f, (const void *)ORC_CURVES[c].b_mont, (const aff##SUF *)aff);
    CURVE_CALL(c, BODY);
#undef BODY
    return ok;
}
/* out = [k]P, k canonical 4 x u64 */
EXPORT void orc_scalar_mul(int c, const uint64_t *aff_in, const uint64_t *k, uint64_t *aff_out) {
#define BODY(SUF, NLV)                                                         \
    PASTE3(fctx_, NLV, ) f = {&ORC_FIELDS[ORC_CURVES[c].base_field]};          \
    jac##SUF r;                                                                \
    jac_scalar_mul##SUF(&f, &r, (const aff##SUF *)aff_in, k, 4);               \
    jac_to_aff##SUF(&f, (aff##SUF *)aff_out, &r);
    CURVE_CALL(c, BODY);
#undef BODY
}
/* out = A + B, all affine */
EXPORT void orc_point_add(int c, const uint64_t *a, const uint64_t *b, uint64_t *aff_out) {
#define BODY(SUF, NLV)                                                         \
    PASTE3(fctx_, NLV, ) f = {&ORC_FIELDS[ORC_CURVES[c].base_field]};          \
    jac##SUF r;                                                                \
    jac_set_inf##SUF(&f, &r);                                                  \
    jac_add_mixed##SUF(&f, &r, (const aff##SUF *)a);                           \
    jac_add_mixed##SUF(&f, &r, (const aff##SUF *)b);                           \
    jac_to_aff##SUF(&f, (aff##SUF *)aff_out, &r);
    CURVE_CALL(c, BODY);
#undef BODY
}
/* Jacobian (X,Y,Z) -> affine */
EXPORT void orc_jac_to_affine(int c, const uint64_t *jac, uint64_t *aff_out) {
#define BODY(SUF, NLV)                                                         \
    PASTE3(fctx_, NLV, ) f = {&ORC_FIELDS[ORC_CURVES[c].base_field]};          \
    jac_to_aff##SUF(&f, (aff##SUF *)aff_out, (const jac##SUF *)jac);
    CURVE_CALL(c, BODY);
#undef BODY
}

/* P_i = [k_i]G for n canonical scalars (threaded): the seeded base generator of SURVEY 8d */
typedef struct { int c; const uint64_t *k; uint64_t *out; size_t lo, hi; } genjob;
static void *gen_worker(void *arg) {
    genjob *j = (genjob *)arg;
    int nl = coord_nl(j->c);
    uint64_t g[24];
    orc_curve_generator(j->c, g);
    for (size_t i = j->lo; i < j->hi; i++) orc_scalar_mul(j->c, g, j->k + 4 * i, j->out + 2 * nl * i);
    return NULL;
}
EXPORT void orc_fixed_base_mul(int c, const uint64_t *scalars, size_t n, int threads, uint64_t *aff_out) {
    if (threads < 1) threads = 1;
    pthread_t th[256];
    genjob jobs[256];
    if (threads > 256) threads = 256;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    for (int t = 0; t < threads; t++) {
        jobs[t] = (genjob){c, scalars, aff_out, n * t / threads, n * (t + 1) / threads};
        pthread_create(&th[t], NULL, gen_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}

/* ---- MSM variants; all return the affine result ---- */
EXPORT void orc_msm_naive(int c, const uint64_t *bases, const uint64_t *scalars_canonical, size_t n, uint64_t *aff_out) {
#define BODY(SUF, NLV)                                                         \
    PASTE3(fctx_, NLV, ) f = {&ORC_FIELDS[ORC_CURVES[c].base_field]};          \
    jac##SUF r;                                                                \
    msm_naive##SUF(&f, &r, (const aff##SUF *)bases, scalars_canonical, 4, n);  \
    jac_to_aff##SUF(&f, (aff##SUF *)aff_out, &r);
    CURVE_CALL(c, BODY);
#undef BODY
}
/* ark-ec 0.3 VariableBaseMSM::multi_scalar_mul(bases, scalars: canonical BigInt) */
EXPORT void orc_msm_ark(int c, const uint64_t *bases, const uint64_t *scalars_canonical, size_t n, int threads, uint64_t *aff_out) {
    int bits = ORC_FIELDS[ORC_CURVES[c].scalar_field].nbits;
#define BODY(SUF, NLV)                                                         \
    PASTE3(fctx_, NLV, ) f = {&ORC_FIELDS[ORC_CURVES[c].base_field]};          \
    jac##SUF r;                                                                \
    msm_ark##SUF(&f, &r, (const aff##SUF *)bases, scalars_canonical, 4, n, bits, threads); \
    jac_to_aff##SUF(&f, (aff##SUF *)aff_out, &r);
    CURVE_CALL(c, BODY);
#undef BODY
}
EXPORT int orc_msm_ark_window_bits(size_t n) { return ark_c_4(n); }
/* halo2_proofs 0.2 best_multiexp(coeffs: Montgomery scalars, bases) */
EXPORT void orc_msm_halo2(int c, const uint64_t *bases, const uint64_t *scalars_mont, size_t n, int threads, uint64_t *aff_out) {
    int sf = ORC_CURVES[c].scalar_field;
    uint8_t *repr = (uint8_t *)malloc(32 * n + 8);
    orc_fe_batch_convert(sf, 0, scalars_mont, (uint64_t *)repr, n); /* to_repr(): canonical LE bytes (x86 is LE) */
#define BODY(SUF, NLV)                                                         \
    PASTE3(fctx_, NLV, ) f = {&ORC_FIELDS[ORC_CURVES[c].base_field]};          \
    jac##SUF r;                                                                \
    msm_halo2##SUF(&f, &r, (const aff##SUF *)bases, repr, n, threads);         \
    jac_to_aff##SUF(&f, (aff##SUF *)aff_out, &r);
    CURVE_CALL(c, BODY);
#undef BODY
    free(repr);
}

/* ================================================================ NTT (all scalar fields are 4-limb) */
static void fe4_root_for(const fctx_4 *f, fe_4 *w, int logn) {
    /* ark-poly 0.3 Radix2EvaluationDomain::new: group_gen = TWO_ADIC_ROOT_OF_UNITY^(2^(S - k));
       halo2 0.2 EvaluationDomain::new: omega = ROOT_OF_UNITY squared (S - k) times -- same element */
    memcpy(w, f->c->root_mont, 32);
    for (int i = logn; i < f->c->two_adicity; i++) fe_sqr_4(f, w, w);
}
EXPORT void orc_root_of_unity(int field, int logn, uint64_t *out) {
    fctx_4 f = {&ORC_FIELDS[field]};
    fe_4 w;
    fe4_root_for(&f, &w, logn);
    memcpy(out, &w, 32);
}

/* O(n^2) DFT: out[k] = sum_j in[j] * omega^(jk) */
EXPORT void orc_dft_naive(int field, const uint64_t *in, uint64_t *out, size_t n, const uint64_t *omega_mont) {
    fctx_4 f = {&ORC_FIELDS[field]};
    const fe_4 *a = (const fe_4 *)in;
    fe_4 *o = (fe_4 *)out;
    fe_4 w, wk, x, t, acc;
    memcpy(&w, omega_mont, 32);
    fe_one_4(&f, &wk); /* omega^k */
    for (size_t k = 0; k < n; k++) {
        fe_zero_4(&acc);
        fe_one_4(&f, &x); /* omega^(jk) */
        for (size_t j = 0; j < n; j++) {
            fe_mul_4(&f, &t, &a[j], &x);
            fe_add_4(&f, &acc, &acc, &t);
            fe_mul_4(&f, &x, &x, &wk);
        }
        o[k] = acc;
        fe_mul_4(&f, &wk, &wk, &w);
    }
}

static inline size_t bitrev(size_t k, int logn) {
    size_t r = 0;
    for (int i = 0; i < logn; i++) r |= ((k >> i) & 1) << (logn - 1 - i);
    return r;
}

/* threaded butterfly stages with a barrier per stage */
typedef struct {
    const fctx_4 *f;
    fe_4 *a;
    const fe_4 *tw; /* tw[i] = omega^i, i < n/2 */
    int logn, dit, tid, nthreads;
    pthread_barrier_t *bar;
} nttjob;

static void *ntt_worker(void *arg) {
    nttjob *j = (nttjob *)arg;
    const fctx_4 *f = j->f;
    size_t n = (size_t)1 << j->logn, half = n / 2;
    size_t lo = half * j->tid / j->nthreads, hi = half * (j->tid + 1) / j->nthreads;
    for (int s = 0; s < j->logn; s++) {
        /* DIT (halo2 best_fft after bit-reversal): gap = 2^s ascending.  DIF (ark io_helper): gap descending. */
        int st = j->dit ? s : j->logn - 1 - s;
        size_t gap = (size_t)1 << st, tstride = half >> st;
        for (size_t b = lo; b < hi; b++) {
            size_t k = b & (gap - 1), i0 = ((b >> st) << (st + 1)) | k, i1 = i0 + gap;
            fe_4 t, u = j->a[i0];
            if (j->dit) {
                fe_mul_4(f, &t, &j->a[i1], &j->tw[k * tstride]);
                fe_add_4(f, &j->a[i0], &u, &t);
                fe_sub_4(f, &j->a[i1], &u, &t);
            } else {
                fe_sub_4(f, &t, &u, &j->a[i1]);
                fe_add_4(f, &j->a[i0], &u, &j->a[i1]);
                fe_mul_4(f, &j->a[i1], &t, &j->tw[k * tstride]);
            }
        }
        if (j->nthreads > 1) pthread_barrier_wait(j->bar);
    }
    return NULL;
}
static void butterflies(const fctx_4 *f, fe_4 *a, const fe_4 *omega, int logn, int dit, int threads) {
    size_t n = (size_t)1 << logn;
    if (n < 2) return;
    fe_4 *tw = (fe_4 *)malloc((n / 2) * sizeof(fe_4));
    fe_one_4(f, &tw[0]);
    for (size_t i = 1; i < n / 2; i++) fe_mul_4(f, &tw[i], &tw[i - 1], omega);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if ((size_t)threads > n / 2) threads = 1;
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, threads);
    pthread_t th[256];
    nttjob jobs[256];
    for (int t = 0; t < threads; t++) {
        jobs[t] = (nttjob){f, a, tw, logn, dit, t, threads, &bar};
        if (t) pthread_create(&th[t], NULL, ntt_worker, &jobs[t]);
    }
    ntt_worker(&jobs[0]);
    for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&bar);
    free(tw);
}
static void derange(fe_4 *a, int logn) {
    size_t n = (size_t)1 << logn;
    for (size_t k = 0; k < n; k++) {
        size_t rk = bitrev(k, logn);
        if (k < rk) { fe_4 t = a[k]; a[k] = a[rk]; a[rk] = t; }
    }
}
/* a[i] *= g^i : ark-poly 0.3 Radix2EvaluationDomain::distribute_powers */
static void distribute_powers(const fctx_4 *f, fe_4 *a, size_t n, const fe_4 *g) {
    fe_4 pw;
    fe_one_4(f, &pw);
    for (size_t i = 0; i < n; i++) {
        fe_mul_4(f, &a[i], &a[i], &pw);
        fe_mul_4(f, &pw, &pw, g);
    }
}

/* halo2_proofs 0.2 arithmetic.rs best_fft(a, omega, log_n): bit-reverse, then DIT butterflies; no scaling */
EXPORT void orc_halo2_best_fft(int field, uint64_t *a, const uint64_t *omega_mont, int logn, int threads) {
    fctx_4 f = {&ORC_FIELDS[field]};
    fe_4 w;
    memcpy(&w, omega_mont, 32);
    derange((fe_4 *)a, logn);
    butterflies(&f, (fe_4 *)a, &w, logn, 1, threads);
}

/* ark-poly 0.3 Radix2EvaluationDomain (domain/radix2/{mod,fft}.rs)
 *   kind 0 fft_in_place        : io_helper (DIF, roots of group_gen) then derange
 *   kind 1 ifft_in_place       : derange, oi_helper (DIT, roots of group_gen_inv), scale by size_inv
 *   kind 2 coset_fft_in_place  : distribute_powers(multiplicative_generator) then fft
 *   kind 3 coset_ifft_in_place : ifft then distribute_powers(generator^-1)
 */
EXPORT void orc_ark_fft(int field, uint64_t *a_, int logn, int kind, int threads) {
    fctx_4 f = {&ORC_FIELDS[field]};
    fe_4 *a = (fe_4 *)a_;
    size_t n = (size_t)1 << logn;
    fe_4 w, g;
    fe4_root_for(&f, &w, logn);
    memcpy(&g, f.c->gen_mont, 32);
    if (kind == 2) distribute_powers(&f, a, n, &g);
    if (kind == 0 || kind == 2) {
        butterflies(&f, a, &w, logn, 0, threads);
        derange(a, logn);
    } else {
        fe_4 winv, ninv, nn;
        fe_inv_4(&f, &winv, &w);
        derange(a, logn);
        butterflies(&f, a, &winv, logn, 1, threads);
        fe_zero_4(&nn);
        nn.v[0] = n;
        fe_to_mont_4(&f, &nn, &nn);
        fe_inv_4(&f, &ninv, &nn);
        for (size_t i = 0; i < n; i++) fe_mul_4(&f, &a[i], &a[i], &ninv);
        if (kind == 3) {
            fe_4 ginv;
            fe_inv_4(&f, &ginv, &g);
            distribute_powers(&f, a, n, &ginv);
        }
    }
}
EXPORT void orc_distribute_powers(int field, uint64_t *a, size_t n, const uint64_t *g_mont) {
    fctx_4 f = {&ORC_FIELDS[field]};
    fe_4 g;
    memcpy(&g, g_mont, 32);
    distribute_powers(&f, (fe_4 *)a, n, &g);
}

/* ark-groth16 0.3 r1cs_to_qap.rs R1CStoQAP::witness_map, from the evaluation vectors a, b, c (size 2^logm):
 *   ifft(a), ifft(b), coset_fft(a), coset_fft(b), ab = a.b, ifft(c), coset_fft(c), ab -= c,
 *   divide_by_vanishing_poly_on_coset_in_place(ab)  [ * (g^m - 1)^-1 ], coset_ifft(ab)  ->  h (written to a) */
EXPORT void orc_groth16_witness_map(int field, uint64_t *a_, uint64_t *b_, uint64_t *c_, int logm, int threads) {
    fctx_4 f = {&ORC_FIELDS[field]};
    size_t m = (size_t)1 << logm;
    fe_4 *a = (fe_4 *)a_, *b = (fe_4 *)b_, *c = (fe_4 *)c_;
    orc_ark_fft(field, a_, logm, 1, threads);
    orc_ark_fft(field, b_, logm, 1, threads);
    orc_ark_fft(field, a_, logm, 2, threads);
    orc_ark_fft(field, b_, logm, 2, threads);
    for (size_t i = 0; i < m; i++) fe_mul_4(&f, &a[i], &a[i], &b[i]);
    orc_ark_fft(field, c_, logm, 1, threads);
    orc_ark_fft(field, c_, logm, 2, threads);
    for (size_t i = 0; i < m; i++) fe_sub_4(&f, &a[i], &a[i], &c[i]);
    fe_4 g, z, one;
    memcpy(&g, f.c->gen_mont, 32);
    fe_pow_u64_4(&f, &z, &g, (uint64_t)m);
    fe_one_4(&f, &one);
    fe_sub_4(&f, &z, &z, &one);
    fe_inv_4(&f, &z, &z);
    for (size_t i = 0; i < m; i++) fe_mul_4(&f, &a[i], &a[i], &z);
    orc_ark_fft(field, a_, logm, 3, threads);
}
