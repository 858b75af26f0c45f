/*
 * zkcp_amd -- C ABI of the MI355X (gfx950) MSM / NTT proving backend.
 *
 * The reference (nulltea/contangle-zkcp) has no FFI or plugin seam of its own: its prover
 * calls `Groth16::<Bls12_381>::prove` (lib/src/zk/verifiable_encryption.rs:92,
 * lib/src/zk/encryption.rs:76, lib/src/zk/sample_entries.rs:86, lib/src/zk/property.rs:133)
 * and all MSM/NTT arithmetic happens inside un-vendored crates.  The entry
 * points below are therefore exactly what a `[patch.crates-io]` shim of those crates binds:
 *
 *   zk_msm / zk_msm_device      <-  ark-ec 0.3  msm/variable_base.rs  VariableBaseMSM::multi_scalar_mul
 *                                   halo2_proofs 0.2  arithmetic.rs   best_multiexp
 *   zk_ntt / zk_ntt_device      <-  ark-poly 0.3  domain/radix2/{mod,fft}.rs  Radix2EvaluationDomain::{fft,ifft}_in_place
 *                                   halo2_proofs 0.2  arithmetic.rs   best_fft
 *   zk_coset_mul[_device]       <-  ark-poly 0.3  Radix2EvaluationDomain::distribute_powers (coset_fft / coset_ifft)
 *                                   halo2_proofs 0.2  poly/domain.rs  EvaluationDomain::distribute_powers_zeta
 *   zk_bases_*                  <-  residency of ark-groth16 0.3 `ProvingKey` query vectors / halo2 `Params::g`
 *                                   (read by the reference at lib/src/utils.rs:104-110)
 *
 * INTEGRATION.md shows the Rust-side stubs.  Conventions:
 *   - plain pointers and sizes, little-endian u64 limbs (== little-endian u32 words);
 *   - field elements in Montgomery form, R = 2^256 (2^384 for BLS12-381 Fq), unless stated;
 *   - affine points are (x, y); the point at infinity is encoded as x = y = 0; on the G2 curves a
 *     coordinate is an Fq2 element stored c0 then c1 (ark-ff 0.3 Fp2 { c0, c1 });
 *   - results are Jacobian (X, Y, Z), identity has Z = 0 -- the in-memory form of ark-ec 0.3
 *     `GroupProjective` and pasta_curves 0.4 `Ep`/`Eq`;
 *   - every function returns 0 on success or a negative zk_status; nothing throws or aborts;
 *   - thread-safe: any host thread may call any entry point (each call binds the calling thread to the device it
 *     uses; host-side enqueue is serialised per device).  Asynchronous (*_device) calls on DIFFERENT streams do
 *     not share scratch: every stream gets its own NTT ping-pong buffer / fixed-base table, every MSM in flight
 *     its own workspaces;
 *   - multi-GPU, two ways: (a) one process drives n GPUs (zk_init_devices): bases are uploaded to every device
 *     and zk_msm / zk_msm_device split the scalar windows over them and add the partial sums on the host;
 *     (b) one process per GPU (zk_init) with zk_msm_opts.window_begin/window_end and the caller's own collective;
 *   - "device" pointers must be 16-byte aligned HBM addresses.
 *   - There is no CPU fallback: without a usable MI355X every compute entry point fails with
 *     ZK_ERR_NO_DEVICE.
 */
#ifndef ZKCP_AMD_H
#define ZKCP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ZK_PALLAS = 0,       /* y^2 = x^3 + 5 over Fp, scalars in Fq   (pasta_curves 0.4 `pallas`) */
    ZK_VESTA = 1,        /* y^2 = x^3 + 5 over Fq, scalars in Fp   (pasta_curves 0.4 `vesta`; halo2 commitments over pallas::Base) */
    ZK_BN254_G1 = 2,     /* ark-bn254 0.3 G1 */
    ZK_BLS12_381_G1 = 3, /* ark-bls12-381 0.3 G1 -- the curve the reference proves on (lib/src/lib.rs:21-24) */
    ZK_BN254_G2 = 4,     /* ark-bn254 0.3 G2: y^2 = x^3 + 3/(9+u) over Fq2 = Fq[u]/(u^2+1); a coordinate is (c0, c1) */
    ZK_BLS12_381_G2 = 5  /* ark-bls12-381 0.3 G2: y^2 = x^3 + 4(1+u) over Fq2 -- Groth16's b_g2_query MSM */
} zk_curve_t;

typedef enum {
    ZK_FP_PALLAS = 0,    /* pasta Fp  (Pallas base = Vesta scalar) */
    ZK_FQ_PALLAS = 1,    /* pasta Fq  (Pallas scalar = Vesta base) */
    ZK_FR_BN254 = 2,
    ZK_FR_BLS12_381 = 3
} zk_field_t;

typedef enum {
    ZK_OK = 0,
    ZK_ERR_INVALID_ARG = -1,
    ZK_ERR_NOT_INITIALIZED = -2,
    ZK_ERR_NO_DEVICE = -3,
    ZK_ERR_HIP = -4,
    ZK_ERR_OOM = -5,
    ZK_ERR_UNSUPPORTED = -6,
    ZK_ERR_BAD_HANDLE = -7,
    ZK_ERR_BUSY = -8          /* 4 MSMs are already in flight on the device: collect one first */
} zk_status;

/* MSM tuning / sharding.  Zero-initialise for defaults. */
typedef struct {
    int window_bits;    /* c; 0 = choose from n */
    int window_begin;   /* this call sums windows [window_begin, window_end) of the signed-digit      */
    int window_end;     /*   decomposition, already weighted by 2^(c*w); 0,0 = all windows           */
    int limb_bits;      /* bucket arithmetic: 0 = default (lazy 29/28-bit limbs), 32 = saturated 32-bit words (A/B, tests) */
    int split_log_plus1;/* 0 = automatic; k + 1 cuts every bucket's entry list into 2^k pieces (k <= 4) */
    int slice_len;      /* ZK_MSM_FLAG_SLICE_REDUCE only: buckets per lane of the slice reduction; 0 = automatic */
    int big_threshold;  /* buckets longer than this take the cooperative segment path; 0 = 2 x mean + 64 */
    int waves_per_simd; /* accumulate-kernel waves launched per SIMD; 0 = what the kernel was compiled for */
    int flags;          /* ZK_MSM_FLAG_* */
    int base_offset;    /* use bases [base_offset, base_offset + n) of the handle (halo2's IPA works on halves of one generator vector) */
    int window_group;   /* windows processed at a time by a single MSM (so that sorted entries + bases stay inside the 256 MiB Infinity Cache);
                         * 0 = all at once, which measured faster at every size tried (profiles/r03_e_window_group_sweep.txt): A/B knob */
    int reserved;
} zk_msm_opts;
#define ZK_MSM_FLAG_NO_HOT_HELP 1   /* skewed witnesses: leave hot regions to their own sort workgroup */
#define ZK_MSM_FLAG_SLICE_REDUCE 2  /* bucket reduction by slices + multiplier (round 1) instead of row / column sums: A/B */
#define ZK_MSM_FLAG_PRECOMPUTED 4   /* ONE bucket set over the handle's table of window multiples (zk_bases_precompute first): no per-window
                                     * reduction, no host Horner; whole MSMs over the whole handle only */
#define ZK_MSM_FLAG_OWN_STREAM 16    /* zk_msm_submit only: run the job on a library stream forked behind the work `hip_stream` holds at the call (it sees
                                     * the scalars that work produces), alternating between two streams, so that MSMs submitted back to back overlap --
                                     * the G2 MSM of a Groth16 proof spends ~4 ms in dependent additions that leave the GPU idle.  Nothing enqueued on
                                     * `hip_stream` afterwards waits for the job: the scalars must stay untouched until zk_msm_collect. */
#define ZK_MSM_FLAG_DEVICE_PARTIALS 8 /* diagnostic: the lazy-limb partial sums are converted to the caller's limb form on the device, the result
                                       * is built from those, and every one is checked against the host's conversion: zk_msm_profile.reserved =
                                       * mismatches (low 24 bits) | first differing conversion stage << 24 | its component << 28 */

/* NTT plan knobs (process-wide, zk_ntt_configure).  Zero-initialise for defaults. */
typedef struct {
    int max_log_radix;   /* bits per pass, 1..10; 0 = 10 */
    int log_tile_plus1;  /* 0 = automatic (log2 T = 2); k + 1 forces log2 T = k (k <= 4) */
    int block;           /* lanes per tile workgroup: 64..1024; 0 = 512 */
    int limb_bits;       /* butterflies inside a tile: 0 = lazy 9 x 29-bit limbs (default), 32 = saturated 32-bit words (A/B, tests) */
} zk_ntt_opts;

/* Wall-clock of the phases of the last zk_msm* call on this process (milliseconds, HIP events).
 * The sort has three phases: digits_ms = digit counts per (scalar block, window, bucket range) + their scans,
 * hist_ms = staging the digits range by range, scatter_ms = the per-range LDS histogram / counting-sort scatter. */
typedef struct {
    float digits_ms, hist_ms, scatter_ms, accumulate_ms, reduce_ms, host_tail_ms, total_ms;
    int window_bits, windows_total, windows_done;
    int groups;      /* window groups the job ran in (zk_msm_opts.window_group): with more than one, the per-phase times are those of the last group; total_ms and accumulate_kernel_ms cover all */
    int limb_bits;   /* bucket arithmetic of the call: 29 = lazy unsaturated limbs (9 x 29 bits; BLS12-381 14 x 28; pairs of those on G2), 32 = saturated words */
    float accumulate_kernel_ms;   /* msm_accumulate_kernel alone (accumulate_ms also covers the piece / segment combine kernels) */
    int reserved;    /* ZK_MSM_FLAG_DEVICE_PARTIALS: see there; 0 otherwise */
} zk_msm_profile;

/* Sums over every MSM collected since the last reset (HIP events on the launch streams): what a bench needs when MSMs
 * run in batches / in flight.  algorithmic_bytes = n x (32 + affine point bytes) x (windows done / windows) per MSM. */
typedef struct {
    uint64_t msms;
    double accumulate_kernel_ms, accumulate_ms, sort_ms, reduce_ms, host_tail_ms, device_ms;
    double algorithmic_bytes;
    uint64_t launches;   /* accumulate-kernel launches: zk_msm_batch_device sums up to 4 scalar vectors per launch sequence */
} zk_msm_totals;

/* NTT pass-kernel timing (off by default): when enabled, every ntt_pass_kernel launch is bracketed by HIP events on its
 * stream; zk_ntt_profile_read waits for them, sums and resets.  algorithmic_bytes = 32 B x (elements read + written) per
 * transform, whatever the number of passes. */
typedef struct {
    uint64_t transforms, launches;
    double kernel_ms;
    double algorithmic_bytes;
} zk_ntt_totals;

/* ---- lifecycle ---- */
int zk_init(int device_id);            /* this process drives one GPU; idempotent for the same id */
/* this process drives n GPUs: per device a context, library streams and workspaces.  Device 0 of the list is
 * the "home" device of NTTs and of device pointers whose owner cannot be determined. */
int zk_init_devices(int n_devices, const int *device_ids);
int zk_device_count(void);             /* devices this process drives (0 before zk_init*) */
int zk_shutdown(void);                 /* frees every device allocation made by the library */
const char *zk_strerror(int status);
int zk_backend_info(char *buf, uint64_t buflen); /* e.g. "hip gfx950 AMD Instinct MI355X cu=256" */

/* ---- sizes ---- */
int zk_field_limbs64(zk_field_t f);            /* u64 limbs per scalar-field element (4) */
int zk_curve_base_limbs64(zk_curve_t c);       /* u64 limbs per point coordinate: Fq 4 (6 for BLS12-381); Fq2 on the G2 curves 8 (12) */
int zk_curve_scalar_field(zk_curve_t c);       /* zk_field_t of the curve's scalars */
int zk_msm_window_bits(zk_curve_t c, uint64_t n, int requested);            /* c actually used */
int zk_msm_window_count(zk_curve_t c, uint64_t n, int window_bits);         /* ceil((bits+1)/c) */

/* ---- SRS residency: bases are fixed per circuit; upload once, reuse for every proof ---- */
int zk_bases_upload(zk_curve_t c, const void *affine_xy_mont_host, uint64_t n, uint64_t *handle_out);
int zk_bases_adopt_device(zk_curve_t c, const void *affine_xy_mont_dev, uint64_t n, uint64_t *handle_out); /* no copy; caller keeps it alive */
int zk_bases_free(uint64_t handle);
/* The caller has rewritten points [offset, offset + count) of an ADOPTED device buffer (on `hip_stream`): bring the library's
 * derived copies (the lazy-limb form; the peers' copies on a multi-device process) up to date, ordered after that stream's
 * work.  halo2's IPA folds its generator vector in place every round (zk_ipa_fold_bases_device). */
int zk_bases_refresh(uint64_t handle, uint64_t offset, uint64_t count, void *hip_stream);
/* [2^(c w)] P_i for every window w of the MSM plan at the handle's size (16 x the points at 2^20: 1 GiB for a G1 key), for
 * ZK_MSM_FLAG_PRECOMPUTED.  One-time cost per resident key (one inversion per table entry). */
int zk_bases_precompute(uint64_t handle, int window_bits);

/* ---- MSM: out = sum_i scalars[i] * bases[i], i < n <= bases length ----
 * scalars: n x 4 u64.  scalars_are_montgomery = 0 for ark-ec (canonical BigInt from into_repr()),
 * 1 for halo2 (Fp/Fq as stored).  out: Jacobian (X, Y, Z), 3 x base limbs, host memory. */
int zk_msm(zk_curve_t c, uint64_t bases_handle, const void *scalars_host, uint64_t n,
           int scalars_are_montgomery, const zk_msm_opts *opts, void *out_jacobian_host);
int zk_msm_device(zk_curve_t c, uint64_t bases_handle, const void *scalars_dev, uint64_t n,
                  int scalars_are_montgomery, const zk_msm_opts *opts, void *out_jacobian_host,
                  void *hip_stream);
int zk_msm_last_profile(zk_msm_profile *out);   /* of the MSM collected last */
int zk_msm_profile_totals(zk_msm_totals *out, int reset);

/* Deferred result: zk_msm_submit enqueues all device work of one MSM on `hip_stream` and returns a ticket without
 * synchronising; zk_msm_collect waits for that MSM's last event, finishes it on the host (Horner over the <= 16
 * window sums) and writes the Jacobian result.  A prover issues its MSMs back to back (5 per Groth16 proof, one per
 * committed column in halo2): submitting MSM k+1 before collecting MSM k lets the GPU run k+1's sort beside k's
 * latency-bound bucket reduction and hides the host tail.  At most 4 MSMs in flight per device; every ticket must be
 * collected exactly once.  Scalars must stay valid and unchanged until the ticket is collected. */
int zk_msm_submit(zk_curve_t c, uint64_t bases_handle, const void *scalars_dev, uint64_t n, int scalars_are_montgomery,
                  const zk_msm_opts *opts, void *hip_stream, uint64_t *ticket_out);
int zk_msm_collect(uint64_t ticket, void *out_jacobian_host);

/* `count` MSMs over the SAME bases: scalars_dev holds count vectors of n scalars, vector k at element offset
 * k * stride_elems (stride_elems >= n); out_jacobian_host receives count results.  halo2 0.2 create_proof commits all
 * advice / lookup / permutation columns against one `Params::g_lagrange`; Groth16's a_query and b_g1_query MSMs share
 * the assignment instead (different bases: use submit/collect for those).  Up to four vectors are summed by ONE launch
 * sequence (their windows are more windows of the same sort / accumulate / reduce kernels); the sequences alternate between
 * two library streams forked from `hip_stream` and joined back into it. */
int zk_msm_batch_device(zk_curve_t c, uint64_t bases_handle, const void *scalars_dev, uint64_t n, uint32_t count,
                        uint64_t stride_elems, int scalars_are_montgomery, const zk_msm_opts *opts,
                        void *out_jacobian_host, void *hip_stream);

/* ---- NTT: in-place radix-2 DFT of size 2^log_n, natural order in and out ----
 * a[k] <- sum_j a[j] * omega^(jk); the caller passes omega (halo2 best_fft semantics: omega or
 * omega^-1, no implicit scaling).  scale_by_n_inv = 1 additionally multiplies by (2^log_n)^-1
 * (ark-poly ifft_in_place semantics when omega = group_gen_inv).  On the device entry points the argument is a bit set:
 * bit 0 = that scaling; ZK_NTT_OUT_R29 = write the results as x R' mod p, R' = 2^261 (the lazy-limb radix), for a
 * consumer that computes on lazy limbs (zk_expr_eval_lazy_device) -- one constant changes in the last pass, no extra work;
 * ZK_NTT_OUT_SUBCOSETS(lp): result k is stored at (k mod P) * (n / P) + k / P, P = 2^lp -- the P sub-cosets of the output domain
 * one after another (sub-coset j = the points g omega^(i P + j)), which is how a prover that evaluates its quotient sub-coset by
 * sub-coset wants a column (halo2.py EvaluationDomain.coeff_to_extended(parts=)); only the last pass's store addresses change. */
#define ZK_NTT_OUT_R29 2
#define ZK_NTT_OUT_SUBCOSETS(log_parts) (((log_parts) & 15) << 4)
int zk_ntt(zk_field_t f, void *a_mont_host, uint32_t log_n, const void *omega_mont_host, int scale_by_n_inv);
int zk_ntt_device(zk_field_t f, void *a_mont_dev, uint32_t log_n, const void *omega_mont_host,
                  int scale_by_n_inv, void *hip_stream);

int zk_ntt_configure(const zk_ntt_opts *opts);   /* NULL restores the defaults */
int zk_ntt_profile_enable(int on);
int zk_ntt_profile_read(zk_ntt_totals *out);

/* The NTT fused with the coset shifts around it (either pointer may be NULL):
 *   a[i] *= g_pre^i  ->  DFT (root omega, optional 1/n scaling)  ->  a[k] *= g_post^k
 * ark-poly 0.3: coset_fft_in_place  = (g_pre = F::multiplicative_generator(), omega = group_gen);
 *               coset_ifft_in_place = (omega = group_gen_inv, scale = 1, g_post = generator^-1).
 * halo2 0.2:    EvaluationDomain::coeff_to_extended / extended_to_coeff (zeta shifts around best_fft).
 * The powers are formed on the fly from two small cached tables (g^j, j < 1024; g^(1024 j)), not read from an n-entry table. */
int zk_ntt_coset_device(zk_field_t f, void *a_mont_dev, uint32_t log_n, const void *omega_mont_host, int scale_by_n_inv,
                        const void *g_pre_mont_host, const void *g_post_mont_host, void *hip_stream);

/* The same transform on a zero-extended input: only a[0 .. 2^log_in) is read, a[2^log_in .. 2^log_n) counts as zero and is
 * overwritten with the result -- halo2 0.2 EvaluationDomain::coeff_to_extended (`a.resize(extended_len(), 0)` ; zeta
 * shift ; best_fft on the extended domain), without storing or re-reading the padding.  log_in <= log_n. */
int zk_ntt_extend_device(zk_field_t f, void *a_mont_dev, uint32_t log_n, uint32_t log_in, const void *omega_mont_host,
                         int scale_by_n_inv, const void *g_pre_mont_host, const void *g_post_mont_host, void *hip_stream);

/* ... out of place: the input src[0 .. 2^log_in) is read by the first pass and left untouched, the 2^log_n results land in dst
 * (src == dst is the in-place form above; otherwise the buffers must not overlap).  halo2 keeps BOTH forms of every column
 * (Lagrange values -> coefficients for the openings -> extended coset for the quotient): no copy in between. */
int zk_ntt_oop_device(zk_field_t f, const void *src_mont_dev, void *dst_mont_dev, uint32_t log_n, uint32_t log_in, const void *omega_mont_host,
                      int scale_by_n_inv, const void *g_pre_mont_host, const void *g_post_mont_host, void *hip_stream);

/* a[i] *= g^i, i < 2^log_n */
int zk_coset_mul(zk_field_t f, void *a_mont_host, uint32_t log_n, const void *g_mont_host);
int zk_coset_mul_device(zk_field_t f, void *a_mont_dev, uint32_t log_n, const void *g_mont_host, void *hip_stream);

/* ---- Groth16 witness map glue (ark-groth16 0.3 r1cs_to_qap.rs R1CStoQAP::witness_map), device buffers ----
 * zk_vec_op_device: op 0 a*=b, 1 a-=b, 2 a+=b, 3 a*=scalar, 4 a=into_repr(a) (Montgomery -> canonical BigInt),
 *                   5 a=from_repr(a), 6 a=(a*b-c)*scalar.  b / c / scalar may be NULL when the op does not use them.
 * zk_groth16_witness_map_device: a, b, c hold the evaluations <A_i,z>, <B_i,z>, <C_i,z> on the size-2^log_m domain;
 *   runs ifft, coset_fft (x3), ab - c, division by Z_H on the coset and coset_ifft entirely in HBM; on return `a`
 *   holds the m coefficients of h (Montgomery), b and c are clobbered.  Feed a[0..m-1) to zk_msm_device with
 *   scalars_are_montgomery = 1 for the h_query MSM. */
int zk_vec_op_device(zk_field_t f, int op, void *a_dev, const void *b_dev, const void *c_dev, uint64_t n,
                     const void *scalar_mont_host, void *hip_stream);
int zk_groth16_witness_map_device(zk_field_t f, void *a_dev, void *b_dev, void *c_dev, uint32_t log_m, void *hip_stream);
/* a[i] *= table[i mod m], m a power of two <= 16, table in host memory (Montgomery):
 * halo2_proofs 0.2 poly/domain.rs EvaluationDomain::divide_by_vanishing_poly (t_evaluations). */
int zk_vec_scale_periodic_device(zk_field_t f, void *a_dev, uint64_t n, const void *table_mont_host, uint32_t m, void *hip_stream);

/* ---- host-side helpers a shim needs around the two kernels ---- */
int zk_field_modulus(zk_field_t f, void *p_canonical_out);                          /* 4 x u64, little-endian */
int zk_field_root_of_unity(zk_field_t f, uint32_t log_n, void *omega_mont_out);   /* ark group_gen / halo2 omega for size 2^log_n */
int zk_field_multiplicative_generator(zk_field_t f, void *g_mont_out);            /* ark coset shift (7 BLS12-381 Fr, 5 BN254 Fr, 5 pasta) */
int zk_field_inverse(zk_field_t f, const void *a_mont, void *out_mont);
int zk_point_add(zk_curve_t c, const void *jac_a, const void *jac_b, void *jac_out);  /* combine per-GPU partial sums */
int zk_point_to_affine(zk_curve_t c, const void *jac, void *affine_out);               /* identity -> (0, 0) */

/* out[i] = [k_i] G (affine, Montgomery) for canonical scalars k_i; device buffers.
 * Seeded synthetic SRS for tests/benches and the kernel under fixed-base setup work. */
int zk_fixed_base_mul_device(zk_curve_t c, const void *scalars_canonical_dev, uint64_t n, void *affine_out_dev,
                             void *hip_stream);

/* out[i] = [k_i] B (affine, Montgomery; identity -> (0, 0)) for ONE base B: replaces, for Groth16 key generation
 * (ark-groth16 0.3 generate_parameters, reached from the reference at lib/src/zk/encryption.rs:169), the upstream sequence
 *   ark_ec::msm::FixedBaseMSM::get_window_table + FixedBaseMSM::multi_scalar_mul   (ark-ec 0.3 src/msm/fixed_base.rs)
 *   ProjectiveCurve::batch_normalization_into_affine                                  (ark-ec 0.3 src/lib.rs)
 * base_affine_mont: host pointer to (x, y) in Montgomery form, or NULL for the curve's generator.  Scalars and output are
 * device buffers; the window table (8-bit windows) is built on the device per call. */
int zk_fixed_base_msm_device(zk_curve_t c, const void *base_affine_mont, const void *scalars_dev, uint64_t n,
                             int scalars_are_montgomery, void *affine_out_dev, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* ZKCP_AMD_H */
