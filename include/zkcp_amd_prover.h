/*
 * zkcp_amd_prover -- the callers and data formats either side of the MSM / NTT path:
 * what a patched ark-groth16 / halo2_proofs needs beyond zkcp_amd.h to keep a whole proof on the device and to exchange
 * keys and proofs with the unmodified reference.
 *
 *   zk_ark_*                   <-  ark-serialize 0.3 CanonicalSerialize / CanonicalDeserialize of ark-ec 0.3 GroupAffine,
 *                                  Vec<T>, ark-groth16 0.3 ProvingKey / VerifyingKey / Proof -- the formats the reference
 *                                  writes and reads at lib/src/utils.rs:85-118 (serialize_unchecked / deserialize_unchecked
 *                                  of the proving key, ark_to_bytes / ark_from_bytes of the verifying key) and
 *                                  circuits-ark/src/utils.rs:12-22 (ark_to_bytes(proof) at lib/src/zk/encryption.rs:80).
 *
 * Same conventions as zkcp_amd.h (which this header includes): plain pointers and sizes, Montgomery field elements as
 * little-endian u64 limbs, affine points (x, y) with infinity = (0, 0), negative zk_status on error.
 */
#ifndef ZKCP_AMD_PROVER_H
#define ZKCP_AMD_PROVER_H

#include "zkcp_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ZK_PAIRING_BN254 = 0,       /* ark-bn254 0.3: G1 = ZK_BN254_G1, G2 = ZK_BN254_G2 */
    ZK_PAIRING_BLS12_381 = 1    /* ark-bls12-381 0.3 -- the reference's PairingEngine (lib/src/lib.rs:21-24) */
} zk_pairing_t;

/* ---- ark-serialize 0.3 wire formats (host memory on both sides) ----
 * A point is serialized as x (and y when uncompressed) in canonical, non-Montgomery little-endian bytes with the two
 * SWFlags bits in the top of the last byte: bit 7 = y is the larger of (y, -y), bit 6 = infinity.
 *   compressed = 1: CanonicalSerialize::serialize            (48 B G1 / 96 B G2 on BLS12-381; 32 / 64 on BN254)
 *   compressed = 0: serialize_uncompressed = serialize_unchecked (96 / 192; 64 / 128); infinity is (0, 1) + flag
 * Decoding uncompressed points follows deserialize_unchecked (no curve / subgroup check) unless check_on_curve is set;
 * decoding compressed points recovers y (p = 3 mod 4 square roots; complex method on Fq2) and fails with
 * ZK_ERR_INVALID_ARG on a non-canonical coordinate, an invalid flag pair or an x that is not on the curve. */
int zk_ark_point_size(zk_curve_t c, int compressed);
int zk_ark_points_encode(zk_curve_t c, const void *affine_mont, uint64_t n, int compressed, uint8_t *out);
int zk_ark_points_decode(zk_curve_t c, const uint8_t *in, uint64_t n, int compressed, int check_on_curve, void *affine_mont_out);

/* Fr elements: canonical little-endian 32 bytes (ark-ff 0.3 Fp256::serialize) <-> Montgomery limbs */
int zk_ark_scalars_encode(zk_field_t f, const void *mont, uint64_t n, uint8_t *out);
int zk_ark_scalars_decode(zk_field_t f, const uint8_t *in, uint64_t n, void *mont_out);

/* Layout of ProvingKey::<E>::serialize_unchecked (ark-groth16 0.3; field order vk { alpha_g1, beta_g2, gamma_g2, delta_g2,
 * gamma_abc_g1 }, beta_g1, delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query; a Vec is a u64 length followed by
 * its items): byte offset of the first point and number of points of every member, so that a key file the reference's
 * `compile` wrote (lib/src/utils.rs:85-102) can be uploaded member by member without an intermediate copy. */
typedef struct {
    uint64_t offset, count;
} zk_ark_span;
typedef struct {
    zk_ark_span alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1, beta_g1, delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query;
    uint64_t total_bytes;
} zk_ark_pk_index;
int zk_ark_proving_key_index(zk_pairing_t p, const uint8_t *buf, uint64_t len, zk_ark_pk_index *out);

/* decode n uncompressed points (a query vector located with zk_ark_proving_key_index) into a resident bases handle */
int zk_bases_upload_ark(zk_curve_t c, const uint8_t *uncompressed_points, uint64_t n, uint64_t *handle_out);

/* ark-groth16 0.3 Proof { a: G1, b: G2, c: G1 }, compressed: 192 bytes on BLS12-381 (128 on BN254) -- what the reference
 * stores as `proof_of_encryption` (lib/src/zk/verifiable_encryption.rs:23-27) */
int zk_ark_proof_size(zk_pairing_t p);
int zk_ark_proof_encode(zk_pairing_t p, const void *a_g1_affine_mont, const void *b_g2_affine_mont, const void *c_g1_affine_mont, uint8_t *out);
int zk_ark_proof_decode(zk_pairing_t p, const uint8_t *in, void *a_g1_affine_mont, void *b_g2_affine_mont, void *c_g1_affine_mont);

/* ---- Groth16 around the MSM / NTT path (ark-groth16 0.3 create_proof; call sites
 * lib/src/zk/encryption.rs:76, verifiable_encryption.rs:92, sample_entries.rs:86, property.rs:133) ----
 *
 * R1CS matrices (ark-relations 0.3 ConstraintMatrices { a, b, c }: one Vec<(coeff, variable index)> per constraint) are
 * fixed per circuit like the proving key: uploaded once in CSR form (row_ptr: n_rows + 1 offsets; col_idx / val per term,
 * val in Montgomery form), resident on the home device. */
int zk_r1cs_matrix_upload(zk_field_t f, const uint64_t *row_ptr_host, const uint32_t *col_idx_host, const void *val_mont_host,
                          uint64_t n_rows, uint64_t n_cols, uint64_t *handle_out);
int zk_r1cs_matrix_free(uint64_t handle);
/* out[i] = <row i, z> for i < n_rows, 0 for n_rows <= i < out_len  (upstream: evaluate_constraint per row) */
int zk_r1cs_matvec_device(uint64_t matrix, const void *z_mont_dev, void *out_mont_dev, uint64_t out_len, void *hip_stream);
/* R1CStoQAP::witness_map from the full assignment z (instance variables first, z[0] = 1), all in HBM:
 *   a = A z, b = B z, c = C z on the first num_constraints rows; a[num_constraints + j] = z[j] for j < num_inputs;
 *   then the seven NTTs and the pointwise glue of zk_groth16_witness_map_device.  a_dev / b_dev / c_dev are caller
 *   buffers of 2^log_m elements; on return a_dev holds h.  All three matrices must have the same number of rows
 *   and num_constraints + num_inputs <= 2^log_m. */
int zk_groth16_witness_map_r1cs_device(zk_field_t f, uint64_t matrix_a, uint64_t matrix_b, uint64_t matrix_c, const void *z_mont_dev,
                                       uint64_t num_inputs, uint32_t log_m, void *a_dev, void *b_dev, void *c_dev, void *hip_stream);

/* The last step of create_proof: the proof from the five MSM results, the key's single elements and the blinding r, s.
 *   A = a_query[0] + a_acc + alpha_g1 + r delta_g1
 *   B = b_g2_query[0] + b_g2_acc + beta_g2 + s delta_g2          (B1 likewise in G1, used for C only)
 *   C = s A + r B1 - r s delta_g1 + l_acc + h_acc
 * Points of the key are affine (x, y) Montgomery; the *_acc are the Jacobian outputs of zk_msm* over query[1..] ; r, s are
 * Fr elements in Montgomery form.  Host arithmetic (a handful of scalar multiplications), like upstream. */
typedef struct {
    const void *alpha_g1, *beta_g1, *delta_g1;
    const void *beta_g2, *delta_g2;
    const void *a_query0, *b_g1_query0, *b_g2_query0;
    const void *a_acc, *b_g1_acc, *l_acc, *h_acc;
    const void *b_g2_acc;
    const void *r, *s;
} zk_groth16_assembly;
int zk_groth16_assemble_proof(zk_pairing_t p, const zk_groth16_assembly *in, void *a_g1_affine_out, void *b_g2_affine_out,
                              void *c_g1_affine_out);

/* ---- halo2_proofs 0.2 prover steps beyond commit / FFT, device buffers, Montgomery elements ----
 * The reference's circuit (circuits-halo2/src/encryption.rs:83-161: 13 advice + 8 fixed columns, a lookup table, a
 * permutation over the equality-enabled columns) is the shape donor; these are the per-row products a create_proof over
 * it runs between the column commitments and the opening.  Blinding rows / scalars and the transcript stay with the
 * caller (RNG and hashing on the CPU), like upstream's structure. */

/* arithmetic.rs BatchInvert: a[i] <- 1 / a[i], zeros stay zero */
int zk_batch_invert_device(zk_field_t f, void *a_dev, uint64_t n, void *hip_stream);
/* out[i] = first * prod_{j < i} in[j] (in == out allowed); first NULL = 1; total_out_host (optional, synchronises) =
 * first * prod of all -- the running value the next chunk of a permutation argument starts from */
int zk_prefix_product_device(zk_field_t f, const void *in_dev, void *out_dev, uint64_t n, const void *first_mont_host,
                             void *total_out_mont_host, void *hip_stream);
/* plonk/permutation/prover.rs Argument::commit, one chunk of <= 8 columns (chunk_len = degree - 2) on the 2^k-row domain:
 *   Z(0) = z_first (NULL = 1);  Z(i + 1) = Z(i) prod_c (v_c(i) + beta delta^(first_column_index + c) omega^i + gamma)
 *                                              / prod_c (v_c(i) + beta sigma_c(i) + gamma)
 * columns_dev / sigmas_dev: host arrays of ncols device pointers (Lagrange form).  z_out_dev gets Z(0 .. 2^k);
 * z_last_out_host (optional, synchronises) the value after the last row. */
int zk_halo2_permutation_product_device(zk_field_t f, uint32_t ncols, const void *const *columns_dev, const void *const *sigmas_dev,
                                        uint32_t first_column_index, const void *beta, const void *gamma, const void *delta, uint32_t k,
                                        const void *z_first, void *z_out_dev, void *z_last_out_host, void *hip_stream);
/* plonk/lookup/prover.rs commit_product: Z(0) = 1, Z(i + 1) = Z(i) (A_i + beta)(S_i + gamma) / ((A'_i + beta)(S'_i + gamma));
 * A', S' are the permuted input / table expressions (permute_expression_pair: a sort, left to the caller) */
int zk_halo2_lookup_product_device(zk_field_t f, const void *a_dev, const void *s_dev, const void *a_perm_dev, const void *s_perm_dev,
                                   const void *beta, const void *gamma, uint64_t n, void *z_out_dev, void *z_last_out_host, void *hip_stream);
/* poly/commitment/prover.rs create_proof, the scalar side of one round: compute_inner_product, and the folds
 * p'[i] += u^-1 p'[i + half], b[i] += u b[i + half] as a[i] += c a[i + half] */
int zk_inner_product_device(zk_field_t f, const void *a_dev, const void *b_dev, uint64_t n, void *out_mont_host, void *hip_stream);
int zk_vec_fold_device(zk_field_t f, void *a_dev, uint64_t half, const void *c_mont_host, void *hip_stream);
/* poly/multiopen/prover.rs: a[i] = a[i] * s + b[i] -- one Horner step of folding the polynomials queried at the same point set with
 * powers of x_1 (and the per-set quotients with x_4), on resident coefficient vectors.  s: Montgomery, host. */
int zk_vec_muladd_device(zk_field_t f, void *a_dev, const void *b_dev, uint64_t n, const void *s_mont_host, void *hip_stream);
/* ... out[i] = a[i] * s + b[i] into a third buffer (out == a or out == b allowed; otherwise disjoint): the first step of such a fold starts
 * from two resident polynomials and must clobber neither (upstream clones; this saves the copy) */
int zk_vec_muladd_to_device(zk_field_t f, void *out_dev, const void *a_dev, const void *b_dev, uint64_t n, const void *s_mont_host, void *hip_stream);
/* arithmetic.rs eval_polynomial: p(x) = sum_i coeffs[i] x^i for a resident coefficient vector (the evaluations create_proof
 * writes to the transcript: every committed polynomial at x and at its rotations omega^r x).  x, the result: Montgomery, host. */
int zk_poly_eval_device(zk_field_t f, const void *coeffs_dev, uint64_t n, const void *x_mont_host, void *out_mont_host, void *hip_stream);
/* ... `count` polynomials of n coefficients (polynomial q at element offset q * stride_elems) at the same x: one launch, one copy */
int zk_poly_eval_batch_device(zk_field_t f, const void *coeffs_dev, uint64_t n, uint32_t count, uint64_t stride_elems, const void *x_mont_host,
                              void *out_mont_host, void *hip_stream);
/* out[j] = sum_{i < count} s^(count - 1 - i) src_i[j] with src_i = first_dev + i * stride_elems (stride may be negative: walk the
 * polynomials backwards): the Horner fold of `count` resident polynomials in one pass, each read once (the x_1 fold of a point set
 * in poly/multiopen/prover.rs; h(X) = sum_i x^(n i) h_i in plonk/vanishing/prover.rs with first = the LAST piece, stride = -n).
 * out_dev may be one of the sources or disjoint from all of them. */
int zk_vec_fold_many_device(zk_field_t f, void *out_dev, const void *first_dev, int64_t stride_elems, uint32_t count, uint64_t n,
                            const void *s_mont_host, void *hip_stream);
/* one round's three folds of the inner-product argument in one launch: p'[i] += u^-1 p'[i + half], b[i] += u b[i + half] for i < half,
 * and (w_dev != NULL, the fold-free form) W[idx] *= u where idx < m0 has bit `half` set.  u != 0. */
int zk_ipa_fold_round_device(zk_field_t f, void *p_dev, void *b_dev, void *w_dev, uint64_t half, uint64_t m0, const void *u_mont_host,
                             void *hip_stream);
/* out[i] = x^i, i < n: the vector b of poly/commitment/prover.rs create_proof (powers of x_3), built from per-call power tables of x */
int zk_vec_powers_device(zk_field_t f, void *out_dev, uint64_t n, const void *x_mont_host, void *hip_stream);
/* arithmetic.rs kate_division(a, x): the quotient of (a(X) - a(x)) / (X - x) as n coefficients (upstream returns n - 1 and
 * poly/multiopen/prover.rs resizes to n: q[n - 1] = 0); the multiopen argument divides every point set's folded polynomial by
 * (X - x_j) for each point of the set in turn, which leaves the quotient by the set's vanishing polynomial.  q_dev == a_dev
 * (in place) or disjoint.  A three-phase suffix scan (q[j] = a[j + 1] + x q[j + 1]), ~3 products per coefficient. */
int zk_kate_division_device(zk_field_t f, const void *a_dev, void *q_dev, uint64_t n, const void *x_mont_host, void *hip_stream);
/* ... and the generator side (parallel_generator_collapse): g[i] <- affine(g[i] + [u] g[i + half]), i < half; g holds
 * 2 * half affine points (x, y) Montgomery on the device, u an element of the curve's scalar field (Montgomery, host) */
int zk_ipa_fold_bases_device(zk_curve_t c, void *g_affine_dev, uint64_t half, const void *u_mont_host, void *hip_stream);

/* The same rounds WITHOUT folding the generators (a fold is one ~255-bit scalar multiplication per surviving point and round;
 * an MSM is ~16 bucket additions per point): with W[idx] the product of the challenges whose fold put idx in an upper half,
 *   L = MSM(G0, S_L), R = MSM(G0, S_R) over the ORIGINAL m0 generators (the resident SRS handle, one zk_msm_batch_device call)
 *   S_L[idx] = p'[i + cur/2] W[idx] for i = idx mod cur < cur/2 (else 0);  S_R[idx] = p'[i - cur/2] W[idx] for i >= cur/2 (else 0)
 * zk_ipa_virtual_scalars_device fills S_L, S_R (m0 elements each); after the round's challenge u (and the p', b folds)
 * zk_ipa_update_weights_device multiplies W[idx] by u where idx has bit cur/2 set.  W starts as all ones; the folded
 * generator at the end is MSM(G0, W) (the prover does not need it). */
int zk_ipa_virtual_scalars_device(zk_field_t f, const void *p_dev, const void *w_dev, uint64_t m0, uint64_t cur, void *sl_dev, void *sr_dev,
                                  void *hip_stream);
int zk_ipa_update_weights_device(zk_field_t f, void *w_dev, uint64_t m0, uint64_t bit, const void *u_mont_host, void *hip_stream);
/* One whole fold-free round in one call (zk_ipa_virtual_scalars_device + two zk_inner_product_device + zk_msm_batch_device without
 * a host round trip per value): s_dev = scratch for 2 * m0 scalars; lr_out_host = L, R as Jacobian points (2 x 3 coordinates);
 * v_out_mont_host = <p'_hi, b_lo>, <p'_lo, b_hi>. */
int zk_ipa_round_device(zk_curve_t c, uint64_t bases_handle, const void *p_dev, const void *b_dev, const void *w_dev, uint64_t m0,
                        uint64_t cur, void *s_dev, void *lr_out_host, void *v_out_mont_host, void *hip_stream);
/* ... and to leave the fold-free form after r of those rounds: the generators r calls of parallel_generator_collapse would
 * have produced, g_out[i] = sum_{t < m0 / cur} W[t cur] G0[t cur + i] for i < cur (affine (x, y) Montgomery, identity (0, 0)),
 * as cur multi-scalar multiplications that share their m0 / cur <= 4096 scalars.  The later rounds then run over g_out
 * (zk_bases_adopt_device) with fresh weights: a few full-size rounds cost a full-size MSM each, the many small ones do not.
 * Synchronises hip_stream: g_out is complete on return. */
int zk_ipa_collapse_device(zk_curve_t c, uint64_t bases_handle, const void *w_dev, uint64_t m0, uint64_t cur, void *g_out_affine_dev,
                           void *hip_stream);
/* ... outputs [first, first + count) only, written to g_out_range_dev[0 .. count): a rank's share when several GPUs split the step
 * (the shares are exchanged with one all_gather: contangle-zkcp_amd/halo2.py IpaProverVirtual.collapse) */
int zk_ipa_collapse_range_device(zk_curve_t c, uint64_t bases_handle, const void *w_dev, uint64_t m0, uint64_t cur, uint64_t first,
                                 uint64_t count, void *g_out_range_dev, void *hip_stream);

/* The quotient numerator: one stack program evaluated at every row of the extended domain (plonk/prover.rs: each gate's
 * Expression over advice / fixed / instance columns with rotations, folded with y).  A rotation by r rows is a shift of
 * r * rot_scale positions (rot_scale = 2^(extended_k - k)), cyclic.  The program must leave exactly one value; it is
 * validated on the host (operand indices, stack depth <= 8, <= 512 ops, <= 64 columns, <= 32 constants). */
typedef struct {
    uint8_t op;     /* ZK_EXPR_* */
    uint8_t pad;
    int16_t rot;    /* ZK_EXPR_COL: rotation in rows */
    uint32_t arg;   /* ZK_EXPR_COL: column index; ZK_EXPR_CONST / ZK_EXPR_SCALE: constant index */
} zk_expr_op;
#define ZK_EXPR_COL 0
#define ZK_EXPR_CONST 1
#define ZK_EXPR_ADD 2
#define ZK_EXPR_SUB 3
#define ZK_EXPR_MUL 4
#define ZK_EXPR_NEG 5
#define ZK_EXPR_SCALE 6
int zk_expr_eval_device(zk_field_t f, const zk_expr_op *program_host, uint32_t n_ops, const void *const *columns_dev, uint32_t n_columns,
                        const void *consts_mont_host, uint32_t n_consts, uint32_t log_n_ext, uint32_t rot_scale, void *out_dev,
                        void *hip_stream);

/* The same evaluation on lazy 29-bit limbs (one MAD per partial product, carry-free additions): the columns must hold x R' mod p
 * with R' = 2^261 -- what the NTT entry points write when their scale argument has ZK_NTT_OUT_R29 set (zkcp_amd.h), or
 * zk_vec_op_device(scale) by 2^5 for key material -- as canonical 256-bit words.  Constants come in the usual Montgomery form and
 * the OUTPUT is in the usual form too.  The host walks the program once to place the carry steps and pick the subtraction
 * biases (the bound discipline of csrc/zk_field29.h); same validation and limits as zk_expr_eval_device. */
int zk_expr_eval_lazy_device(zk_field_t f, const zk_expr_op *program_host, uint32_t n_ops, const void *const *columns_r29_dev, uint32_t n_columns,
                             const void *consts_mont_host, uint32_t n_consts, uint32_t log_n_ext, uint32_t rot_scale, void *out_dev,
                             void *hip_stream);

/* A gate expression is fixed per proving key: for evaluations of 2^16 rows and more zk_expr_eval_lazy_device writes the annotated
 * program out as straight-line HIP (the stack resolved at generation time: no interpreter, no LDS), compiles it once per
 * (program, device) with hiprtc -- ~10 s for the reference circuit's 268 operations; the headers it includes ship next to the
 * library (csrc/, or $ZKCP_AMD_CSRC) -- and runs that: about half the instructions per row.  Same operations, same carry steps,
 * same results.  jit_mode: 0 = that rule (default), 1 = always, 2 = never (the interpreter kernel).  If the specialised kernel
 * cannot be built the interpreter runs. */
int zk_expr_configure(int jit_mode);
/* The HIP source zk_expr_eval_lazy_device would compile for this program (no device needed: diagnostics, and the CPU test tier
 * cross-compiles it).  *len_out = its length; at most cap - 1 bytes and a terminating 0 are written to out (out may be NULL). */
int zk_expr_specialised_source(zk_field_t f, const zk_expr_op *program_host, uint32_t n_ops, uint32_t n_columns, uint32_t n_consts, char *out,
                               uint64_t cap, uint64_t *len_out);

#ifdef __cplusplus
}
#endif
#endif /* ZKCP_AMD_PROVER_H */
