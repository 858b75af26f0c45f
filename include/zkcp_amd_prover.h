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

#ifdef __cplusplus
}
#endif
#endif /* ZKCP_AMD_PROVER_H */
