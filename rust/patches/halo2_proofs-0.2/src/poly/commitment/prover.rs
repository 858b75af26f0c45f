//! REPLACES the round loop of `create_proof` in halo2_proofs 0.2.0 `src/poly/commitment/prover.rs` (the inner-product
//! argument that opens the combined polynomial; the reference's halo2 crate reaches it from `plonk::create_proof` through
//! `poly::multiopen::create_proof`).  Everything before the loop (the blinded `p_prime`, the powers `b` of x_3, the
//! challenge `z` and `u = params.u * z`) and everything after it (`c = p_prime[0]`, `f`) is upstream's.  NOT COMPILED here.
//!
//! Upstream halves three vectors per round and pays one 255-bit scalar multiplication per surviving generator
//! (`parallel_generator_collapse`).  Here `p_prime` and `b` live in device buffers and the generators are left alone for the
//! first `COLLAPSE_AFTER` rounds -- L_j and R_j are multi-scalar multiplications over the RESIDENT `params.g` with the
//! challenges folded into a weight vector (include/zkcp_amd_prover.h, zk_ipa_round_device) -- then materialised once
//! (zk_ipa_collapse_device) and the remaining rounds run over the 2^(k - COLLAPSE_AFTER) survivors.  L_j, R_j, the transcript
//! and the proof bytes are the same as upstream's.
use ff::Field;
use group::Curve;
use rand_core::RngCore;
use zkcp_amd_sys as zk;

use super::super::super::arithmetic::{best_multiexp, CurveAffine, FieldExt};
use super::super::super::transcript::{EncodedChallenge, TranscriptWrite};
use super::Params;

const COLLAPSE_AFTER: u32 = 6; // measured on one MI355X at k = 20: 4 .. 8 are within 10 %

/// The k rounds.  Returns (c, f): the last coefficient and the accumulated blinding, which upstream writes to the transcript.
#[allow(clippy::too_many_arguments)]
pub fn ipa_rounds_device<C: CurveAffine, E: EncodedChallenge<C>, R: RngCore, T: TranscriptWrite<C, E>>(
    params: &Params<C>, curve: i32, field: i32, srs_handle: u64, // params.g, uploaded once (arithmetic.rs: srs_handle)
    mut rng: R, transcript: &mut T, p_prime: &[C::Scalar], b: &[C::Scalar], z: C::Scalar, u: C, mut f: C::Scalar,
    stream: *mut core::ffi::c_void,
) -> std::io::Result<(C::Scalar, C::Scalar)> {
    let k = params.k;
    let mut m0 = 1u64 << k; // generators behind the current handle
    let mut cur = m0; // live length of p', b
    let mut handle = srs_handle;
    let mut owned: Option<(u64, zk::DeviceBuf)> = None; // the materialised generators and their handle
    let flat = |xs: &[C::Scalar]| -> Vec<u64> { xs.iter().flat_map(|x| limbs_of(x)).collect() };
    let one = limbs_of(&C::Scalar::one());
    let d_p = zk::DeviceBuf::upload(&flat(p_prime));
    let d_b = zk::DeviceBuf::upload(&flat(b));
    let mut d_w = zk::DeviceBuf::filled(m0 as usize, &one);
    let mut d_s = zk::DeviceBuf::zeroed(2 * 4 * m0 as usize);
    for j in 0..k {
        let half = cur / 2;
        // L_j = <p'_hi, G'_lo>, R_j = <p'_lo, G'_hi> and the two inner products with b: one call, no host round trip per value
        let (mut lr, mut v) = ([0u64; 24], [0u64; 8]);
        zk::check(unsafe { zk::zk_ipa_round_device(curve, handle, d_p.ptr(), d_b.ptr(), d_w.ptr(), m0, cur, d_s.ptr(), lr.as_mut_ptr() as _,
                                                   v.as_mut_ptr() as _, stream) }, "zk_ipa_round_device").unwrap();
        let (l_j, r_j) = (curve_from_jacobian_limbs::<C>(&lr[0..12]), curve_from_jacobian_limbs::<C>(&lr[12..24]));
        let (value_l_j, value_r_j) = (from_montgomery_limbs::<C::Scalar>(&v[0..4]), from_montgomery_limbs::<C::Scalar>(&v[4..8]));
        // the blinding terms and the transcript are upstream's, on the CPU
        let l_j_randomness = C::Scalar::random(&mut rng);
        let r_j_randomness = C::Scalar::random(&mut rng);
        let l_j = (l_j + &best_multiexp(&[value_l_j * &z, l_j_randomness], &[u, params.w])).to_affine();
        let r_j = (r_j + &best_multiexp(&[value_r_j * &z, r_j_randomness], &[u, params.w])).to_affine();
        transcript.write_point(l_j)?;
        transcript.write_point(r_j)?;
        let u_j = *transcript.squeeze_challenge_scalar::<()>();
        let u_j_inv = u_j.invert().unwrap();
        // p'[i] += u_j^-1 p'[i + half] ; b[i] += u_j b[i + half] ; the generators' fold goes into the weights
        let (ui, uu) = (limbs_of(&u_j_inv), limbs_of(&u_j));
        let _ = ui; // (the library inverts u_j itself: the three folds are one launch)
        zk::check(unsafe { zk::zk_ipa_fold_round_device(field, d_p.ptr(), d_b.ptr(), d_w.ptr(), half, m0, uu.as_ptr() as _, stream) }, "zk_ipa_fold_round_device").unwrap();
        cur = half;
        f += &(l_j_randomness * &u_j_inv);
        f += &(r_j_randomness * &u_j);
        if j + 1 == COLLAPSE_AFTER && j + 1 < k {
            // what COLLAPSE_AFTER calls of parallel_generator_collapse would have left in g_prime[..cur], in one step
            let g = zk::DeviceBuf::zeroed(8 * cur as usize); // cur affine points, (x, y) = 2 x 4 x u64 each
            zk::check(unsafe { zk::zk_ipa_collapse_device(curve, handle, d_w.ptr(), m0, cur, g.ptr(), stream) }, "zk_ipa_collapse_device").unwrap();
            let mut h = 0u64;
            zk::check(unsafe { zk::zk_bases_adopt_device(curve, g.ptr(), cur, &mut h) }, "zk_bases_adopt_device").unwrap();
            if let Some((old, _)) = owned.take() {
                unsafe { zk::zk_bases_free(old) };
            }
            owned = Some((h, g));
            handle = h;
            m0 = cur;
            d_w = zk::DeviceBuf::filled(m0 as usize, &one);
            d_s = zk::DeviceBuf::zeroed(2 * 4 * m0 as usize);
        }
    }
    if let Some((h, _)) = owned.take() {
        unsafe { zk::zk_bases_free(h) };
    }
    let mut c_limbs = [0u64; 4];
    d_p.download(0, &mut c_limbs);
    Ok((from_montgomery_limbs::<C::Scalar>(&c_limbs), f))
}

// shared with arithmetic.rs of this fork
use super::super::super::arithmetic::{curve_from_jacobian_limbs, from_montgomery_limbs, limbs_of};
