//! REPLACES the MSM section of `create_proof_with_reduction_and_matrices` in ark-groth16 0.3.0 `src/prover.rs`: the five
//! MSMs are SUBMITTED back to back (deferred results) so that the latency-bound bucket reduction and the host tail of MSM k
//! overlap the sort of MSM k+1, then collected; the final combination with r, s is upstream's.  NOT COMPILED here.
use ark_ec::{AffineCurve, PairingEngine, ProjectiveCurve};
use ark_ff::{PrimeField, Zero};
use ark_serialize::{CanonicalDeserialize, CanonicalSerialize};
use zkcp_amd_sys as zk;

use crate::device::DeviceVec;
use crate::{r1cs_to_qap::WitnessMapOnDevice, Proof, ProvingKey};

fn submit<G: AffineCurve>(curve: i32, bases: &[G], scalars_dev: *const core::ffi::c_void, n: usize, montgomery: bool,
                          stream: *mut core::ffi::c_void) -> u64 {
    let handle = zk::SRS.get_or_upload(curve, bases.as_ptr() as usize, n, &|i, out: &mut Vec<u8>| bases[i].serialize_uncompressed(out).unwrap());
    let mut ticket = 0u64;
    let st = unsafe { zk::zk_msm_submit(curve, handle, scalars_dev, n as u64, montgomery as i32, core::ptr::null(), stream, &mut ticket) };
    zk::check(st, "zk_msm_submit").unwrap();
    ticket
}
fn collect<G: AffineCurve>(curve: i32, ticket: u64) -> G::Projective {
    let limbs = unsafe { zk::zk_curve_base_limbs64(curve) } as usize;
    let mut out = vec![0u64; 3 * limbs];
    zk::check(unsafe { zk::zk_msm_collect(ticket, out.as_mut_ptr() as _) }, "zk_msm_collect").unwrap();
    G::deserialize_unchecked(&zk::jacobian_to_ark_uncompressed(curve, &out)[..]).unwrap().into_projective()
}

/// h: witness map output resident on the device; `z_dev`: the full assignment as canonical BigInts (into_repr done by
/// zk_vec_op_device op 4 on the uploaded Fr values); `g1`, `g2`: library curve ids of E::G1Affine / E::G2Affine.
pub fn create_proof_msms<E: PairingEngine>(pk: &ProvingKey<E>, h: &WitnessMapOnDevice, z_dev: &DeviceVec, num_inputs: usize,
                                           r: E::Fr, s: E::Fr, input0: E::Fr, g1: i32, g2: i32, stream: *mut core::ffi::c_void) -> Proof<E> {
    let n = z_dev.len();                      // instance + witness variables, the constant-one wire first
    let aux_dev = z_dev.offset(num_inputs);   // witness part
    // at most 4 MSMs in flight per device: submit four, collect one, submit the fifth
    let t_h = submit(g1, &pk.h_query, h.h.ptr(), h.domain_size - 1, true, stream);
    let t_l = submit(g1, &pk.l_query, aux_dev, n - num_inputs, false, stream);
    let t_a = submit(g1, &pk.a_query[1..], z_dev.offset(1), n - 1, false, stream);
    let t_b1 = submit(g1, &pk.b_g1_query[1..], z_dev.offset(1), n - 1, false, stream);
    let h_acc: E::G1Projective = collect::<E::G1Affine>(g1, t_h);
    let t_b2 = submit(g2, &pk.b_g2_query[1..], z_dev.offset(1), n - 1, false, stream);
    let l_aux_acc = collect::<E::G1Affine>(g1, t_l);
    let a_acc = collect::<E::G1Affine>(g1, t_a);
    let b1_acc = collect::<E::G1Affine>(g1, t_b1);
    let b2_acc = collect::<E::G2Affine>(g2, t_b2);
    // ---- upstream from here (calculate_coeff + assembly), unchanged in substance
    let delta_g1 = pk.delta_g1.into_projective();
    let mut g_a = pk.a_query[0].mul(input0.into_repr()) + a_acc + pk.vk.alpha_g1.into_projective() + delta_g1.mul(r.into_repr());
    let g1_b = pk.b_g1_query[0].mul(input0.into_repr()) + b1_acc + pk.beta_g1.into_projective() + delta_g1.mul(s.into_repr());
    let g2_b = pk.b_g2_query[0].mul(input0.into_repr()) + b2_acc + pk.vk.beta_g2.into_projective() + pk.vk.delta_g2.mul(s.into_repr());
    let mut g_c = g_a.mul(s.into_repr()) + g1_b.mul(r.into_repr()) - delta_g1.mul((r * s).into_repr());
    g_c += l_aux_acc + h_acc;
    let _ = E::Fr::zero();
    g_a = g_a;
    Proof { a: g_a.into_affine(), b: g2_b.into_affine(), c: g_c.into_affine() }
}
