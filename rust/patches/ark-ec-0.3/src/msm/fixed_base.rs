//! ADDS a batched entry next to ark-ec 0.3.0 `FixedBaseMSM` (src/msm/fixed_base.rs): key generation
//! (ark-groth16 0.3 generate_parameters, reached from the reference at lib/src/zk/encryption.rs:169 via `Groth16::setup`)
//! computes every query vector as  get_window_table + multi_scalar_mul + batch_normalization_into_affine  for ONE base;
//! the library does the three steps in one call.  NOT COMPILED here.
use ark_ff::PrimeField;
use ark_serialize::CanonicalSerialize;
use ark_std::vec::Vec;
use zkcp_amd_sys as zk;

use crate::{AffineCurve, ProjectiveCurve};

/// out[i] = [v[i]] g  as affine points.  `upload` / `download` are the caller's device-buffer plumbing (hipMalloc'd
/// scratch owned by the fork of ark-groth16, which keeps `v` resident across the five query vectors).
pub fn fixed_base_msm_affine<G: ProjectiveCurve>(curve: i32, g: G, v_dev: *const core::ffi::c_void, n: usize,
                                                 out_dev: *mut core::ffi::c_void, stream: *mut core::ffi::c_void) {
    zk::init_once();
    // the base goes over as Montgomery limbs of its affine form: decode our own uncompressed encoding on the library side
    let mut bytes = Vec::new();
    g.into_affine().serialize_uncompressed(&mut bytes).unwrap();
    let limbs = unsafe { zk::zk_curve_base_limbs64(curve) } as usize;
    let mut base = ark_std::vec![0u64; 2 * limbs];
    zk::check(unsafe { zk::zk_ark_points_decode(curve, bytes.as_ptr(), 1, 0, 0, base.as_mut_ptr() as _) }, "zk_ark_points_decode").unwrap();
    // `v` holds Fr values as stored (Montgomery): scalars_are_montgomery = 1
    let st = unsafe { zk::zk_fixed_base_msm_device(curve, base.as_ptr() as _, v_dev, n as u64, 1, out_dev, stream) };
    zk::check(st, "zk_fixed_base_msm_device").unwrap();
    let _ = core::marker::PhantomData::<<G::ScalarField as PrimeField>::BigInt>;
}
