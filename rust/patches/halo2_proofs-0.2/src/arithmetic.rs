//! REPLACES `best_multiexp` and `best_fft` in halo2_proofs 0.2.0 `src/arithmetic.rs`.  The reference's halo2 crate never
//! reaches them (circuits-halo2/src/encryption.rs:335 only runs MockProver); a prover over that circuit would,
//! once per committed column and per domain change.  NOT COMPILED here.
use group::{ff::PrimeField, Group as _};
use pasta_curves::arithmetic::{CurveAffine, FieldExt};
use zkcp_amd_sys as zk;

use super::Group;

fn pasta_curve<C: CurveAffine>() -> Option<i32> {
    // Vesta's scalar field is Pallas's base field Fp (modulus low limb 0x992d30ed00000001); Pallas's is Fq
    let m = C::Scalar::MODULUS; // "0x4000...0001" hex string in pasta_curves 0.4
    if m.ends_with("992d30ed00000001") {
        Some(zk::ZK_VESTA)
    } else if m.ends_with("8c46eb2100000001") {
        Some(zk::ZK_PALLAS)
    } else {
        None
    }
}

pub fn best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve {
    assert_eq!(coeffs.len(), bases.len());
    let curve = match pasta_curve::<C>() {
        Some(c) if coeffs.len() >= 1 << 10 => c,
        _ => return cpu_best_multiexp(coeffs, bases), // upstream body (chunk-per-thread multiexp_serial), renamed
    };
    zk::init_once();
    // Params::g / g_lagrange are fixed per circuit: uploaded once.  pasta points have no ark encoding, so the bases go over
    // as Montgomery limbs of (x, y): `coordinates()` + `to_repr()` would give canonical bytes; the limb view is checked by
    // zk::layout_is_flat on the field type before it is used.
    let handle = srs_handle(curve, bases);
    // Fp / Fq are 4 x u64 Montgomery limbs in memory: scalars_are_montgomery = 1
    let one = C::Scalar::one();
    let flat = zk::layout_is_flat(&one, &montgomery_one::<C::Scalar>());
    let mut scratch = Vec::new();
    let sc = zk::flat_or_copy(coeffs, flat, &|s| limbs_of(s), &mut scratch);
    let mut out = [0u64; 12];
    let st = unsafe { zk::zk_msm(curve, handle, sc.as_ptr() as _, coeffs.len() as u64, 1, core::ptr::null(), out.as_mut_ptr() as _) };
    zk::check(st, "zk_msm").unwrap();
    curve_from_jacobian_limbs::<C>(&out) // pasta Ep { x, y, z } is Jacobian with Montgomery coordinates
}

/// every advice / fixed / permutation column of a phase against the same `g_lagrange`: one batched call
pub fn best_multiexp_batch<C: CurveAffine>(columns_dev: *const core::ffi::c_void, n: usize, count: usize, bases: &[C],
                                           stream: *mut core::ffi::c_void) -> Vec<C::Curve> {
    let curve = pasta_curve::<C>().expect("pasta curves only");
    zk::init_once();
    let handle = srs_handle(curve, bases);
    let mut out = vec![0u64; 12 * count];
    let st = unsafe { zk::zk_msm_batch_device(curve, handle, columns_dev, n as u64, count as u32, n as u64, 1, core::ptr::null(),
                                              out.as_mut_ptr() as _, stream) };
    zk::check(st, "zk_msm_batch_device").unwrap();
    out.chunks(12).map(|j| curve_from_jacobian_limbs::<C>(j)).collect()
}

pub fn best_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32) {
    // G = Fp / Fq: the library; G = a curve (Params::new, setup only): upstream
    if let Some(field) = scalar_field_of_group::<G>() {
        if a.len() >= 1 << 12 {
            zk::init_once();
            let w = limbs_of(&omega);
            // halo2 semantics: no scaling, no coset logic -- the caller supplies omega or omega^-1
            zk::check(unsafe { zk::zk_ntt(field, a.as_mut_ptr() as _, log_n, w.as_ptr() as _, 0) }, "zk_ntt").unwrap();
            return;
        }
    }
    cpu_best_fft(a, omega, log_n)
}

// ---- helpers (bodies are mechanical; omitted where they only restate pasta_curves accessors)
fn srs_handle<C: CurveAffine>(curve: i32, bases: &[C]) -> u64 { unimplemented!("zk_bases_upload of (x, y) Montgomery limbs, cached by (ptr, len, probe) like zk::SRS: {} {}", curve, bases.len()) }
fn limbs_of<F: FieldExt>(x: &F) -> [u64; 4] { let mut l = [0u64; 4]; unsafe { core::ptr::copy_nonoverlapping(x as *const F as *const u64, l.as_mut_ptr(), 4) }; l }
fn montgomery_one<F: FieldExt>() -> [u64; 4] { unimplemented!("R mod p of F (pasta_curves 0.4 fields/{{fp,fq}}.rs const R)") }
fn curve_from_jacobian_limbs<C: CurveAffine>(j: &[u64]) -> C::Curve { unimplemented!("Ep/Eq {{ x, y, z }} from 3 x 4 Montgomery limbs: {}", j.len()) }
fn scalar_field_of_group<G: Group>() -> Option<i32> { unimplemented!("ZK_FP_PALLAS / ZK_FQ_PALLAS when G is the field itself") }
fn cpu_best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve { unimplemented!("upstream body: {} {}", coeffs.len(), bases.len()) }
fn cpu_best_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32) { let _ = (a, omega, log_n); unimplemented!("upstream body") }
