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

// ---- helpers.  Nothing here depends on the in-memory layout of pasta's Fp / Fq unless `layout_is_flat` has confirmed it
// on a known value; the portable path goes through `to_repr` / `from_bytes_wide`, which are part of the public traits.
use std::any::TypeId;
use std::collections::HashMap;
use std::sync::Mutex;

/// 2^256 as a field element (the Montgomery radix R of pasta_curves 0.4's 4 x u64 fields)
fn radix<F: FieldExt>() -> F {
    let mut wide = [0u8; 64];
    wide[32] = 1;
    F::from_bytes_wide(&wide)
}
fn repr_limbs<F: FieldExt>(x: &F) -> [u64; 4] {
    let r = x.to_repr(); // 32 canonical little-endian bytes
    let b = r.as_ref();
    let mut l = [0u64; 4];
    for (i, c) in b.chunks(8).enumerate().take(4) {
        l[i] = u64::from_le_bytes([c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]]);
    }
    l
}
/// R mod p: what `F::one()` looks like in memory if the struct is the bare limb array
fn montgomery_one<F: FieldExt>() -> [u64; 4] { repr_limbs(&radix::<F>()) }
/// Montgomery limbs (x R mod p) of x, without looking at the struct
pub(crate) fn limbs_of<F: FieldExt>(x: &F) -> [u64; 4] { repr_limbs(&(*x * radix::<F>())) }
/// the field element whose Montgomery limbs are `m`
pub(crate) fn from_montgomery_limbs<F: FieldExt>(m: &[u64]) -> F {
    let mut wide = [0u8; 64];
    for i in 0..4 {
        wide[8 * i..8 * i + 8].copy_from_slice(&m[i].to_le_bytes());
    }
    F::from_bytes_wide(&wide) * radix::<F>().invert().unwrap() // (x R) R^-1
}

lazy_static::lazy_static! {
    /// (address, length, curve) -> (probe limbs of the first / middle / last point, handle): Params::g and g_lagrange live
    /// as long as the Params, so each is uploaded once
    static ref PASTA_SRS: Mutex<HashMap<(usize, usize, i32), (Vec<u64>, u64)>> = Mutex::new(HashMap::new());
}
fn affine_limbs<C: CurveAffine>(p: &C, out: &mut Vec<u64>) {
    // identity -> (0, 0), which is how the library marks it (0, 0 is not on either curve)
    match Option::<pasta_curves::arithmetic::Coordinates<C>>::from(p.coordinates()) {
        Some(c) => {
            out.extend_from_slice(&limbs_of(c.x()));
            out.extend_from_slice(&limbs_of(c.y()));
        }
        None => out.extend_from_slice(&[0u64; 8]),
    }
}
fn srs_handle<C: CurveAffine>(curve: i32, bases: &[C]) -> u64 {
    let n = bases.len();
    let mut probe = Vec::new();
    if n > 0 {
        for i in [0, n / 2, n - 1] {
            affine_limbs(&bases[i], &mut probe);
        }
    }
    let key = (bases.as_ptr() as usize, n, curve);
    let mut map = PASTA_SRS.lock().unwrap();
    if let Some((p, h)) = map.get(&key) {
        if *p == probe {
            return *h;
        }
        unsafe { zk::zk_bases_free(*h) };
    }
    let mut flat = Vec::with_capacity(8 * n);
    for b in bases {
        affine_limbs(b, &mut flat);
    }
    let mut h = 0u64;
    zk::check(unsafe { zk::zk_bases_upload(curve, flat.as_ptr() as _, n as u64, &mut h) }, "zk_bases_upload").unwrap();
    map.insert(key, (probe, h));
    h
}
/// pasta Ep / Eq are Jacobian (x, y, z); z = 0 is the identity.  `new_jacobian` checks the curve equation.
pub(crate) fn curve_from_jacobian_limbs<C: CurveAffine>(j: &[u64]) -> C::Curve {
    use pasta_curves::arithmetic::CurveExt;
    if j[8..12].iter().all(|&w| w == 0) {
        return C::Curve::identity();
    }
    let x = from_montgomery_limbs::<C::Base>(&j[0..4]);
    let y = from_montgomery_limbs::<C::Base>(&j[4..8]);
    let z = from_montgomery_limbs::<C::Base>(&j[8..12]);
    Option::from(C::Curve::new_jacobian(x, y, z)).expect("zkcp_amd returned a point off the curve")
}
/// `best_fft` is generic over halo2's `Group`; the library takes the case where the group IS its scalar field
fn scalar_field_of_group<G: Group>() -> Option<i32> {
    if TypeId::of::<G>() != TypeId::of::<G::Scalar>() {
        return None; // a curve: Params::new's FFT over points stays on the CPU
    }
    let m = <G::Scalar as PrimeField>::MODULUS;
    if m.ends_with("992d30ed00000001") {
        Some(zk::ZK_FP_PALLAS)
    } else if m.ends_with("8c46eb2100000001") {
        Some(zk::ZK_FQ_PALLAS)
    } else {
        None
    }
}

// ---- CPU paths for the inputs the library does not take (small sizes, other curves, FFTs over points).  Plain
// single-threaded forms with the same results as upstream's; a maintainer who wants upstream's multi-threaded bodies for
// these sizes keeps them under these names instead.
fn cpu_best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve {
    // bucket method, 8-bit unsigned windows over the canonical bytes, most significant window first
    let reprs: Vec<_> = coeffs.iter().map(|c| c.to_repr()).collect();
    let mut acc = C::Curve::identity();
    for byte in (0..32).rev() {
        for _ in 0..8 {
            acc = acc.double();
        }
        let mut buckets = vec![C::Curve::identity(); 255];
        for (r, b) in reprs.iter().zip(bases) {
            let d = r.as_ref()[byte] as usize;
            if d != 0 {
                buckets[d - 1] += *b;
            }
        }
        let mut run = C::Curve::identity();
        for b in buckets.iter().rev() {
            run += b;
            acc += run; // sum_d d * bucket[d]
        }
    }
    acc
}
fn cpu_best_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32) {
    let n = a.len();
    assert_eq!(n, 1usize << log_n);
    for i in 0..n {
        let r = if log_n == 0 { 0 } else { ((i as u64).reverse_bits() >> (64 - log_n)) as usize };
        if i < r {
            a.swap(i, r);
        }
    }
    let mut m = 1usize;
    for s in 0..log_n {
        // w_m = omega^(n / 2m)
        let mut w_m = omega;
        for _ in 0..(log_n - 1 - s) {
            w_m = w_m * w_m;
        }
        for k in (0..n).step_by(2 * m) {
            let mut w = G::Scalar::one();
            for j in 0..m {
                let mut t = a[k + j + m];
                t.group_scale(&w);
                let mut u = a[k + j];
                a[k + j].group_add(&t);
                u.group_sub(&t);
                a[k + j + m] = u;
                w = w * w_m;
            }
        }
        m *= 2;
    }
}
