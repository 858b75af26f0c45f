//! REPLACES the body of `create_proof` in halo2_proofs 0.2.0 `src/poly/multiopen/prover.rs` between the challenges x_1 / x_2 and
//! the call of `commitment::create_proof` when the committed polynomials live in device buffers (one `zkcp_amd_sys::DeviceBuf`
//! of n coefficients each, left there by `EvaluationDomain::lagrange_to_coeff_device(.., out)`).  The transcript, the blinds
//! and `construct_intermediate_sets` are upstream's.  NOT COMPILED here.
//!
//! upstream                                             here
//!   q_polys[set] = q_polys[set] * x_1 + poly  (loop)     zk_vec_fold_many_device: the whole fold of a set in one pass
//!   kate_division(&poly, point) folded over the points   zk_kate_division_device per point (out of place first, then in place)
//!   q_prime_poly = q_prime_poly * x_2 + poly             zk_vec_muladd_device
//!   params.commit(&q_prime_poly, blind)                  best_multiexp on the resident buffer (arithmetic.rs of this fork)
//!   eval_polynomial(q_i, x_3) for every set              zk_poly_eval_batch_device (the set polynomials are one table)
//!   q_prime_poly * x_4 + q_i  (loop)                     zk_vec_muladd_device
use zkcp_amd_sys as zk;

use super::super::super::arithmetic::{limbs_of, CurveAffine, FieldExt};

/// `sets[s]` = (first row, row count) of set s in `table` (rows of n coefficients: the caller lays the committed polynomials
/// out grouped by point set), `points[s]` = the set's points.  Leaves q_s in `q_polys` row s and returns q' in `q_prime`.
pub fn fold_and_divide_device<C: CurveAffine>(field: i32, n: u64, table: &zk::DeviceBuf, sets: &[(u64, u32)], points: &[Vec<C::Scalar>],
                                              x_1: C::Scalar, x_2: C::Scalar, q_polys: &mut zk::DeviceBuf, q_prime: &mut zk::DeviceBuf,
                                              tmp: &mut zk::DeviceBuf, stream: *mut core::ffi::c_void) {
    let row = |buf: &zk::DeviceBuf, r: u64| unsafe { (buf.ptr() as *mut u8).add((r * n * 32) as usize) as *mut core::ffi::c_void };
    let (x1, x2) = (limbs_of(&x_1), limbs_of(&x_2));
    for (s, &(first, count)) in sets.iter().enumerate() {
        zk::check(unsafe { zk::zk_vec_fold_many_device(field, row(q_polys, s as u64), row(table, first) as _, n as i64, count, n, x1.as_ptr() as _, stream) },
                  "zk_vec_fold_many_device").unwrap();
    }
    for (s, pts) in points.iter().enumerate() {
        let dst = if s == 0 { q_prime.ptr() } else { tmp.ptr() };
        for (j, pt) in pts.iter().enumerate() {
            let p = limbs_of(pt);
            let src = if j == 0 { row(q_polys, s as u64) as *const core::ffi::c_void } else { dst as *const core::ffi::c_void };
            zk::check(unsafe { zk::zk_kate_division_device(field, src, dst, n, p.as_ptr() as _, stream) }, "zk_kate_division_device").unwrap();
        }
        if s > 0 {
            zk::check(unsafe { zk::zk_vec_muladd_device(field, q_prime.ptr(), tmp.ptr(), n, x2.as_ptr() as _, stream) }, "zk_vec_muladd_device").unwrap();
        }
    }
}

/// the evaluations of the set polynomials at x_3 (written to the transcript by the caller), then p = q' x_4^sets + ...
pub fn evals_and_final_fold_device<C: CurveAffine>(field: i32, n: u64, n_sets: u32, q_polys: &zk::DeviceBuf, q_prime: &mut zk::DeviceBuf,
                                                   x_3: C::Scalar, x_4: C::Scalar, stream: *mut core::ffi::c_void) -> Vec<[u64; 4]> {
    let (x3, x4) = (limbs_of(&x_3), limbs_of(&x_4));
    let mut out = vec![[0u64; 4]; n_sets as usize];
    zk::check(unsafe { zk::zk_poly_eval_batch_device(field, q_polys.ptr(), n, n_sets, n, x3.as_ptr() as _, out.as_mut_ptr() as _, stream) },
              "zk_poly_eval_batch_device").unwrap();
    for s in 0..n_sets as u64 {
        let q = unsafe { (q_polys.ptr() as *const u8).add((s * n * 32) as usize) as *const core::ffi::c_void };
        zk::check(unsafe { zk::zk_vec_muladd_device(field, q_prime.ptr(), q, n, x4.as_ptr() as _, stream) }, "zk_vec_muladd_device").unwrap();
    }
    out
}
