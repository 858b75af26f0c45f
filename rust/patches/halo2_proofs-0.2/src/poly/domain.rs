//! REPLACES three methods of `EvaluationDomain<G>` in halo2_proofs 0.2.0 `src/poly/domain.rs` when the polynomial lives in
//! a device buffer (`zkcp_amd_sys::DeviceBuf`): each is ONE fused library call.  The constants (omega, extended_omega,
//! g_coset = ZETA, ifft divisors, t_evaluations) are the ones `EvaluationDomain::new(j, k)` computes upstream; the Python
//! mirror contangle-zkcp_amd/halo2.py restates them and is parity-tested.  NOT COMPILED here.
use zkcp_amd_sys as zk;

impl<G: Group> EvaluationDomain<G> {
    /// ifft(a, omega_inv, k, ifft_divisor)
    pub fn lagrange_to_coeff_device(&self, field: i32, a: &mut zk::DeviceBuf, stream: *mut core::ffi::c_void) {
        let w = limbs_of(&self.omega_inv);
        zk::check(unsafe { zk::zk_ntt_device(field, a.ptr(), self.k, w.as_ptr() as _, 1, stream) }, "zk_ntt_device").unwrap();
    }
    /// a.resize(extended_len, 0) ; distribute_powers_zeta(a, true) ; best_fft(a, extended_omega, extended_k):
    /// the zero padding is implied (never stored or read); ZETA^(i mod 3) = ZETA^i because ZETA^3 = 1
    pub fn coeff_to_extended_device(&self, field: i32, a_ext: &mut zk::DeviceBuf, stream: *mut core::ffi::c_void) {
        let (w, z) = (limbs_of(&self.extended_omega), limbs_of(&self.g_coset));
        let st = unsafe { zk::zk_ntt_extend_device(field, a_ext.ptr(), self.extended_k, self.k, w.as_ptr() as _, 0, z.as_ptr() as _,
                                                  core::ptr::null(), stream) };
        zk::check(st, "zk_ntt_extend_device").unwrap();
    }
    /// Both forms of a column are kept (coefficients for the openings, the extended coset for the quotient): out of place, no copy.
    /// `lazy_out`: the coset values leave in the radix the lazy-limb quotient evaluator reads (ZK_NTT_OUT_R29, one constant of the
    /// last pass).  `part` of `parts`: only the sub-coset ZETA extended_omega^(i parts + part) (a sharded quotient, one part per GPU).
    pub fn lagrange_to_coeff_to(&self, field: i32, lagrange: &zk::DeviceBuf, coeffs: &mut zk::DeviceBuf, stream: *mut core::ffi::c_void) {
        let w = limbs_of(&self.omega_inv);
        zk::check(unsafe { zk::zk_ntt_oop_device(field, lagrange.ptr() as _, coeffs.ptr(), self.k, self.k, w.as_ptr() as _, 1, core::ptr::null(),
                                                 core::ptr::null(), stream) }, "zk_ntt_oop_device").unwrap();
    }
    pub fn coeff_to_extended_part_to(&self, field: i32, coeffs: &zk::DeviceBuf, out: &mut zk::DeviceBuf, part: u32, parts: u32, lazy_out: bool,
                                     stream: *mut core::ffi::c_void) {
        let g = limbs_of(&(self.g_coset * self.extended_omega.pow_vartime([part as u64])));
        let w = limbs_of(&self.extended_omega.pow_vartime([parts as u64]));
        let log_len = self.extended_k - parts.trailing_zeros();
        zk::check(unsafe { zk::zk_ntt_oop_device(field, coeffs.ptr() as _, out.ptr(), log_len, self.k, w.as_ptr() as _, if lazy_out { 2 } else { 0 },
                                                 g.as_ptr() as _, core::ptr::null(), stream) }, "zk_ntt_oop_device").unwrap();
    }
    /// best_fft(a, extended_omega_inv) ; * extended_ifft_divisor ; distribute_powers_zeta(a, false) ; truncate to n * (j - 1)
    pub fn extended_to_coeff_device(&self, field: i32, a_ext: &mut zk::DeviceBuf, stream: *mut core::ffi::c_void) {
        let (w, zi) = (limbs_of(&self.extended_omega_inv), limbs_of(&self.g_coset_inv));
        let st = unsafe { zk::zk_ntt_coset_device(field, a_ext.ptr(), self.extended_k, w.as_ptr() as _, 1, core::ptr::null(),
                                                 zi.as_ptr() as _, stream) };
        zk::check(st, "zk_ntt_coset_device").unwrap();
    }
    /// a[i] *= t_evaluations[i mod 2^(extended_k - k)]
    pub fn divide_by_vanishing_poly_device(&self, field: i32, a_ext: &mut zk::DeviceBuf, stream: *mut core::ffi::c_void) {
        let t: Vec<u64> = self.t_evaluations.iter().flat_map(|x| limbs_of(x)).collect();
        let st = unsafe { zk::zk_vec_scale_periodic_device(field, a_ext.ptr(), 1u64 << self.extended_k, t.as_ptr() as _,
                                                          self.t_evaluations.len() as u32, stream) };
        zk::check(st, "zk_vec_scale_periodic_device").unwrap();
    }
}
