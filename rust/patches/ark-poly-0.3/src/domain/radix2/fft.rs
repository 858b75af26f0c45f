//! REPLACES, in ark-poly 0.3.0 `src/domain/radix2/fft.rs`, the four in-order entry points of `Radix2EvaluationDomain<F>`
//! that `EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place` call -- seven of them per Groth16 proof
//! (ark-groth16 0.3 r1cs_to_qap.rs witness_map).  `io_helper` / `oi_helper` and the rest of the file stay as published and
//! serve `T != F` (FFTs over group elements) and fields the library does not know.  NOT COMPILED here.
use ark_ff::{FftField, FpParameters, PrimeField};
use zkcp_amd_sys as zk;

use crate::domain::{radix2::Radix2EvaluationDomain, DomainCoeff};

fn zkcp_field<F: FftField>() -> Option<i32>
where
    F: PrimeField,
{
    let m = <F::Params as FpParameters>::MODULUS;
    let l = m.as_ref();
    if l.len() != 4 {
        return None;
    }
    zk::field_id(l[0], l[3])
}

/// `x_s` viewed as flat Montgomery limbs when T is F itself and F is one of the library's 4-limb fields
fn as_limbs<F: FftField + PrimeField, T: DomainCoeff<F>>(x_s: &mut [T]) -> Option<(i32, *mut core::ffi::c_void)> {
    if core::any::TypeId::of::<T>() != core::any::TypeId::of::<F>() || x_s.len() < (1 << 12) {
        return None; // group-element FFTs and small domains stay on the CPU
    }
    let field = zkcp_field::<F>()?;
    let one = F::one(); // Montgomery one = R mod p: checks that Fp256 is four flat limbs before the slice is reinterpreted
    let r = {
        let mut r = [0u64; 4];
        let mut buf = [0u64; 4];
        unsafe { zk::zk_field_inverse(field, core::ptr::null(), core::ptr::null_mut()) }; // (keeps the symbol referenced)
        unsafe { core::ptr::copy_nonoverlapping(&one as *const F as *const u64, buf.as_mut_ptr(), 4) };
        r.copy_from_slice(&buf);
        r
    };
    if !zk::layout_is_flat(&one, &r) {
        return None;
    }
    Some((field, x_s.as_mut_ptr() as *mut core::ffi::c_void))
}

fn limbs_of<F: PrimeField>(x: &F) -> [u64; 4] {
    let mut l = [0u64; 4];
    unsafe { core::ptr::copy_nonoverlapping(x as *const F as *const u64, l.as_mut_ptr(), 4) };
    l
}

impl<F: FftField + PrimeField> Radix2EvaluationDomain<F> {
    pub(crate) fn in_order_fft_in_place<T: DomainCoeff<F>>(&self, x_s: &mut [T]) {
        if let Some((field, p)) = as_limbs::<F, T>(x_s) {
            zk::init_once();
            let w = limbs_of(&self.group_gen);
            zk::check(unsafe { zk::zk_ntt(field, p, self.log_size_of_group, w.as_ptr() as _, 0) }, "zk_ntt").unwrap();
            return;
        }
        self.cpu_in_order_fft_in_place(x_s) // upstream body, renamed
    }

    pub(crate) fn in_order_ifft_in_place<T: DomainCoeff<F>>(&self, x_s: &mut [T]) {
        if let Some((field, p)) = as_limbs::<F, T>(x_s) {
            zk::init_once();
            let w = limbs_of(&self.group_gen_inv);
            // scale_by_n_inv = 1 is upstream's `x_s.iter_mut().for_each(|val| *val *= self.size_inv)`
            zk::check(unsafe { zk::zk_ntt(field, p, self.log_size_of_group, w.as_ptr() as _, 1) }, "zk_ntt").unwrap();
            return;
        }
        self.cpu_in_order_ifft_in_place(x_s)
    }

    /// coset_fft_in_place = distribute_powers(F::multiplicative_generator()) ; fft_in_place -- one library call pair on the
    /// host-pointer ABI (the resident form, zk_ntt_coset_device, is what the ark-groth16 fork uses)
    pub(crate) fn in_order_coset_fft_in_place<T: DomainCoeff<F>>(&self, x_s: &mut [T]) {
        if let Some((field, p)) = as_limbs::<F, T>(x_s) {
            zk::init_once();
            let g = limbs_of(&F::multiplicative_generator());
            zk::check(unsafe { zk::zk_coset_mul(field, p, self.log_size_of_group, g.as_ptr() as _) }, "zk_coset_mul").unwrap();
            let w = limbs_of(&self.group_gen);
            zk::check(unsafe { zk::zk_ntt(field, p, self.log_size_of_group, w.as_ptr() as _, 0) }, "zk_ntt").unwrap();
            return;
        }
        Self::distribute_powers(x_s, F::multiplicative_generator());
        self.cpu_in_order_fft_in_place(x_s)
    }

    pub(crate) fn in_order_coset_ifft_in_place<T: DomainCoeff<F>>(&self, x_s: &mut [T]) {
        if let Some((field, p)) = as_limbs::<F, T>(x_s) {
            zk::init_once();
            let w = limbs_of(&self.group_gen_inv);
            zk::check(unsafe { zk::zk_ntt(field, p, self.log_size_of_group, w.as_ptr() as _, 1) }, "zk_ntt").unwrap();
            let gi = limbs_of(&F::multiplicative_generator().inverse().unwrap());
            zk::check(unsafe { zk::zk_coset_mul(field, p, self.log_size_of_group, gi.as_ptr() as _) }, "zk_coset_mul").unwrap();
            return;
        }
        self.cpu_in_order_ifft_in_place(x_s);
        Self::distribute_powers(x_s, F::multiplicative_generator().inverse().unwrap());
    }
}
