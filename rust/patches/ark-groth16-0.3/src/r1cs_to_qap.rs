//! REPLACES `R1CStoQAP::witness_map` in ark-groth16 0.3.0 `src/r1cs_to_qap.rs` for the library's scalar fields: the three
//! evaluation vectors are uploaded ONCE, the seven NTTs and the pointwise glue run in HBM (`zk_groth16_witness_map_device`),
//! and `h` stays resident for the h_query MSM (call site lib/src/zk/encryption.rs:76).  NOT COMPILED here.
//! `DeviceVec` is the fork's own thin RAII wrapper over hipMalloc / hipMemcpy (hip-sys), omitted for brevity.
use ark_ff::PrimeField;
use ark_relations::r1cs::{ConstraintSystemRef, Result as R1CSResult, SynthesisError};
use zkcp_amd_sys as zk;

use crate::device::DeviceVec;

pub struct WitnessMapOnDevice {
    pub h: DeviceVec,      // m Montgomery field elements; the first m - 1 feed zk_msm_submit(.., scalars_are_montgomery = 1, ..)
    pub domain_size: usize,
}

pub fn witness_map_on_device<F: PrimeField>(field: i32, cs: ConstraintSystemRef<F>, stream: *mut core::ffi::c_void) -> R1CSResult<WitnessMapOnDevice> {
    let matrices = cs.to_matrices().ok_or(SynthesisError::AssignmentMissing)?;
    let (num_inputs, num_constraints) = (cs.num_instance_variables(), cs.num_constraints());
    let m = (num_constraints + num_inputs).next_power_of_two();
    let cs = cs.borrow().unwrap();
    let full_assignment: Vec<F> = [cs.instance_assignment.as_slice(), cs.witness_assignment.as_slice()].concat();
    // <A_i, z>, <B_i, z>, <C_i, z>: the sparse products stay upstream's (rayon); the CSR kernel of the library
    // (zk_r1cs_matvec_device, include/zkcp_amd_prover.h) takes over once the matrices are uploaded with the proving key
    let mut a = vec![F::zero(); m];
    let mut b = vec![F::zero(); m];
    let mut c = vec![F::zero(); m];
    for (i, ((at, bt), ct)) in matrices.a.iter().zip(&matrices.b).zip(&matrices.c).enumerate() {
        a[i] = evaluate_constraint(at, &full_assignment);
        b[i] = evaluate_constraint(bt, &full_assignment);
        c[i] = evaluate_constraint(ct, &full_assignment);
    }
    a[num_constraints..num_constraints + num_inputs].clone_from_slice(&full_assignment[..num_inputs]);
    zk::init_once();
    let (da, db, dc) = (DeviceVec::from_slice(&a, stream), DeviceVec::from_slice(&b, stream), DeviceVec::from_slice(&c, stream));
    let st = unsafe { zk::zk_groth16_witness_map_device(field, da.ptr(), db.ptr(), dc.ptr(), m.trailing_zeros(), stream) };
    zk::check(st, "zk_groth16_witness_map_device").map_err(|_| SynthesisError::Unsatisfiable)?;
    Ok(WitnessMapOnDevice { h: da, domain_size: m })
}

fn evaluate_constraint<F: PrimeField>(terms: &[(F, usize)], assignment: &[F]) -> F {
    terms.iter().fold(F::zero(), |acc, (coeff, idx)| acc + *coeff * assignment[*idx])
}
