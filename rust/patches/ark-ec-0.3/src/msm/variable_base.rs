//! REPLACES `VariableBaseMSM::multi_scalar_mul` in ark-ec 0.3.0 `src/msm/variable_base.rs` (the Pippenger the reference
//! reaches five times per proof through `Groth16::<Bls12_381>::prove`, lib/src/zk/verifiable_encryption.rs:92).
//! `cpu_multi_scalar_mul` serves the curves and sizes the library does not.
//! NOT COMPILED in this repository's build image (no Rust toolchain); binds include/zkcp_amd.h through zkcp-amd-sys.
use ark_ff::{BigInteger, FpParameters, PrimeField};
use ark_serialize::{CanonicalDeserialize, CanonicalSerialize};
use ark_std::vec::Vec;
use zkcp_amd_sys as zk;

use crate::{AffineCurve, ProjectiveCurve};

pub struct VariableBaseMSM;

/// (library curve id, u64 limbs per coordinate) of `G`, from its base field: prime subfield modulus + extension degree
fn zkcp_curve<G: AffineCurve>() -> Option<(i32, usize)>
where
    G::BaseField: ark_ff::Field,
{
    use ark_ff::Field;
    type Prime<G> = <<G as AffineCurve>::BaseField as Field>::BasePrimeField;
    let m = <<Prime<G> as PrimeField>::Params as FpParameters>::MODULUS;
    let limbs = m.as_ref().len();
    let ext = <G::BaseField as Field>::extension_degree() as usize;
    zk::curve_id(m.as_ref()[0], limbs, ext).map(|c| (c, limbs * ext))
}

impl VariableBaseMSM {
    pub fn multi_scalar_mul<G: AffineCurve>(bases: &[G], scalars: &[<G::ScalarField as PrimeField>::BigInt]) -> G::Projective {
        let size = ark_std::cmp::min(bases.len(), scalars.len());
        let (curve, limbs) = match zkcp_curve::<G>() {
            Some(c) if size >= 1 << 10 => c, // below ~2^10 pairs the launch sequence costs more than the CPU
            _ => return Self::cpu_multi_scalar_mul(bases, scalars),
        };
        zk::init_once();
        let bases = &bases[..size];
        // SRS residency: uploaded once per query vector, in ark-serialize's uncompressed form (layout-independent)
        let handle = zk::SRS.get_or_upload(curve, bases.as_ptr() as usize, size, &|i, out: &mut Vec<u8>| {
            bases[i].serialize_uncompressed(out).unwrap()
        });
        // scalars are canonical BigInts (`into_repr()`): scalars_are_montgomery = 0
        let mut scratch = Vec::new();
        let one = <G::ScalarField as PrimeField>::BigInt::from(1u64);
        let flat = zk::layout_is_flat(&one, &[1, 0, 0, 0]);
        let sc = zk::flat_or_copy(&scalars[..size], flat, &|s| {
            let mut l = [0u64; 4];
            l.copy_from_slice(&s.as_ref()[..4]);
            l
        }, &mut scratch);
        let mut out = ark_std::vec![0u64; 3 * limbs];
        let st = unsafe { zk::zk_msm(curve, handle, sc.as_ptr() as _, size as u64, 0, core::ptr::null(), out.as_mut_ptr() as _) };
        zk::check(st, "zk_msm").unwrap(); // no CPU fallback behind a failing device: fail loudly
        let bytes = zk::jacobian_to_ark_uncompressed(curve, &out);
        G::deserialize_unchecked(&bytes[..]).unwrap().into_projective()
    }

    /// The CPU path for curves / sizes the library does not take.  A plain single-threaded bucket method over 8-bit
    /// windows of the canonical scalar (the sum is the same group element whatever the window schedule); a maintainer who
    /// wants upstream's window-parallel body for these inputs keeps it under this name instead.
    fn cpu_multi_scalar_mul<G: AffineCurve>(bases: &[G], scalars: &[<G::ScalarField as PrimeField>::BigInt]) -> G::Projective {
        use ark_ff::Zero;
        let size = ark_std::cmp::min(bases.len(), scalars.len());
        let bits = <G::ScalarField as PrimeField>::size_in_bits();
        let windows = (bits + 7) / 8;
        let mut acc = G::Projective::zero();
        for w in (0..windows).rev() {
            for _ in 0..8 {
                acc.double_in_place();
            }
            let mut buckets = ark_std::vec![G::Projective::zero(); 255];
            for (b, s) in bases[..size].iter().zip(&scalars[..size]) {
                let limb = s.as_ref()[w / 8];
                let d = ((limb >> (8 * (w % 8))) & 0xff) as usize;
                if d != 0 {
                    buckets[d - 1].add_assign_mixed(b);
                }
            }
            let mut run = G::Projective::zero();
            for b in buckets.iter().rev() {
                run += b;
                acc += &run; // sum over d of d * bucket[d]
            }
        }
        acc
    }
}
