"""Mirror of halo2_proofs 0.2 `arithmetic.rs` free functions:

  best_multiexp(coeffs, bases) -> C::Curve      coeffs are field elements as stored (Montgomery)
  best_fft(a, omega, log_n)                     no scaling, no coset logic; caller supplies omega or omega^-1
  coeff_to_extended(...)                        poly/domain.rs EvaluationDomain::coeff_to_extended: zero-extend to the
                                                extended domain, zeta coset shift, best_fft -- one fused device call

The reference's halo2 crate (circuits-halo2/src/encryption.rs:254-296) never reaches these --
it only runs MockProver -- so they are exercised here as the shape donor for the
2^20-row synthetic workload.
"""
from . import msm, ntt


def best_multiexp(coeffs, bases):
    if int(coeffs.shape[0]) != bases.n:
        raise AssertionError("assertion failed: coeffs.len() == bases.len()")   # halo2: assert_eq!
    return msm(bases, coeffs, montgomery=True)


def best_fft(field, a, omega, log_n):
    if int(a.shape[0]) != 1 << log_n:
        raise AssertionError("assertion failed: a.len() == 1 << log_n")
    return ntt(field, a, omega)


def coeff_to_extended(field, d_ext, k, omega_ext, zeta, stream=0):
    """halo2_proofs 0.2 EvaluationDomain::coeff_to_extended: `d_ext` is a device buffer of the EXTENDED length whose first
    2^k entries hold the coefficients; they are taken as zero-extended (`a.resize(extended_len, 0)` upstream -- the padding
    is neither written nor read here), shifted onto the zeta coset (`distribute_powers_zeta`) and transformed in place."""
    return ntt(field, d_ext, omega_ext, stream=stream, coset_pre=zeta, in_log=k)
