"""Mirror of halo2_proofs 0.2 `arithmetic.rs` free functions:

  best_multiexp(coeffs, bases) -> C::Curve      coeffs are field elements as stored (Montgomery)
  best_fft(a, omega, log_n)                     no scaling, no coset logic; caller supplies omega or omega^-1

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
