"""Mirror of halo2_proofs 0.2 on the commitment / quotient path:

  arithmetic.rs    best_multiexp(coeffs, bases) -> C::Curve    coeffs are field elements as stored (Montgomery)
                   best_fft(a, omega, log_n)                   no scaling, no coset logic; caller supplies omega or omega^-1
  poly/domain.rs   EvaluationDomain::new(j, k), lagrange_to_coeff, coeff_to_extended, extended_to_coeff,
                   divide_by_vanishing_poly, and the constants it derives (omega, extended_omega, g_coset = ZETA,
                   ifft divisors, t_evaluations)

The reference's halo2 crate (circuits-halo2/src/encryption.rs:254-296) never reaches these -- it only runs
MockProver -- so they are exercised here as the shape donor for the 2^20-row synthetic workload.
Same names, argument meaning and assertion behaviour as upstream; the arithmetic runs in the HIP library.
"""
import numpy as np

from . import (field_id, field_inverse, field_modulus, msm, msm_batch, multiplicative_generator, ntt, root_of_unity, vec_op)


def best_multiexp(coeffs, bases):
    if int(coeffs.shape[0]) != bases.n:
        raise AssertionError("assertion failed: coeffs.len() == bases.len()")   # halo2: assert_eq!
    return msm(bases, coeffs, montgomery=True)


def best_multiexp_batch(columns, bases, stream=0):
    """the commitments of several columns against the same `Params::g_lagrange` (upstream: a loop of commit_lagrange
    calls with no data dependence between them); columns: device buffer [count, n, 4]"""
    if int(columns.shape[1]) != bases.n:
        raise AssertionError("assertion failed: coeffs.len() == bases.len()")
    return msm_batch(bases, columns, montgomery=True, stream=stream)


def best_fft(field, a, omega, log_n):
    if int(a.shape[0]) != 1 << log_n:
        raise AssertionError("assertion failed: a.len() == 1 << log_n")
    return ntt(field, a, omega)


def coeff_to_extended(field, d_ext, k, omega_ext, zeta, stream=0):
    """halo2_proofs 0.2 EvaluationDomain::coeff_to_extended: `d_ext` is a device buffer of the EXTENDED length whose first
    2^k entries hold the coefficients; they are taken as zero-extended (`a.resize(extended_len, 0)` upstream -- the padding
    is neither written nor read here), shifted onto the zeta coset (`distribute_powers_zeta`) and transformed in place."""
    return ntt(field, d_ext, omega_ext, stream=stream, coset_pre=zeta, in_log=k)


def _mont_limbs(x, p):
    v = (x << 256) % p
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def zeta(field):
    """pasta_curves 0.4 `FieldExt::ZETA` = GENERATOR^((p - 1) / 3), a primitive cube root of unity, Montgomery"""
    p = field_modulus(field)
    g = multiplicative_generator(field)
    gi = sum(int(w) << (64 * i) for i, w in enumerate(g.tolist())) * pow(1 << 256, -1, p) % p
    assert (p - 1) % 3 == 0
    return _mont_limbs(pow(gi, (p - 1) // 3, p), p)


class EvaluationDomain:
    """halo2_proofs 0.2 poly/domain.rs EvaluationDomain<G>::new(j, k): n = 2^k rows, gates of degree j; the quotient is
    computed on the extended domain of size 2^extended_k, extended_k = k + ceil(log2(j - 1)), on the coset ZETA * H_ext."""

    def __init__(self, field, j, k):
        self.field = field_id(field)
        self.k = k
        self.quotient_poly_degree = j - 1
        self.n = 1 << k
        extended_k = k
        while (1 << extended_k) < self.n * self.quotient_poly_degree:
            extended_k += 1
        self.extended_k = extended_k
        self.omega = root_of_unity(field, k)                       # ROOT_OF_UNITY^(2^(S - k))
        self.omega_inv = field_inverse(field, self.omega)
        self.extended_omega = root_of_unity(field, extended_k)
        self.extended_omega_inv = field_inverse(field, self.extended_omega)
        self.g_coset = zeta(field)                                 # ZETA
        self.g_coset_inv = field_inverse(field, self.g_coset)      # = ZETA^2
        p = field_modulus(field)
        self._p = p
        # t_evaluations[i] = 1 / ((ZETA * extended_omega^i)^n - 1), i < 2^(extended_k - k): the vanishing polynomial X^n - 1
        # takes only that many distinct values on the coset
        to_int = lambda a: sum(int(w) << (64 * i) for i, w in enumerate(a.tolist())) * pow(1 << 256, -1, p) % p
        zi, wi = to_int(self.g_coset), to_int(self.extended_omega)
        m = 1 << (extended_k - k)
        self.t_evaluations = np.stack([_mont_limbs(pow((pow(zi * pow(wi, i, p) % p, self.n, p) - 1) % p, -1, p), p) for i in range(m)])

    def extended_len(self):
        return 1 << self.extended_k

    # ---- device buffers (torch tensors; numpy under the CPU test emulator), in place
    def lagrange_to_coeff(self, a, stream=0):
        """ifft(a, omega_inv, k, ifft_divisor): best_fft with omega^-1, then every element times n^-1"""
        if int(a.shape[0]) != self.n:
            raise AssertionError("assertion failed: a.values.len() == 1 << self.k")
        return ntt(self.field, a, self.omega_inv, scale_by_n_inv=True, stream=stream, device=True)

    def coeff_to_lagrange(self, a, stream=0):
        if int(a.shape[0]) != self.n:
            raise AssertionError("assertion failed: a.values.len() == 1 << self.k")
        return ntt(self.field, a, self.omega, stream=stream, device=True)

    def coeff_to_extended(self, a_ext, stream=0):
        """`a_ext`: buffer of extended_len() whose first n entries are the coefficients (the rest is treated as the zeros
        upstream's `resize` appends): distribute_powers_zeta(into_coset) ; best_fft(extended_omega)"""
        if int(a_ext.shape[0]) != self.extended_len():
            raise AssertionError("assertion failed: a.len() == extended_len")
        return ntt(self.field, a_ext, self.extended_omega, stream=stream, coset_pre=self.g_coset,
                   in_log=self.k if self.extended_k > self.k else None, device=True)

    def extended_to_coeff(self, a_ext, stream=0):
        """best_fft(extended_omega_inv) ; times extended_ifft_divisor ; distribute_powers_zeta(out of the coset).  Upstream
        then truncates to n * quotient_poly_degree coefficients: use the first quotient_len() entries."""
        if int(a_ext.shape[0]) != self.extended_len():
            raise AssertionError("assertion failed: a.values.len() == extended_len")
        return ntt(self.field, a_ext, self.extended_omega_inv, scale_by_n_inv=True, stream=stream, coset_post=self.g_coset_inv, device=True)

    def quotient_len(self):
        return self.n * self.quotient_poly_degree

    def divide_by_vanishing_poly(self, a_ext, stream=0):
        """a[i] *= t_evaluations[i mod 2^(extended_k - k)]  (the inverse of X^n - 1 on the coset, periodic)"""
        if int(a_ext.shape[0]) != self.extended_len():
            raise AssertionError("assertion failed: a.values.len() == extended_len")
        return vec_op(self.field, "scale_periodic", a_ext, b=self.t_evaluations, stream=stream)
