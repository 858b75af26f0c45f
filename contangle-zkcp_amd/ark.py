"""Mirror of the arkworks 0.3 interfaces on the reference's Groth16 prove path.

  VariableBaseMSM.multi_scalar_mul(bases, scalars)   ark-ec 0.3  src/msm/variable_base.rs
  Radix2EvaluationDomain(num_coeffs)                 ark-poly 0.3 src/domain/radix2/mod.rs
      .fft_in_place / .ifft_in_place / .coset_fft_in_place / .coset_ifft_in_place
  FixedBaseMSM.multi_scalar_mul(...)                 ark-ec 0.3  src/msm/fixed_base.rs  (key generation,
                                                     lib/src/zk/encryption.rs:169 via ark-groth16 0.3 generate_parameters)

reached from lib/src/zk/verifiable_encryption.rs:92, lib/src/zk/encryption.rs:76,
lib/src/zk/sample_entries.rs:86, lib/src/zk/property.rs:133 via ark-groth16 0.3 create_proof
(witness_map: 7 NTTs; then 4 G1 MSMs + 1 G2 MSM).  Same names, argument meaning
and error behaviour; the arithmetic runs in the HIP library.
"""
import numpy as np

from . import (Bases, coset_mul, curve_id, field_id, field_inverse, fixed_base_msm_device, msm, multiplicative_generator, ntt,
               root_of_unity)


class VariableBaseMSM:
    @staticmethod
    def multi_scalar_mul(bases, scalars):
        """bases: `Bases` (device-resident affine points) ; scalars: canonical BigInt limbs [n,4]
        (what `into_repr()` yields).  Like ark-ec, uses min(len(bases), len(scalars)) pairs.
        Returns the projective (Jacobian) sum."""
        n = min(bases.n, int(scalars.shape[0]))
        return msm(bases, scalars[:n], montgomery=False)


class FixedBaseMSM:
    """ark-ec 0.3 FixedBaseMSM: `get_mul_window_size` / `get_window_table` / `multi_scalar_mul`, followed upstream by
    `batch_normalization_into_affine`.  The window table is an implementation detail of the device path (8-bit windows,
    rebuilt per call), so the three upstream steps collapse into one call that returns affine points."""

    @staticmethod
    def get_mul_window_size(num_scalars):
        """upstream heuristic (3 below 32 scalars, else ceil(ln n)); informational here -- the device path uses 8"""
        import math
        return 3 if num_scalars < 32 else int(math.ceil(math.log(num_scalars)))

    @staticmethod
    def multi_scalar_mul(curve, base, d_scalars, d_out, montgomery=False, stream=0):
        """d_out[i] = [d_scalars[i]] base as affine points (device buffers); base = (x, y) Montgomery limbs or None (generator)"""
        fixed_base_msm_device(curve, d_scalars, d_out, int(d_scalars.shape[0]), base=base, montgomery=montgomery, stream=stream)
        return d_out


class Radix2EvaluationDomain:
    """ark-poly 0.3 Radix2EvaluationDomain<F>::new(num_coeffs): size = next power of two,
    group_gen = TWO_ADIC_ROOT_OF_UNITY^(2^(S - log_size)); None if the field has no such subgroup."""

    def __new__(cls, field, num_coeffs):
        size = 1 if num_coeffs <= 1 else 1 << (int(num_coeffs) - 1).bit_length()
        log_size = size.bit_length() - 1
        try:
            gen = root_of_unity(field, log_size)
        except Exception:
            return None  # ark returns None when log_size > TWO_ADICITY
        self = object.__new__(cls)
        self.field = field_id(field)
        self.size = size
        self.log_size_of_group = log_size
        self.group_gen = gen
        self.group_gen_inv = field_inverse(field, gen)
        self.generator = multiplicative_generator(field)      # F::multiplicative_generator(): coset shift
        self.generator_inv = field_inverse(field, self.generator)
        return self

    def _check(self, a):
        if int(a.shape[0]) != self.size:
            raise ValueError("ark-poly resizes to the domain size; pass exactly %d coefficients" % self.size)

    def fft_in_place(self, a):
        self._check(a)
        return ntt(self.field, a, self.group_gen)

    def ifft_in_place(self, a):
        self._check(a)
        return ntt(self.field, a, self.group_gen_inv, scale_by_n_inv=True)

    def coset_fft_in_place(self, a):
        """distribute_powers(generator) ; fft  -- one fused launch sequence on device buffers"""
        self._check(a)
        return ntt(self.field, a, self.group_gen, coset_pre=self.generator)

    def coset_ifft_in_place(self, a):
        """ifft ; distribute_powers(generator^-1)"""
        self._check(a)
        return ntt(self.field, a, self.group_gen_inv, scale_by_n_inv=True, coset_post=self.generator_inv)
