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
import ctypes

import numpy as np

from . import (Bases, _check, _np64, _ptr, base_limbs, curve_id, field_id, field_inverse, field_modulus, load, msm, msm_batch, msm_submit,
               multiplicative_generator, ntt, root_of_unity, scalar_field, vec_op)

PROVER_EXPORTS = ["zk_batch_invert_device", "zk_prefix_product_device", "zk_halo2_permutation_product_device",
                  "zk_halo2_lookup_product_device", "zk_inner_product_device", "zk_vec_fold_device", "zk_ipa_fold_bases_device",
                  "zk_expr_eval_device", "zk_ipa_virtual_scalars_device", "zk_ipa_update_weights_device", "zk_ipa_collapse_device", "zk_ipa_collapse_range_device", "zk_ipa_round_device",
                  "zk_poly_eval_device", "zk_poly_eval_batch_device", "zk_vec_muladd_device", "zk_vec_muladd_to_device", "zk_kate_division_device", "zk_vec_powers_device", "zk_vec_fold_many_device",
                  "zk_ipa_fold_round_device", "zk_expr_eval_lazy_device", "zk_expr_configure", "zk_expr_specialised_source"]


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
    def lagrange_to_coeff(self, a, stream=0, out=None):
        """ifft(a, omega_inv, k, ifft_divisor): best_fft with omega^-1, then every element times n^-1.  out: write the
        coefficients there and leave the Lagrange values alone (upstream returns a new Polynomial; the prover keeps both)"""
        if int(a.shape[0]) != self.n:
            raise AssertionError("assertion failed: a.values.len() == 1 << self.k")
        if out is not None:
            if int(out.shape[0]) != self.n:
                raise AssertionError("assertion failed: out.len() == 1 << self.k")
            return ntt(self.field, out, self.omega_inv, scale_by_n_inv=True, stream=stream, src=a)
        return ntt(self.field, a, self.omega_inv, scale_by_n_inv=True, stream=stream, device=True)

    def coeff_to_lagrange(self, a, stream=0):
        if int(a.shape[0]) != self.n:
            raise AssertionError("assertion failed: a.values.len() == 1 << self.k")
        return ntt(self.field, a, self.omega, stream=stream, device=True)

    def coeff_to_extended(self, a_ext, stream=0, coeffs=None, lazy_out=False, parts=1):
        """`a_ext`: buffer of extended_len() whose first n entries are the coefficients (the rest is treated as the zeros
        upstream's `resize` appends): distribute_powers_zeta(into_coset) ; best_fft(extended_omega).  coeffs: take the n
        coefficients from that buffer instead (it is left untouched: the openings evaluate it later).
        parts: store the result sub-coset by sub-coset -- a_ext.view(parts, extended_len / parts)[j] is what
        coeff_to_extended_part(.., j, parts) computes (ZK_NTT_OUT_SUBCOSETS: the same transform, other store addresses)"""
        if int(a_ext.shape[0]) != self.extended_len():
            raise AssertionError("assertion failed: a.len() == extended_len")
        self._part_check(0, parts)
        flags = (2 if lazy_out else 0) | ((parts.bit_length() - 1) << 4)
        if coeffs is not None:
            if int(coeffs.shape[0]) != self.n:
                raise AssertionError("assertion failed: a.len() == 1 << self.k")
            return ntt(self.field, a_ext, self.extended_omega, stream=stream, coset_pre=self.g_coset, in_log=self.k, src=coeffs,
                       scale_by_n_inv=flags)
        return ntt(self.field, a_ext, self.extended_omega, stream=stream, coset_pre=self.g_coset,
                   in_log=self.k if self.extended_k > self.k else None, device=True, scale_by_n_inv=flags)

    def extended_to_coeff(self, a_ext, stream=0):
        """best_fft(extended_omega_inv) ; times extended_ifft_divisor ; distribute_powers_zeta(out of the coset).  Upstream
        then truncates to n * quotient_poly_degree coefficients: use the first quotient_len() entries."""
        if int(a_ext.shape[0]) != self.extended_len():
            raise AssertionError("assertion failed: a.values.len() == extended_len")
        return ntt(self.field, a_ext, self.extended_omega_inv, scale_by_n_inv=True, stream=stream, coset_post=self.g_coset_inv, device=True)

    # ---- the extended coset in `parts` sub-cosets (one per GPU of a sharded prover): sub-coset j holds the extended
    # evaluations e = i * parts + j, i.e. the points ZETA extended_omega^j (extended_omega^parts)^i.  Every column's values on
    # sub-coset j are ONE transform of size extended_len / parts (coset generator ZETA extended_omega^j), rotations by r
    # rows stay inside the sub-coset (shift r * rot_scale_part), and the vanishing polynomial takes 2^(extended_k - k) / parts
    # values there -- so the quotient numerator shards with no exchange until h itself (32 bytes per extended row).
    def _part_check(self, part, parts):
        m = 1 << (self.extended_k - self.k)
        if parts < 1 or parts > m or (parts & (parts - 1)) or not 0 <= part < parts:
            raise AssertionError("parts must be a power of two <= 2^(extended_k - k) and 0 <= part < parts")

    def part_len(self, parts):
        return self.extended_len() // parts

    def rot_scale_part(self, parts):
        return (1 << (self.extended_k - self.k)) // parts

    def _part_constants(self, part, parts):
        key = (part, parts)
        if not hasattr(self, "_parts"):
            self._parts = {}
        if key not in self._parts:
            p = self._p
            to_int = lambda a: sum(int(w) << (64 * i) for i, w in enumerate(a.tolist())) * pow(1 << 256, -1, p) % p
            wi, zi = to_int(self.extended_omega), to_int(self.g_coset)
            g = zi * pow(wi, part, p) % p
            self._parts[key] = (_mont_limbs(g, p), _mont_limbs(pow(wi, parts, p), p))
        return self._parts[key]

    def coeff_to_extended_part(self, coeffs, out, part, parts, stream=0, lazy_out=False):
        """out[i] = the polynomial `coeffs` (n coefficients, untouched) at ZETA extended_omega^(i parts + part), i < extended_len / parts.
        lazy_out: in the lazy-limb radix (x R' mod p) for evaluate_expression(lazy=True)"""
        self._part_check(part, parts)
        if int(coeffs.shape[0]) != self.n or int(out.shape[0]) != self.part_len(parts):
            raise AssertionError("assertion failed: coeffs.len() == n && out.len() == extended_len / parts")
        g, w = self._part_constants(part, parts)
        return ntt(self.field, out, w, stream=stream, coset_pre=g, in_log=self.k, src=coeffs, scale_by_n_inv=2 if lazy_out else 0)

    def part_to_coeff(self, a_part, part, parts, stream=0):
        """in place, the inverse of coeff_to_extended_part for a polynomial of degree < extended_len / parts =: m -- and for a
        longer one (the quotient: degree < extended_len) the sub-coset's FOLDED coefficients
            A_part[r] = sum_i (g^m)^i h[i m + r],   g = ZETA extended_omega^part   (g^m is the same for every point of the sub-coset)
        from which quotient_pieces_from_parts recovers h: inverse transform of size m, times 1 / m, times g^-r."""
        self._part_check(part, parts)
        if int(a_part.shape[0]) != self.part_len(parts):
            raise AssertionError("assertion failed: a.len() == extended_len / parts")
        key = ("inv", part, parts)
        if key not in self.__dict__.setdefault("_parts", {}):
            g, w = self._part_constants(part, parts)
            self._parts[key] = (field_inverse(self.field, g), field_inverse(self.field, w))
        ginv, winv = self._parts[key]
        return ntt(self.field, a_part, winv, scale_by_n_inv=True, stream=stream, coset_post=ginv, device=True)

    def part_mix(self, parts):
        """c[i][j] (Python integers) with  h[i m : (i + 1) m] = sum_j c[i][j] A_j  for the folded coefficients A_j of part_to_coeff:
        A_j = sum_i (ZETA^m theta^j)^i h^(i), theta = extended_omega^m a primitive parts-th root of unity, so the h^(i) are an inverse
        DFT of size `parts` over the sub-cosets, scaled: c[i][j] = ZETA^(-m i) theta^(-i j) / parts."""
        self._part_check(0, parts)
        p = self._p
        to_int = lambda a: sum(int(w) << (64 * i) for i, w in enumerate(a.tolist())) * pow(1 << 256, -1, p) % p
        m = self.part_len(parts)
        theta = pow(to_int(self.extended_omega), m, p)
        zm_inv = pow(pow(to_int(self.g_coset), m, p), -1, p)
        pinv = pow(parts, -1, p)
        return [[pinv * pow(zm_inv, i, p) * pow(theta, (-i * j) % parts, p) % p for j in range(parts)] for i in range(parts)]

    def piece_scalars(self, parts):
        """the quotient's pieces of n coefficients (upstream commits each: h(X) = sum_q X^(n q) h_q) as combinations of the n-coefficient
        slices of the folded sub-coset coefficients: piece q = i r + s (r = m / n slices per part) = sum_j c[i][j] A_j[s n : (s + 1) n].
        Returns [(q, [(j, s, c_ij), ...])]: by linearity the same combination of the slices' COMMITMENTS is the piece's commitment."""
        c = self.part_mix(parts)
        r = self.part_len(parts) // self.n
        return [(i * r + s, [(j, s, c[i][j]) for j in range(parts)]) for i in range(parts) for s in range(r)]

    def fold_scalars(self, parts, xn_int):
        """e[j][s] with  sum_q xn^q h_q = sum_{j, s} e[j][s] A_j[s n : (s + 1) n]  (the folded quotient the evaluation phase opens)"""
        c = self.part_mix(parts)
        p = self._p
        r = self.part_len(parts) // self.n
        xr = pow(xn_int, r, p)
        return [[pow(xn_int, s, p) * sum(pow(xr, i, p) * c[i][j] for i in range(parts)) % p for s in range(r)] for j in range(parts)]

    def divide_by_vanishing_poly_part(self, a_part, part, parts, stream=0):
        """a[i] *= t_evaluations[(i parts + part) mod 2^(extended_k - k)]"""
        self._part_check(part, parts)
        m = 1 << (self.extended_k - self.k)
        tbl = np.stack([self.t_evaluations[(i * parts + part) % m] for i in range(m // parts)])
        return vec_op(self.field, "scale_periodic", a_part, b=tbl, stream=stream)

    def quotient_len(self):
        return self.n * self.quotient_poly_degree

    def divide_by_vanishing_poly(self, a_ext, stream=0):
        """a[i] *= t_evaluations[i mod 2^(extended_k - k)]  (the inverse of X^n - 1 on the coset, periodic)"""
        if int(a_ext.shape[0]) != self.extended_len():
            raise AssertionError("assertion failed: a.values.len() == extended_len")
        return vec_op(self.field, "scale_periodic", a_ext, b=self.t_evaluations, stream=stream)


def combine_commitments(curve, points_jac, rows, to_device=None, stream=0):
    """sum_k scalar_k * points[index_k] for every row of `rows` = [[(index, integer scalar), ...], ...]: the commitments of the quotient's
    pieces from the commitments of the sub-cosets' folded coefficients (EvaluationDomain.piece_scalars) -- commitments are linear, so
    the pieces never have to exist as vectors before they are committed.  A batched MSM over a throwaway handle of len(points) bases
    (to_device: host array -> device buffer; without it one host-scalar MSM per row).  Returns [len(rows), 3 * limbs] Jacobian."""
    from . import point_to_affine
    aff = np.stack([point_to_affine(curve, pj) for pj in points_jac])
    cols = np.zeros((len(rows), len(points_jac), 4), dtype=np.uint64)
    for q, terms in enumerate(rows):
        for idx, sc in terms:
            cols[q, idx] = [(int(sc) >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)]
    bases = Bases(curve, aff)
    try:
        if to_device is not None:
            return msm_batch(bases, to_device(cols), stream=stream)
        return np.stack([msm(bases, cols[q]) for q in range(len(rows))])
    finally:
        bases.free()


# ------------------------------------------------------------------ prover steps beyond commit / FFT
class ExprOp(ctypes.Structure):
    _fields_ = [("op", ctypes.c_uint8), ("pad", ctypes.c_uint8), ("rot", ctypes.c_int16), ("arg", ctypes.c_uint32)]


EXPR_CODES = {"col": 0, "const": 1, "add": 2, "sub": 3, "mul": 4, "neg": 5, "scale": 6}


def _plib():
    lib = load()
    u64, vp, i32, u32 = ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32
    pp = ctypes.POINTER(ctypes.c_void_p)
    lib.zk_batch_invert_device.argtypes = [i32, vp, u64, vp]
    lib.zk_prefix_product_device.argtypes = [i32, vp, vp, u64, vp, vp, vp]
    lib.zk_halo2_permutation_product_device.argtypes = [i32, u32, pp, pp, u32, vp, vp, vp, u32, vp, vp, vp, vp]
    lib.zk_halo2_lookup_product_device.argtypes = [i32, vp, vp, vp, vp, vp, vp, u64, vp, vp, vp]
    lib.zk_inner_product_device.argtypes = [i32, vp, vp, u64, vp, vp]
    lib.zk_vec_fold_device.argtypes = [i32, vp, u64, vp, vp]
    lib.zk_ipa_fold_bases_device.argtypes = [i32, vp, u64, vp, vp]
    lib.zk_ipa_virtual_scalars_device.argtypes = [i32, vp, vp, u64, u64, vp, vp, vp]
    lib.zk_ipa_update_weights_device.argtypes = [i32, vp, u64, u64, vp, vp]
    lib.zk_ipa_collapse_device.argtypes = [i32, u64, vp, u64, u64, vp, vp]
    lib.zk_ipa_collapse_range_device.argtypes = [i32, u64, vp, u64, u64, u64, u64, vp, vp]
    lib.zk_ipa_round_device.argtypes = [i32, u64, vp, vp, vp, u64, u64, vp, vp, vp, vp]
    lib.zk_poly_eval_device.argtypes = [i32, vp, u64, vp, vp, vp]
    lib.zk_vec_muladd_device.argtypes = [i32, vp, vp, u64, vp, vp]
    lib.zk_vec_muladd_to_device.argtypes = [i32, vp, vp, vp, u64, vp, vp]
    lib.zk_poly_eval_batch_device.argtypes = [i32, vp, u64, ctypes.c_uint32, u64, vp, vp, vp]
    lib.zk_kate_division_device.argtypes = [i32, vp, vp, u64, vp, vp]
    lib.zk_vec_powers_device.argtypes = [i32, vp, u64, vp, vp]
    lib.zk_vec_fold_many_device.argtypes = [i32, vp, vp, ctypes.c_int64, u32, u64, vp, vp]
    lib.zk_ipa_fold_round_device.argtypes = [i32, vp, vp, vp, u64, u64, vp, vp]
    lib.zk_expr_eval_device.argtypes = [i32, ctypes.POINTER(ExprOp), u32, pp, u32, vp, u32, u32, u32, vp, vp]
    lib.zk_expr_eval_lazy_device.argtypes = [i32, ctypes.POINTER(ExprOp), u32, pp, u32, vp, u32, u32, u32, vp, vp]
    lib.zk_expr_configure.argtypes = [i32]
    lib.zk_expr_specialised_source.argtypes = [i32, ctypes.POINTER(ExprOp), u32, u32, u32, ctypes.c_char_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    return lib


def _ptr_array(bufs):
    return (ctypes.c_void_p * len(bufs))(*[ctypes.cast(_ptr(b), ctypes.c_void_p).value for b in bufs])


def permute_expression_pair(field, inputs_mont, table_mont, usable_rows):
    """plonk/lookup/prover.rs permute_expression_pair -- a CPU step upstream (a sort and a BTreeMap) and here: host arrays of
    Montgomery limbs [n, 4] in, (A', S') of usable_rows each out (the caller appends its blinding rows and uploads).  Values are
    ordered as canonical integers (the field's Ord); raises ValueError where upstream returns ConstraintSystemFailure."""
    p = field_modulus(field)
    r_inv = pow(1 << 256, -1, p)

    def canon(arr):            # Montgomery limbs -> canonical limbs, via Python integers (a host step; sizes are one column)
        a = _np64(arr)[:usable_rows].astype(object)
        v = (a[:, 0] + (a[:, 1] << 64) + (a[:, 2] << 128) + (a[:, 3] << 192)) * r_inv % p
        return np.stack([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], axis=1).astype(np.uint64)

    def to_mont(c):
        a = c.astype(object)
        v = ((a[:, 0] + (a[:, 1] << 64) + (a[:, 2] << 128) + (a[:, 3] << 192)) << 256) % p
        return np.stack([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], axis=1).astype(np.uint64)

    order = lambda c: np.lexsort((c[:, 0], c[:, 1], c[:, 2], c[:, 3]))     # most significant limb last = primary key
    ci, ct = canon(inputs_mont), canon(table_mont)
    a = ci[order(ci)]
    t = ct[order(ct)]
    first = np.ones(usable_rows, dtype=bool)
    first[1:] = (a[1:] != a[:-1]).any(axis=1)
    # the table's multiset: unique values and counts (t is sorted)
    tb = np.ones(len(t), dtype=bool)
    tb[1:] = (t[1:] != t[:-1]).any(axis=1)
    tvals = t[tb]
    tcnt = np.diff(np.append(np.flatnonzero(tb), len(t)))
    key = lambda c: [tuple(int(x) for x in row[::-1]) for row in c]
    index = {k: i for i, k in enumerate(key(tvals))}
    for k in key(a[first]):
        i = index.get(k)
        if i is None or tcnt[i] == 0:
            raise ValueError("ConstraintSystemFailure: a lookup input is not in the table")
        tcnt[i] -= 1
    s_perm = np.zeros((usable_rows, 4), dtype=np.uint64)
    s_perm[first] = a[first]
    leftovers = np.repeat(tvals, tcnt, axis=0)                     # ascending
    repeated = np.flatnonzero(~first)
    assert len(leftovers) == len(repeated)
    s_perm[repeated[::-1]] = leftovers                              # upstream pops the repeated rows from the last one down
    return to_mont(a), to_mont(s_perm)


def batch_invert(field, a, stream=0):
    """arithmetic.rs BatchInvert on a device buffer, in place; zeros stay zero"""
    _check(_plib().zk_batch_invert_device(field_id(field), _ptr(a), int(a.shape[0]), ctypes.c_void_p(stream)), "zk_batch_invert_device")
    return a


def prefix_product(field, a, out=None, first=None, want_total=False, stream=0):
    """out[i] = first * prod_{j<i} a[j] (in place when out is None) -> out, or (out, total)"""
    out = a if out is None else out
    tot = np.zeros(4, dtype=np.uint64)
    _check(_plib().zk_prefix_product_device(field_id(field), _ptr(a), _ptr(out), int(a.shape[0]), _ptr(_np64(first)) if first is not None else None,
                                            _ptr(tot) if want_total else None, ctypes.c_void_p(stream)), "zk_prefix_product_device")
    return (out, tot) if want_total else out


def permutation_product(field, columns, sigmas, beta, gamma, delta, k, z_out, first_column_index=0, z_first=None, stream=0):
    """plonk/permutation/prover.rs Argument::commit, one chunk (<= 8 columns): fills z_out with Z and returns the value
    after the last row (the next chunk's z_first)"""
    last = np.zeros(4, dtype=np.uint64)
    keep = [_np64(x) for x in (beta, gamma, delta)]
    zf = _np64(z_first) if z_first is not None else None
    _check(_plib().zk_halo2_permutation_product_device(field_id(field), len(columns), _ptr_array(columns), _ptr_array(sigmas),
                                                       first_column_index, _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), k,
                                                       _ptr(zf) if zf is not None else None, _ptr(z_out), _ptr(last), ctypes.c_void_p(stream)),
           "zk_halo2_permutation_product_device")
    return last


def lookup_product(field, a, s, a_perm, s_perm, beta, gamma, z_out, stream=0):
    """plonk/lookup/prover.rs commit_product; returns Z after the last row (must be 1 for a valid lookup)"""
    last = np.zeros(4, dtype=np.uint64)
    keep = [_np64(beta), _np64(gamma)]
    _check(_plib().zk_halo2_lookup_product_device(field_id(field), _ptr(a), _ptr(s), _ptr(a_perm), _ptr(s_perm), _ptr(keep[0]), _ptr(keep[1]),
                                                  int(a.shape[0]), _ptr(z_out), _ptr(last), ctypes.c_void_p(stream)), "zk_halo2_lookup_product_device")
    return last


def inner_product(field, a, b, stream=0):
    out = np.zeros(4, dtype=np.uint64)
    _check(_plib().zk_inner_product_device(field_id(field), _ptr(a), _ptr(b), int(a.shape[0]), _ptr(out), ctypes.c_void_p(stream)),
           "zk_inner_product_device")
    return out


def vec_muladd(field, a, b, s, stream=0, out=None):
    """a[i] = a[i] * s + b[i] (multiopen: folding the polynomials of a point set with powers of x_1); out: write there instead
    and leave a alone (upstream clones the first polynomial of a fold)"""
    ss = _np64(s)
    if out is not None:
        _check(_plib().zk_vec_muladd_to_device(field_id(field), _ptr(out), _ptr(a), _ptr(b), int(a.shape[0]), _ptr(ss), ctypes.c_void_p(stream)),
               "zk_vec_muladd_to_device")
        return out
    _check(_plib().zk_vec_muladd_device(field_id(field), _ptr(a), _ptr(b), int(a.shape[0]), _ptr(ss), ctypes.c_void_p(stream)), "zk_vec_muladd_device")
    return a


def eval_polynomial(field, d_poly, x, stream=0):
    """arithmetic.rs eval_polynomial: p(x) for a device-resident coefficient vector; x, result: Montgomery limbs"""
    xx = _np64(x)
    out = np.zeros(4, dtype=np.uint64)
    _check(_plib().zk_poly_eval_device(field_id(field), _ptr(d_poly), int(d_poly.shape[0]), _ptr(xx), _ptr(out), ctypes.c_void_p(stream)),
           "zk_poly_eval_device")
    return out


def eval_polynomials(field, d_polys, x, stream=0):
    """d_polys: device buffer [count, n, 4]; every polynomial at the same x in one launch -> [count, 4]"""
    xx = _np64(x)
    count, n = int(d_polys.shape[0]), int(d_polys.shape[1])
    out = np.zeros((count, 4), dtype=np.uint64)
    _check(_plib().zk_poly_eval_batch_device(field_id(field), _ptr(d_polys), n, count, n, _ptr(xx), _ptr(out), ctypes.c_void_p(stream)),
           "zk_poly_eval_batch_device")
    return out


def vec_fold_many(field, out, polys, s, stream=0, reverse=False):
    """out[j] = sum_i s^(count - 1 - i) polys[i][j] (Horner over the rows of `polys` [count, n, 4], one pass); reverse: walk the rows
    from the last to the first (h(X) = sum_i (x^n)^i h_i: the last piece is the leading one)"""
    ss = _np64(s)
    count, n = int(polys.shape[0]), int(polys.shape[1])
    first = polys[count - 1] if reverse else polys[0]
    _check(_plib().zk_vec_fold_many_device(field_id(field), _ptr(out), _ptr(first), -n if reverse else n, count, n, _ptr(ss), ctypes.c_void_p(stream)),
           "zk_vec_fold_many_device")
    return out


def ipa_fold_round(field, p, b, half, u, w=None, m0=0, stream=0):
    """the three folds of one argument round in one launch (w: the fold-free form's weight vector over m0 generators)"""
    uu = _np64(u)
    _check(_plib().zk_ipa_fold_round_device(field_id(field), _ptr(p), _ptr(b), _ptr(w) if w is not None else None, half, m0, _ptr(uu),
                                            ctypes.c_void_p(stream)), "zk_ipa_fold_round_device")


def vec_powers(field, out, x, stream=0):
    """out[i] = x^i (the vector b of the inner-product argument: powers of x_3)"""
    xx = _np64(x)
    _check(_plib().zk_vec_powers_device(field_id(field), _ptr(out), int(out.shape[0]), _ptr(xx), ctypes.c_void_p(stream)), "zk_vec_powers_device")
    return out


def kate_division(field, a, x, out=None, stream=0):
    """arithmetic.rs kate_division: (a(X) - a(x)) / (X - x) as len(a) coefficients (the last one zero: upstream returns one fewer
    and the multiopen prover resizes); in place unless `out` is given"""
    xx = _np64(x)
    out = a if out is None else out
    _check(_plib().zk_kate_division_device(field_id(field), _ptr(a), _ptr(out), int(a.shape[0]), _ptr(xx), ctypes.c_void_p(stream)),
           "zk_kate_division_device")
    return out


def vec_fold(field, a, half, c, stream=0):
    """a[i] += c a[i + half] for i < half"""
    cc = _np64(c)
    _check(_plib().zk_vec_fold_device(field_id(field), _ptr(a), half, _ptr(cc), ctypes.c_void_p(stream)), "zk_vec_fold_device")
    return a


def ipa_fold_bases(curve, g, half, u, stream=0):
    """parallel_generator_collapse: g[i] <- affine(g[i] + [u] g[i + half]); g: device buffer [2 * half, 2 * limbs]"""
    uu = _np64(u)
    _check(_plib().zk_ipa_fold_bases_device(curve_id(curve), _ptr(g), half, _ptr(uu), ctypes.c_void_p(stream)), "zk_ipa_fold_bases_device")
    return g


def evaluate_expression(field, program, columns, consts, log_n_ext, rot_scale, out, stream=0, lazy=False):
    """program: list of ("col", column, rotation) / ("const", i) / ("add",) / ("sub",) / ("mul",) / ("neg",) / ("scale", i).
    lazy: the columns hold x R' mod p (coeff_to_extended(..., lazy_out=True) / to_lazy_form) and the evaluation runs on lazy
    29-bit limbs; constants and the output stay in the usual Montgomery form"""
    ops = (ExprOp * len(program))()
    for k, o in enumerate(program):
        ops[k].op = EXPR_CODES[o[0]]
        ops[k].rot = o[2] if o[0] == "col" else 0
        ops[k].arg = o[1] if len(o) > 1 else 0
    cs = _np64(consts).reshape(-1, 4) if len(consts) else np.zeros((1, 4), dtype=np.uint64)
    fn = _plib().zk_expr_eval_lazy_device if lazy else _plib().zk_expr_eval_device
    _check(fn(field_id(field), ops, len(program), _ptr_array(columns), len(columns), _ptr(cs), len(consts), log_n_ext,
              rot_scale, _ptr(out), ctypes.c_void_p(stream)), "zk_expr_eval_lazy_device" if lazy else "zk_expr_eval_device")
    return out


def expr_configure(jit="auto"):
    """zk_expr_configure: the lazy evaluator's specialised (hiprtc-compiled) kernel: "auto" (2^16 rows and more), "always", "never" """
    _check(_plib().zk_expr_configure({"auto": 0, "always": 1, "never": 2}[jit]), "zk_expr_configure")


def expr_specialised_source(field, program, n_columns, n_consts):
    """the HIP source of the kernel zk_expr_eval_lazy_device compiles for `program` (no device needed)"""
    ops = (ExprOp * len(program))()
    for i, o in enumerate(program):
        ops[i].op = EXPR_CODES[o[0]]
        ops[i].rot = int(o[2]) if o[0] == "col" and len(o) > 2 else 0
        ops[i].arg = int(o[1]) if len(o) > 1 else 0
    n = ctypes.c_uint64(0)
    fid = field_id(field)
    _check(_plib().zk_expr_specialised_source(fid, ops, len(program), n_columns, n_consts, None, 0, ctypes.byref(n)), "zk_expr_specialised_source")
    buf = ctypes.create_string_buffer(n.value + 1)
    _check(_plib().zk_expr_specialised_source(fid, ops, len(program), n_columns, n_consts, buf, n.value + 1, ctypes.byref(n)), "zk_expr_specialised_source")
    return buf.value.decode()


def to_lazy_form(field, a, stream=0):
    """x R -> x R' (R' = 2^261 for the 256-bit fields: times 2^5) in place: key material (fixed columns on the extended coset) for
    evaluate_expression(lazy=True)"""
    p = field_modulus(field)
    return vec_op(field, "scale", a, scalar=_mont_limbs(32, p), stream=stream)


class IpaProver:
    """poly/commitment/prover.rs create_proof, the k rounds of the inner-product argument on device buffers.  The caller
    owns the transcript: `round()` returns L_j, R_j (before the U / W blinding terms) and <p'_hi, b_lo>, <p'_lo, b_hi>;
    after hashing them into its transcript the caller feeds the challenge to `fold(u_j)`."""

    def __init__(self, curve, d_p, d_b, d_g, stream=0):
        """d_p, d_b: device buffers [n, 4] (p' coefficients and the powers of x_3, Montgomery); d_g: [n, 2 * limbs] affine
        generators (a working copy: it is folded in place).  The generator vector is adopted ONCE as a bases handle; every
        round's two MSMs address its halves through zk_msm_opts.base_offset, and the fold refreshes the handle's derived
        copy for the half that survives."""
        self.curve, self.field = curve_id(curve), scalar_field(curve)
        self.p, self.b, self.g, self.stream = d_p, d_b, d_g, stream
        self.n = int(d_p.shape[0])
        assert self.n & (self.n - 1) == 0 and int(d_b.shape[0]) == self.n and int(d_g.shape[0]) == self.n
        self.bases = Bases(self.curve, device_tensor=d_g, n=self.n)

    def round(self, sharded=False):
        """sharded: every rank of a torch.distributed job holds the same p', b, G' and sums its own scalar windows of the
        two MSMs; one all_gather combines them (contangle-zkcp_amd/dist.py)"""
        half = self.n // 2
        if sharded:
            from . import dist as zkdist
            L, R = zkdist.msm_many_sharded([(self.bases, self.p[half:self.n], True, 0), (self.bases, self.p[:half], True, half)], stream=self.stream)
        else:
            tl = msm_submit(self.bases, self.p[half:self.n], montgomery=True, stream=self.stream)                    # L = <p'_hi, G'_lo>
            tr = msm_submit(self.bases, self.p[:half], montgomery=True, stream=self.stream, base_offset=half)        # R = <p'_lo, G'_hi>
        vl = inner_product(self.field, self.p[half:self.n], self.b[:half], stream=self.stream)
        vr = inner_product(self.field, self.p[:half], self.b[half:self.n], stream=self.stream)
        if not sharded:
            L, R = tl.collect(), tr.collect()
        return L, R, vl, vr

    def fold(self, u):
        half = self.n // 2
        vec_fold(self.field, self.p, half, field_inverse(self.field, u), stream=self.stream)
        vec_fold(self.field, self.b, half, u, stream=self.stream)
        ipa_fold_bases(self.curve, self.g, half, u, stream=self.stream)
        self.bases.refresh(0, half, stream=self.stream)
        self.n = half

    def free(self):
        self.bases.free()


class IpaProverVirtual:
    """The same argument with the generators left alone: every round's L, R are MSMs over the ORIGINAL generators (a resident
    `Bases`, e.g. the SRS itself) with scalars p'[i] W[idx]; W collects the challenges.  2 n-point MSMs (half of the scalars
    zero) per round in one batched call, instead of a 255-bit scalar multiplication per surviving generator.

    `collapse()` materialises the generators that the rounds so far would have left (zk_ipa_collapse_device: the
    surviving points as multi-scalar multiplications that share their scalars) and continues over them: the first rounds
    cost one full-size MSM each, the remaining ones only what their own size costs."""

    def __init__(self, curve, d_p, d_b, bases, new_buffer, stream=0, buffers=None):
        """new_buffer(shape) -> zero device buffer (torch on the GPU, numpy under the test emulator).
        buffers = (S [2, n, 4], W [n, 4]): caller-owned scratch reused across proofs (no allocation inside the argument);
        every entry is rewritten on the stream before it is read"""
        self.curve, self.field = curve_id(curve), scalar_field(curve)
        self.p, self.b, self.bases, self.stream = d_p, d_b, bases, stream
        self.new_buffer = new_buffer
        self.m0 = self.n = int(d_p.shape[0])
        assert self.n & (self.n - 1) == 0 and int(d_b.shape[0]) == self.n and bases.n >= self.n
        self._own_bases = None
        self._g = None
        self._buffers = buffers
        if buffers is not None:
            assert int(buffers[0].shape[0]) == 2 and int(buffers[0].shape[1]) >= self.n and int(buffers[1].shape[0]) >= self.n
        self._fresh_weights()

    def _fresh_weights(self):
        one = _mont_limbs(1, field_modulus(self.field))
        if self._buffers is not None:       # prefixes of the caller's buffers; W = 1 everywhere as the powers of 1, on the stream
            self.S = self._buffers[0].reshape(-1)[: 2 * self.m0 * 4].reshape(2, self.m0, 4)
            self.W = self._buffers[1][: self.m0]
            vec_powers(self.field, self.W, one, stream=self.stream)
            return
        self.S = self.new_buffer((2, self.m0, 4))
        self.W = self.new_buffer((self.m0, 4))
        ones = np.tile(one, (self.m0, 1))
        if isinstance(self.W, np.ndarray):
            self.W[:] = ones
        else:
            import torch
            self.W.copy_(torch.from_numpy(ones.view(np.int64)))

    def round(self, sharded=False):
        half = self.n // 2
        if sharded:
            from . import dist as zkdist
            sharded = self.m0 >= zkdist.SHARD_MIN_POINTS      # small rounds: every rank computes them whole, no collective
        if not sharded:     # the whole round in one call: scalars, inner products and both MSMs enqueued together
            nl = _plib().zk_curve_base_limbs64(self.curve)
            lr = np.zeros((2, 3 * nl), dtype=np.uint64)
            v = np.zeros((2, 4), dtype=np.uint64)
            _check(_plib().zk_ipa_round_device(self.curve, self.bases.handle, _ptr(self.p), _ptr(self.b), _ptr(self.W), self.m0, self.n,
                                               _ptr(self.S), _ptr(lr), _ptr(v), ctypes.c_void_p(self.stream)), "zk_ipa_round_device")
            return lr[0], lr[1], v[0], v[1]
        _check(_plib().zk_ipa_virtual_scalars_device(self.field, _ptr(self.p), _ptr(self.W), self.m0, self.n, _ptr(self.S[0]), _ptr(self.S[1]),
                                                     ctypes.c_void_p(self.stream)), "zk_ipa_virtual_scalars_device")
        L, R = zkdist.msm_batch_sharded(self.bases, self.S, montgomery=True, stream=self.stream)
        vl = inner_product(self.field, self.p[half:self.n], self.b[:half], stream=self.stream)
        vr = inner_product(self.field, self.p[:half], self.b[half:self.n], stream=self.stream)
        return L, R, vl, vr

    def fold(self, u):
        half = self.n // 2
        if half < self.m0:
            ipa_fold_round(self.field, self.p, self.b, half, u, w=self.W, m0=self.m0, stream=self.stream)
        else:      # (a single generator left over: nothing to weight)
            ipa_fold_round(self.field, self.p, self.b, half, u, stream=self.stream)
        self.n = half

    def collapse(self, sharded=False):
        """G' of the rounds done so far, as a device buffer [n, 2 * limbs] of affine points; the later rounds run over it.
        sharded: the ranks of a torch.distributed job (all holding the same state) compute n / world survivors each and
        exchange them with one all_gather."""
        nl = _plib().zk_curve_base_limbs64(self.curve)
        g = self.new_buffer((self.n, 2 * nl))
        world, rank = 1, 0
        if sharded:
            import torch.distributed as tdist
            if tdist.is_initialized():
                world, rank = tdist.get_world_size(), tdist.get_rank()
        if world > 1 and self.n % world == 0 and self.n // world >= 64:
            import torch
            share = self.n // world
            mine = g[rank * share:(rank + 1) * share]
            _check(_plib().zk_ipa_collapse_range_device(self.curve, self.bases.handle, _ptr(self.W), self.m0, self.n, rank * share, share,
                                                        _ptr(mine), ctypes.c_void_p(self.stream)), "zk_ipa_collapse_range_device")
            # (the call above synchronises the stream: `mine` is complete)
            if isinstance(g, np.ndarray):                                   # the CPU test emulator: "device" memory is host memory
                t = torch.from_numpy(np.ascontiguousarray(mine).view(np.int64))
                outs = [torch.empty_like(t) for _ in range(world)]
                tdist.all_gather(outs, t)
                for r, o in enumerate(outs):
                    g[r * share:(r + 1) * share] = o.numpy().view(np.uint64).reshape(share, 2 * nl)
            elif tdist.get_backend() == "nccl":                             # RCCL over xGMI, device to device (the same
                t = mine.reshape(-1).clone()                                # all_gather form as dist._gather_add)
                outs = [torch.empty_like(t) for _ in range(world)]
                tdist.all_gather(outs, t)
                g.view(-1).copy_(torch.cat(outs))
                torch.cuda.synchronize()                                    # g is adopted as a bases handle next (another stream)
            else:                                                           # gloo rehearsal on a GPU box: staged through the host
                t = mine.cpu()
                outs = [torch.empty_like(t) for _ in range(world)]
                tdist.all_gather(outs, t)
                g.copy_(torch.cat(outs).to(g.device))
                torch.cuda.synchronize()
        else:
            _check(_plib().zk_ipa_collapse_device(self.curve, self.bases.handle, _ptr(self.W), self.m0, self.n, _ptr(g), ctypes.c_void_p(self.stream)),
                   "zk_ipa_collapse_device")
        if self._own_bases is not None:
            self._own_bases.free()
        self._g = g                                   # adopted, not copied: kept alive here
        self._own_bases = self.bases = Bases(self.curve, device_tensor=g, n=self.n)
        self.m0 = self.n
        self._fresh_weights()
        return g

    def free(self):
        if self._own_bases is not None:
            self._own_bases.free()
            self._own_bases = None

    def folded_generator(self):
        """G' after the rounds so far are all done (n == 1): MSM(G0, W) -- what upstream's collapsed g_prime[0] is"""
        return msm(self.bases, self.W, montgomery=True, stream=self.stream)
