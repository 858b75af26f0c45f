"""TEST INFRASTRUCTURE -- ctypes binding of oracle/libzkoracle.so (the CPU oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
PARITY UNPINNED (see zk_oracle_impl.h): pinned by oracle/pyref.py fixtures, not by
reference outputs.

Arrays are numpy uint64, little-endian limbs: field elements [n, L], affine points
[n, 2L] (x limbs then y limbs, infinity = all zero), scalars [n, 4].
"""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libzkoracle.so")

FIELD_IDS = ["PallasFp", "PallasFq", "Bn254Fr", "Bls381Fr", "Bn254Fq", "Bls381Fq"]
CURVE_IDS = ["Pallas", "Vesta", "Bn254G1", "Bls381G1", "Bn254G2", "Bls381G2"]


def build(force=False):
    # make decides what is stale (the prebuilt library travels to the GPU box with the snapshot; a box without make keeps it)
    try:
        subprocess.check_call(["make", "-s", "-C", _DIR] + (["-B"] if force else []))
    except (OSError, subprocess.CalledProcessError):
        if not os.path.exists(_SO):
            raise


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_field_name.restype = ctypes.c_char_p
        _lib.orc_curve_name.restype = ctypes.c_char_p
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def fid(field):
    return field if isinstance(field, int) else FIELD_IDS.index(field)


def cid(curve):
    return curve if isinstance(curve, int) else CURVE_IDS.index(curve)


def field_nlimbs(field):
    return lib().orc_field_nlimbs(fid(field))


def curve_base_field(curve):
    return lib().orc_curve_base_field(cid(curve))


def curve_scalar_field(curve):
    return lib().orc_curve_scalar_field(cid(curve))


def coord_limbs(curve):
    """u64 limbs per point coordinate: Fq limbs, times 2 on the G2 curves (Fq2 = c0 | c1)."""
    return lib().orc_curve_coord_limbs(cid(curve))


def int_to_limbs(x, nl):
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nl)], dtype=np.uint64)


def limbs_to_int(a):
    v = 0
    for i, w in enumerate(np.asarray(a, dtype=np.uint64).tolist()):
        v |= int(w) << (64 * i)
    return v


def ints_to_array(xs, nl):
    return np.stack([int_to_limbs(x, nl) for x in xs]) if len(xs) else np.zeros((0, nl), dtype=np.uint64)


def array_to_ints(a):
    return [limbs_to_int(r) for r in a]


def modulus(field):
    nl = field_nlimbs(field)
    out = np.zeros(nl, dtype=np.uint64)
    lib().orc_field_modulus(fid(field), _p(out))
    return limbs_to_int(out)


def fe_op(field, op, a, b=None):
    ops = {"add": 0, "sub": 1, "mul": 2, "inv": 3, "to_mont": 4, "from_mont": 5, "neg": 6}
    nl = field_nlimbs(field)
    a = _u64(a)
    out = np.zeros(nl, dtype=np.uint64)
    bb = _u64(b) if b is not None else None
    lib().orc_fe_op(fid(field), ops[op], _p(a), _p(bb) if bb is not None else None, _p(out))
    return out


def to_mont(field, a):
    a = _u64(a)
    out = np.empty_like(a)
    lib().orc_fe_batch_convert(fid(field), 1, _p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def from_mont(field, a):
    a = _u64(a)
    out = np.empty_like(a)
    lib().orc_fe_batch_convert(fid(field), 0, _p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def root_of_unity(field, logn):
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_root_of_unity(fid(field), logn, _p(out))
    return out


def field_generator(field):
    out = np.zeros(field_nlimbs(field), dtype=np.uint64)
    lib().orc_field_generator(fid(field), _p(out))
    return out


def curve_generator(curve):
    nl = coord_limbs(curve)
    out = np.zeros(2 * nl, dtype=np.uint64)
    lib().orc_curve_generator(cid(curve), _p(out))
    return out


def on_curve(curve, aff):
    return bool(lib().orc_on_curve(cid(curve), _p(_u64(aff))))


def scalar_mul(curve, aff, k):
    aff = _u64(aff)
    out = np.zeros_like(aff)
    lib().orc_scalar_mul(cid(curve), _p(aff), _p(_u64(k)), _p(out))
    return out


def point_add(curve, a, b):
    a = _u64(a)
    out = np.zeros_like(a)
    lib().orc_point_add(cid(curve), _p(a), _p(_u64(b)), _p(out))
    return out


def jac_to_affine(curve, jac):
    jac = _u64(jac)
    nl = jac.shape[-1] // 3
    out = np.zeros(2 * nl, dtype=np.uint64)
    lib().orc_jac_to_affine(cid(curve), _p(jac), _p(out))
    return out


def fixed_base_mul(curve, scalars, threads=8):
    """P_i = [k_i]G (affine) for canonical scalars [n,4]."""
    scalars = _u64(scalars)
    nl = coord_limbs(curve)
    out = np.zeros((scalars.shape[0], 2 * nl), dtype=np.uint64)
    lib().orc_fixed_base_mul(cid(curve), _p(scalars), ctypes.c_size_t(scalars.shape[0]), threads, _p(out))
    return out


def _msm(fn, curve, bases, scalars, *extra):
    bases, scalars = _u64(bases), _u64(scalars)
    assert bases.shape[0] == scalars.shape[0]
    nl = coord_limbs(curve)
    out = np.zeros(2 * nl, dtype=np.uint64)
    fn(cid(curve), _p(bases), _p(scalars), ctypes.c_size_t(bases.shape[0]), *extra, _p(out))
    return out


def msm_naive(curve, bases, scalars_canonical):
    return _msm(lib().orc_msm_naive, curve, bases, scalars_canonical)


def msm_ark(curve, bases, scalars_canonical, threads=1):
    """ark-ec 0.3 VariableBaseMSM::multi_scalar_mul restatement -> affine."""
    return _msm(lib().orc_msm_ark, curve, bases, scalars_canonical, threads)


def msm_halo2(curve, bases, scalars_mont, threads=1):
    """halo2_proofs 0.2 best_multiexp restatement -> affine."""
    return _msm(lib().orc_msm_halo2, curve, bases, scalars_mont, threads)


def ark_window_bits(n):
    return lib().orc_msm_ark_window_bits(ctypes.c_size_t(n))


def dft_naive(field, a, omega_mont):
    a = _u64(a)
    out = np.empty_like(a)
    lib().orc_dft_naive(fid(field), _p(a), _p(out), ctypes.c_size_t(a.shape[0]), _p(_u64(omega_mont)))
    return out


def halo2_best_fft(field, a, omega_mont, logn, threads=1):
    a = _u64(a).copy()
    assert a.shape[0] == 1 << logn
    lib().orc_halo2_best_fft(fid(field), _p(a), _p(_u64(omega_mont)), logn, threads)
    return a


ARK_KINDS = {"fft": 0, "ifft": 1, "coset_fft": 2, "coset_ifft": 3}


def ark_fft(field, a, kind, threads=1):
    """ark-poly 0.3 Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place restatement."""
    a = _u64(a).copy()
    logn = int(a.shape[0]).bit_length() - 1
    assert a.shape[0] == 1 << logn
    lib().orc_ark_fft(fid(field), _p(a), logn, ARK_KINDS[kind], threads)
    return a


def distribute_powers(field, a, g_mont):
    a = _u64(a).copy()
    lib().orc_distribute_powers(fid(field), _p(a), ctypes.c_size_t(a.shape[0]), _p(_u64(g_mont)))
    return a


def groth16_witness_map(field, a, b, c, threads=1):
    """ark-groth16 0.3 R1CStoQAP::witness_map restatement from the evaluation vectors; returns h (m coefficients)."""
    a, b, c = _u64(a).copy(), _u64(b).copy(), _u64(c).copy()
    logm = int(a.shape[0]).bit_length() - 1
    lib().orc_groth16_witness_map(fid(field), _p(a), _p(b), _p(c), logm, threads)
    return a
