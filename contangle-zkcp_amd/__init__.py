"""contangle-zkcp_amd -- MI355X-native MSM / NTT backend for Contangle's prover path.

Host-side mirror (Python, ctypes over the C ABI of include/zkcp_amd.h) of the upstream
interfaces the reference's prover reaches:

  ark_ec.VariableBaseMSM.multi_scalar_mul   <- ark-ec 0.3 msm/variable_base.rs
  ark_poly.Radix2EvaluationDomain           <- ark-poly 0.3 domain/radix2/mod.rs
  halo2.best_multiexp / halo2.best_fft      <- halo2_proofs 0.2 arithmetic.rs

The compute path is the HIP library `libzkcp_amd.so` only.  There is no CPU fallback: if the
library is missing or no MI355X is visible, `load()` / `init()` raise.
"""
import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libzkcp_amd.so")

# zk_curve_t / zk_field_t
PALLAS, VESTA, BN254_G1, BLS12_381_G1, BN254_G2, BLS12_381_G2 = 0, 1, 2, 3, 4, 5
FP_PALLAS, FQ_PALLAS, FR_BN254, FR_BLS12_381 = 0, 1, 2, 3
CURVE_NAMES = {"Pallas": PALLAS, "Vesta": VESTA, "Bn254G1": BN254_G1, "Bls381G1": BLS12_381_G1,
               "Bn254G2": BN254_G2, "Bls381G2": BLS12_381_G2}
FIELD_NAMES = {"PallasFp": FP_PALLAS, "PallasFq": FQ_PALLAS, "Bn254Fr": FR_BN254, "Bls381Fr": FR_BLS12_381}

EXPORTS = [
    "zk_init", "zk_shutdown", "zk_strerror", "zk_backend_info", "zk_field_limbs64", "zk_curve_base_limbs64",
    "zk_curve_scalar_field", "zk_msm_window_bits", "zk_msm_window_count", "zk_bases_upload", "zk_bases_adopt_device",
    "zk_bases_free", "zk_msm", "zk_msm_device", "zk_msm_last_profile", "zk_ntt", "zk_ntt_device", "zk_coset_mul",
    "zk_coset_mul_device", "zk_ntt_coset_device", "zk_field_root_of_unity", "zk_field_multiplicative_generator", "zk_field_inverse",
    "zk_point_add", "zk_point_to_affine", "zk_fixed_base_mul_device", "zk_vec_op_device", "zk_groth16_witness_map_device",
    "zk_fixed_base_msm_device", "zk_ntt_extend_device", "zk_init_devices", "zk_device_count", "zk_msm_submit", "zk_msm_collect",
    "zk_msm_batch_device", "zk_ntt_configure", "zk_msm_profile_totals", "zk_ntt_profile_enable", "zk_ntt_profile_read",
    "zk_field_modulus", "zk_vec_scale_periodic_device", "zk_bases_refresh", "zk_bases_precompute", "zk_ntt_oop_device",
]


class ZkError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__("%s failed: %s (%d)" % (what, _strerror(status), status))


class MsmOpts(ctypes.Structure):
    """zk_msm_opts: zero = defaults.  split_log_plus1 = k + 1 forces 2^k pieces per bucket."""
    _fields_ = [("window_bits", ctypes.c_int), ("window_begin", ctypes.c_int), ("window_end", ctypes.c_int),
                ("limb_bits", ctypes.c_int), ("split_log_plus1", ctypes.c_int), ("slice_len", ctypes.c_int),
                ("big_threshold", ctypes.c_int), ("waves_per_simd", ctypes.c_int), ("flags", ctypes.c_int),
                ("base_offset", ctypes.c_int), ("window_group", ctypes.c_int), ("reserved", ctypes.c_int)]


MSM_FLAG_NO_HOT_HELP = 1
MSM_FLAG_SLICE_REDUCE = 2
MSM_FLAG_PRECOMPUTED = 4
MSM_FLAG_DEVICE_PARTIALS = 8
MSM_FLAG_OWN_STREAM = 16


class NttOpts(ctypes.Structure):
    _fields_ = [("max_log_radix", ctypes.c_int), ("log_tile_plus1", ctypes.c_int), ("block", ctypes.c_int),
                ("limb_bits", ctypes.c_int)]


class MsmProfile(ctypes.Structure):
    _fields_ = [(k, ctypes.c_float) for k in ("digits_ms", "hist_ms", "scatter_ms", "accumulate_ms", "reduce_ms",
                                              "host_tail_ms", "total_ms")] + \
               [(k, ctypes.c_int) for k in ("window_bits", "windows_total", "windows_done", "groups", "limb_bits")] + \
               [("accumulate_kernel_ms", ctypes.c_float), ("reserved", ctypes.c_int)]


class MsmTotals(ctypes.Structure):
    _fields_ = [("msms", ctypes.c_uint64)] + [(k, ctypes.c_double) for k in (
        "accumulate_kernel_ms", "accumulate_ms", "sort_ms", "reduce_ms", "host_tail_ms", "device_ms", "algorithmic_bytes")] + [
        ("launches", ctypes.c_uint64)]


class NttTotals(ctypes.Structure):
    _fields_ = [("transforms", ctypes.c_uint64), ("launches", ctypes.c_uint64), ("kernel_ms", ctypes.c_double),
                ("algorithmic_bytes", ctypes.c_double)]


_lib = None


def load(path=None):
    """dlopen the HIP library (or, for the CPU test tier only, an explicitly given emulator build)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError("HIP extension %s is missing -- run `python contangle-zkcp_amd/build.py` "
                           "(there is no CPU fallback)" % p)
    lib = ctypes.CDLL(p)
    lib.zk_strerror.restype = ctypes.c_char_p
    u64, vp, i32 = ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int
    lib.zk_bases_upload.argtypes = [i32, vp, u64, ctypes.POINTER(u64)]
    lib.zk_bases_adopt_device.argtypes = [i32, vp, u64, ctypes.POINTER(u64)]
    lib.zk_bases_free.argtypes = [u64]
    lib.zk_bases_refresh.argtypes = [u64, u64, u64, vp]
    lib.zk_bases_precompute.argtypes = [u64, i32]
    lib.zk_msm.argtypes = [i32, u64, vp, u64, i32, ctypes.POINTER(MsmOpts), vp]
    lib.zk_msm_device.argtypes = [i32, u64, vp, u64, i32, ctypes.POINTER(MsmOpts), vp, vp]
    lib.zk_msm_last_profile.argtypes = [ctypes.POINTER(MsmProfile)]
    lib.zk_msm_submit.argtypes = [i32, u64, vp, u64, i32, ctypes.POINTER(MsmOpts), vp, ctypes.POINTER(u64)]
    lib.zk_msm_collect.argtypes = [u64, vp]
    lib.zk_msm_batch_device.argtypes = [i32, u64, vp, u64, ctypes.c_uint32, u64, i32, ctypes.POINTER(MsmOpts), vp, vp]
    lib.zk_init_devices.argtypes = [i32, ctypes.POINTER(i32)]
    lib.zk_ntt_configure.argtypes = [ctypes.POINTER(NttOpts)]
    lib.zk_msm_profile_totals.argtypes = [ctypes.POINTER(MsmTotals), i32]
    lib.zk_ntt_profile_read.argtypes = [ctypes.POINTER(NttTotals)]
    lib.zk_field_modulus.argtypes = [i32, vp]
    lib.zk_vec_scale_periodic_device.argtypes = [i32, vp, u64, vp, ctypes.c_uint32, vp]
    lib.zk_ntt.argtypes = [i32, vp, ctypes.c_uint32, vp, i32]
    lib.zk_ntt_device.argtypes = [i32, vp, ctypes.c_uint32, vp, i32, vp]
    lib.zk_ntt_coset_device.argtypes = [i32, vp, ctypes.c_uint32, vp, i32, vp, vp, vp]
    lib.zk_ntt_extend_device.argtypes = [i32, vp, ctypes.c_uint32, ctypes.c_uint32, vp, i32, vp, vp, vp]
    lib.zk_ntt_oop_device.argtypes = [i32, vp, vp, ctypes.c_uint32, ctypes.c_uint32, vp, i32, vp, vp, vp]
    lib.zk_coset_mul.argtypes = [i32, vp, ctypes.c_uint32, vp]
    lib.zk_coset_mul_device.argtypes = [i32, vp, ctypes.c_uint32, vp, vp]
    lib.zk_field_root_of_unity.argtypes = [i32, ctypes.c_uint32, vp]
    lib.zk_field_multiplicative_generator.argtypes = [i32, vp]
    lib.zk_field_inverse.argtypes = [i32, vp, vp]
    lib.zk_point_add.argtypes = [i32, vp, vp, vp]
    lib.zk_point_to_affine.argtypes = [i32, vp, vp]
    lib.zk_fixed_base_mul_device.argtypes = [i32, vp, u64, vp, vp]
    lib.zk_fixed_base_msm_device.argtypes = [i32, vp, vp, u64, i32, vp, vp]
    lib.zk_vec_op_device.argtypes = [i32, i32, vp, vp, vp, u64, vp, vp]
    lib.zk_groth16_witness_map_device.argtypes = [i32, vp, vp, vp, ctypes.c_uint32, vp]
    lib.zk_msm_window_bits.argtypes = [i32, u64, i32]
    lib.zk_msm_window_count.argtypes = [i32, u64, i32]
    lib.zk_backend_info.argtypes = [ctypes.c_char_p, u64]
    _lib = lib
    return lib


def _strerror(status):
    return _lib.zk_strerror(status).decode() if _lib is not None else "?"


def _check(status, what):
    if status != 0:
        raise ZkError(status, what)


def init(device_id=0):
    _check(load().zk_init(device_id), "zk_init")


def init_devices(device_ids):
    """one process drives several GPUs: bases go to every device, whole MSMs are split over them by scalar window"""
    ids = (ctypes.c_int * len(device_ids))(*device_ids)
    _check(load().zk_init_devices(len(device_ids), ids), "zk_init_devices")


def device_count():
    return load().zk_device_count()


def ntt_configure(max_log_radix=0, log_tile=None, block=0, limb_bits=0):
    """process-wide NTT plan knobs (tuning harness / tests); no arguments restores the defaults.
    limb_bits: 0 = lazy 29-bit limbs inside the tiles (default), 32 = saturated words"""
    o = NttOpts(max_log_radix, 0 if log_tile is None else log_tile + 1, block, limb_bits)
    _check(load().zk_ntt_configure(ctypes.byref(o)), "zk_ntt_configure")


def shutdown():
    if _lib is not None:
        _check(_lib.zk_shutdown(), "zk_shutdown")


def backend_info():
    buf = ctypes.create_string_buffer(256)
    _check(load().zk_backend_info(buf, 256), "zk_backend_info")
    return buf.value.decode()


def _ptr(a):
    """numpy array -> host pointer; torch tensor -> data_ptr(); int -> as is."""
    if isinstance(a, int):
        return ctypes.c_void_p(a)
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(ctypes.c_void_p)
    return ctypes.c_void_p(a.data_ptr())


def _np64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def curve_id(c):
    return CURVE_NAMES[c] if isinstance(c, str) else c


def field_id(f):
    return FIELD_NAMES[f] if isinstance(f, str) else f


def base_limbs(curve):
    return load().zk_curve_base_limbs64(curve_id(curve))


def scalar_field(curve):
    return load().zk_curve_scalar_field(curve_id(curve))


def field_modulus(field):
    """the field's prime as a Python int"""
    out = np.zeros(4, dtype=np.uint64)
    _check(load().zk_field_modulus(field_id(field), _ptr(out)), "zk_field_modulus")
    return sum(int(w) << (64 * i) for i, w in enumerate(out.tolist()))


def root_of_unity(field, log_n):
    out = np.zeros(4, dtype=np.uint64)
    _check(load().zk_field_root_of_unity(field_id(field), log_n, _ptr(out)), "zk_field_root_of_unity")
    return out


def multiplicative_generator(field):
    out = np.zeros(4, dtype=np.uint64)
    _check(load().zk_field_multiplicative_generator(field_id(field), _ptr(out)), "zk_field_multiplicative_generator")
    return out


def field_inverse(field, a):
    a = _np64(a)
    out = np.zeros(4, dtype=np.uint64)
    _check(load().zk_field_inverse(field_id(field), _ptr(a), _ptr(out)), "zk_field_inverse")
    return out


def point_add(curve, ja, jb):
    ja, jb = _np64(ja), _np64(jb)
    out = np.zeros_like(ja)
    _check(load().zk_point_add(curve_id(curve), _ptr(ja), _ptr(jb), _ptr(out)), "zk_point_add")
    return out


def point_to_affine(curve, jac):
    jac = _np64(jac)
    out = np.zeros(2 * (jac.shape[-1] // 3), dtype=np.uint64)
    _check(load().zk_point_to_affine(curve_id(curve), _ptr(jac), _ptr(out)), "zk_point_to_affine")
    return out


def msm_window_count(curve, n, window_bits=0):
    return load().zk_msm_window_count(curve_id(curve), n, window_bits)


def msm_window_bits(curve, n, window_bits=0):
    return load().zk_msm_window_bits(curve_id(curve), n, window_bits)


class Bases:
    """Device-resident SRS / proving-key query vector (ark-groth16 0.3 ProvingKey::{a,b_g1,h,l}_query,
    halo2 Params::g), uploaded once and reused for every proof."""

    def __init__(self, curve, points=None, device_tensor=None, n=None):
        self.curve = curve_id(curve)
        h = ctypes.c_uint64(0)
        if device_tensor is not None:
            self._keep = device_tensor
            self.n = int(n if n is not None else device_tensor.shape[0])
            _check(load().zk_bases_adopt_device(self.curve, _ptr(device_tensor), self.n, ctypes.byref(h)),
                   "zk_bases_adopt_device")
        else:
            pts = _np64(points)
            self.n = int(pts.shape[0])
            _check(load().zk_bases_upload(self.curve, _ptr(pts), self.n, ctypes.byref(h)), "zk_bases_upload")
        self.handle = h.value

    def precompute(self, window_bits=0):
        """build the table of window multiples for msm(..., precomputed=True)"""
        _check(load().zk_bases_precompute(self.handle, window_bits), "zk_bases_precompute")

    def refresh(self, offset, count, stream=0):
        """points [offset, offset + count) of the adopted device buffer were rewritten on `stream`: update the derived copies"""
        _check(load().zk_bases_refresh(self.handle, offset, count, ctypes.c_void_p(stream)), "zk_bases_refresh")

    def free(self):
        if self.handle:
            _check(load().zk_bases_free(self.handle), "zk_bases_free")
            self.handle = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def msm_opts(window_bits=0, windows=None, limb_bits=0, split_log=None, slice_len=0, big_threshold=0, waves_per_simd=0,
             no_hot_help=False, base_offset=0, slice_reduce=False, precomputed=False, device_partials=False, own_stream=False, window_group=0):
    o = MsmOpts()
    o.window_bits = window_bits
    if windows is not None:
        o.window_begin, o.window_end = windows
    o.limb_bits = limb_bits
    o.split_log_plus1 = 0 if split_log is None else split_log + 1
    o.slice_len = slice_len
    o.big_threshold = big_threshold
    o.waves_per_simd = waves_per_simd
    o.flags = ((MSM_FLAG_NO_HOT_HELP if no_hot_help else 0) | (MSM_FLAG_SLICE_REDUCE if slice_reduce else 0)
               | (MSM_FLAG_PRECOMPUTED if precomputed else 0) | (MSM_FLAG_DEVICE_PARTIALS if device_partials else 0) | (MSM_FLAG_OWN_STREAM if own_stream else 0))
    o.base_offset = base_offset
    o.window_group = window_group
    return o


def msm(bases, scalars, montgomery=False, window_bits=0, windows=None, stream=0, **tuning):
    """sum_i scalars[i] * bases[i] -> Jacobian (X, Y, Z) as uint64[3 * limbs].

    scalars: numpy uint64 [n, 4] (host) or a torch uint64/int64 tensor on the GPU (device path).
    tuning: the remaining zk_msm_opts fields (limb_bits, split_log, slice_len, big_threshold, waves_per_simd, no_hot_help)."""
    lib = load()
    nl = lib.zk_curve_base_limbs64(bases.curve)
    out = np.zeros(3 * nl, dtype=np.uint64)
    opts = msm_opts(window_bits, windows, **tuning)
    if isinstance(scalars, np.ndarray):
        sc = _np64(scalars)
        n = int(sc.shape[0])
        _check(lib.zk_msm(bases.curve, bases.handle, _ptr(sc), n, int(montgomery), ctypes.byref(opts), _ptr(out)), "zk_msm")
    else:
        n = int(scalars.shape[0])
        _check(lib.zk_msm_device(bases.curve, bases.handle, _ptr(scalars), n, int(montgomery), ctypes.byref(opts),
                                 _ptr(out), ctypes.c_void_p(stream)), "zk_msm_device")
    return out


class MsmTicket:
    """an MSM in flight (zk_msm_submit); .collect() waits for it and returns the Jacobian sum"""

    def __init__(self, ticket, nl, keep):
        self.ticket, self._nl, self._keep = ticket, nl, keep

    def collect(self):
        out = np.zeros(3 * self._nl, dtype=np.uint64)
        _check(load().zk_msm_collect(self.ticket, _ptr(out)), "zk_msm_collect")
        self._keep = None
        return out


def msm_submit(bases, d_scalars, montgomery=False, window_bits=0, windows=None, stream=0, **tuning):
    """enqueue one MSM over device-resident scalars without synchronising; collect the ticket later"""
    lib = load()
    opts = msm_opts(window_bits, windows, **tuning)
    t = ctypes.c_uint64(0)
    _check(lib.zk_msm_submit(bases.curve, bases.handle, _ptr(d_scalars), int(d_scalars.shape[0]), int(montgomery), ctypes.byref(opts),
                             ctypes.c_void_p(stream), ctypes.byref(t)), "zk_msm_submit")
    return MsmTicket(t.value, lib.zk_curve_base_limbs64(bases.curve), d_scalars)


def msm_batch(bases, d_scalars, montgomery=False, window_bits=0, windows=None, stream=0, **tuning):
    """count MSMs over the same bases: d_scalars is a device buffer [count, n, 4]; returns [count, 3 * limbs]"""
    lib = load()
    count, n = int(d_scalars.shape[0]), int(d_scalars.shape[1])
    nl = lib.zk_curve_base_limbs64(bases.curve)
    out = np.zeros((count, 3 * nl), dtype=np.uint64)
    opts = msm_opts(window_bits, windows, **tuning)
    _check(lib.zk_msm_batch_device(bases.curve, bases.handle, _ptr(d_scalars), n, count, n, int(montgomery), ctypes.byref(opts),
                                   _ptr(out), ctypes.c_void_p(stream)), "zk_msm_batch_device")
    return out


def msm_last_profile():
    p = MsmProfile()
    _check(load().zk_msm_last_profile(ctypes.byref(p)), "zk_msm_last_profile")
    return {k: getattr(p, k) for k, _ in MsmProfile._fields_}


def msm_profile_totals(reset=True):
    """sums over every MSM collected since the last reset (batches, MSMs in flight)"""
    t = MsmTotals()
    _check(load().zk_msm_profile_totals(ctypes.byref(t), int(reset)), "zk_msm_profile_totals")
    return {k: getattr(t, k) for k, _ in MsmTotals._fields_}


def ntt_profile_enable(on=True):
    _check(load().zk_ntt_profile_enable(int(on)), "zk_ntt_profile_enable")


def ntt_profile_read():
    """waits for the bracketed ntt_pass_kernel launches since the last read; sums and resets"""
    t = NttTotals()
    _check(load().zk_ntt_profile_read(ctypes.byref(t)), "zk_ntt_profile_read")
    return {k: getattr(t, k) for k, _ in NttTotals._fields_}


def ntt(field, a, omega, scale_by_n_inv=False, stream=0, coset_pre=None, coset_post=None, device=False, in_log=None, src=None):
    """In-place size-2^k DFT with root `omega` (Montgomery); numpy (host, returns a new array) or torch GPU tensor.
    coset_pre / coset_post: a[i] *= g^i before / a[k] *= g^k after the transform, fused into the first / last pass on
    device buffers.  device=True treats a numpy array as a device buffer (only meaningful under the CPU test emulator).
    in_log: the input is a[:2^in_log] zero-extended to len(a) (halo2 coeff_to_extended); the padding is never read.
    src (device buffers): read the input from there instead and leave it untouched -- the result lands in `a` (out of place)."""
    lib = load()
    om = _np64(omega)
    if src is not None:
        n = int(a.shape[0])
        log_n = n.bit_length() - 1
        assert n == 1 << log_n
        li = log_n if in_log is None else int(in_log)
        assert int(src.shape[0]) >= 1 << li
        gp = _np64(coset_pre) if coset_pre is not None else None
        gq = _np64(coset_post) if coset_post is not None else None
        _check(lib.zk_ntt_oop_device(field_id(field), _ptr(src), _ptr(a), log_n, li, _ptr(om), int(scale_by_n_inv),
                                     _ptr(gp) if gp is not None else None, _ptr(gq) if gq is not None else None,
                                     ctypes.c_void_p(stream)), "zk_ntt_oop_device")
        return a
    if isinstance(a, np.ndarray) and not device and in_log is None:
        buf = _np64(a).copy()
        n = buf.shape[0]
        log_n = n.bit_length() - 1
        assert n == 1 << log_n
        if coset_pre is not None:
            _check(lib.zk_coset_mul(field_id(field), _ptr(buf), log_n, _ptr(_np64(coset_pre))), "zk_coset_mul")
        _check(lib.zk_ntt(field_id(field), _ptr(buf), log_n, _ptr(om), int(scale_by_n_inv)), "zk_ntt")
        if coset_post is not None:
            _check(lib.zk_coset_mul(field_id(field), _ptr(buf), log_n, _ptr(_np64(coset_post))), "zk_coset_mul")
        return buf
    n = int(a.shape[0])
    log_n = n.bit_length() - 1
    assert n == 1 << log_n
    if in_log is not None:      # zero-extended input (device buffers): only a[:2^in_log] is read
        gp = _np64(coset_pre) if coset_pre is not None else None
        gq = _np64(coset_post) if coset_post is not None else None
        _check(lib.zk_ntt_extend_device(field_id(field), _ptr(a), log_n, int(in_log), _ptr(om), int(scale_by_n_inv),
                                        _ptr(gp) if gp is not None else None, _ptr(gq) if gq is not None else None,
                                        ctypes.c_void_p(stream)), "zk_ntt_extend_device")
        return a
    if coset_pre is None and coset_post is None:
        _check(lib.zk_ntt_device(field_id(field), _ptr(a), log_n, _ptr(om), int(scale_by_n_inv), ctypes.c_void_p(stream)),
               "zk_ntt_device")
    else:
        gp = _np64(coset_pre) if coset_pre is not None else None
        gq = _np64(coset_post) if coset_post is not None else None
        _check(lib.zk_ntt_coset_device(field_id(field), _ptr(a), log_n, _ptr(om), int(scale_by_n_inv),
                                       _ptr(gp) if gp is not None else None, _ptr(gq) if gq is not None else None,
                                       ctypes.c_void_p(stream)), "zk_ntt_coset_device")
    return a


def coset_mul(field, a, g, stream=0):
    lib = load()
    gm = _np64(g)
    if isinstance(a, np.ndarray):
        buf = _np64(a).copy()
        log_n = buf.shape[0].bit_length() - 1
        _check(lib.zk_coset_mul(field_id(field), _ptr(buf), log_n, _ptr(gm)), "zk_coset_mul")
        return buf
    log_n = int(a.shape[0]).bit_length() - 1
    _check(lib.zk_coset_mul_device(field_id(field), _ptr(a), log_n, _ptr(gm), ctypes.c_void_p(stream)), "zk_coset_mul_device")
    return a


VEC_OPS = {"mul": 0, "sub": 1, "add": 2, "scale": 3, "into_repr": 4, "from_repr": 5, "qap": 6}


def vec_op(field, op, a, b=None, c=None, scalar=None, stream=0):
    """Pointwise kernels on device buffers (torch tensors, or numpy arrays under the test emulator)."""
    n = int(a.shape[0])
    if op == "scale_periodic":      # a[i] *= b[i mod len(b)], b a small host table
        tbl = _np64(b)
        _check(load().zk_vec_scale_periodic_device(field_id(field), _ptr(a), n, _ptr(tbl), int(tbl.shape[0]), ctypes.c_void_p(stream)),
               "zk_vec_scale_periodic_device")
        return a
    sp = _ptr(_np64(scalar)) if scalar is not None else None
    _check(load().zk_vec_op_device(field_id(field), VEC_OPS[op], _ptr(a), _ptr(b) if b is not None else None,
                                   _ptr(c) if c is not None else None, n, sp, ctypes.c_void_p(stream)), "zk_vec_op_device")
    return a


def groth16_witness_map(field, a, b, c, stream=0):
    """ark-groth16 0.3 R1CStoQAP::witness_map from the evaluation vectors a, b, c (device buffers); h lands in `a`."""
    m = int(a.shape[0])
    log_m = m.bit_length() - 1
    assert m == 1 << log_m and int(b.shape[0]) == m and int(c.shape[0]) == m
    _check(load().zk_groth16_witness_map_device(field_id(field), _ptr(a), _ptr(b), _ptr(c), log_m, ctypes.c_void_p(stream)),
           "zk_groth16_witness_map_device")
    return a


def fixed_base_mul_device(curve, d_scalars, d_out, n, stream=0):
    _check(load().zk_fixed_base_mul_device(curve_id(curve), _ptr(d_scalars), n, _ptr(d_out), ctypes.c_void_p(stream)),
           "zk_fixed_base_mul_device")


def fixed_base_msm_device(curve, d_scalars, d_out, n, base=None, montgomery=False, stream=0):
    """d_out[i] = [k_i] base (affine); base: numpy uint64 (x, y) Montgomery limbs on the host, None = the generator."""
    b = _np64(base) if base is not None else None
    _check(load().zk_fixed_base_msm_device(curve_id(curve), _ptr(b) if b is not None else None, _ptr(d_scalars), n, int(montgomery),
                                           _ptr(d_out), ctypes.c_void_p(stream)), "zk_fixed_base_msm_device")


from . import ark, ark_serialize, groth16, halo2  # noqa: E402,F401  (interface mirrors)

PROVER_EXPORTS = groth16.PROVER_EXPORTS + halo2.PROVER_EXPORTS
