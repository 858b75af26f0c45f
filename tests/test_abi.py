"""CPU: the C-ABI library loads and exports every symbol include/zkcp_amd.h declares; without a
GPU the compute entry points fail loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="zkcp_amd.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    import contangle_zkcp_amd as zk
    if not os.path.exists(zk.LIB_PATH):
        import importlib.util
        spec = importlib.util.spec_from_file_location("zk_build", os.path.join(ROOT, "contangle-zkcp_amd", "build.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        b.build_hip()
    import ctypes
    lib = ctypes.CDLL(zk.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "missing export " + s
    assert sorted(zk.EXPORTS) == syms
    # the prover-side header (wire formats, polynomial helpers, Groth16 glue)
    psyms = declared_symbols("zkcp_amd_prover.h")
    for s in psyms:
        assert hasattr(lib, s), "missing export " + s
    assert sorted(zk.ark_serialize.PROVER_EXPORTS + getattr(zk, "PROVER_EXPORTS", [])) == psyms


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import ctypes
    import contangle_zkcp_amd as zk
    lib = ctypes.CDLL(zk.LIB_PATH)
    lib.zk_strerror.restype = ctypes.c_char_p
    st = lib.zk_init(0)
    assert st == -3 and b"no CPU fallback" in lib.zk_strerror(st)
    out = np.zeros(12, dtype=np.uint64)
    # every compute entry point refuses to run uninitialised
    assert lib.zk_ntt(0, out.ctypes.data_as(ctypes.c_void_p), 1, out.ctypes.data_as(ctypes.c_void_p), 0) == -2
    assert lib.zk_msm(0, ctypes.c_uint64(1), None, ctypes.c_uint64(0), 0, None, out.ctypes.data_as(ctypes.c_void_p)) == -2


def test_host_helpers_need_no_gpu():
    """root of unity / generator / inverse / point add are host-side and match the oracle."""
    import contangle_zkcp_amd as zk
    from oracle import zk_oracle as orc
    zk._lib = None
    zk.load()
    for name in ("PallasFp", "PallasFq", "Bn254Fr", "Bls381Fr"):
        for k in (0, 1, 7, 20, orc.lib().orc_field_two_adicity(orc.fid(name))):
            assert (zk.root_of_unity(name, k) == orc.root_of_unity(name, k)).all()
        g = zk.multiplicative_generator(name)
        assert (g == orc.field_generator(name)).all()
        assert (zk.field_inverse(name, g) == orc.fe_op(name, "inv", g)).all()
    for cname in ("Pallas", "Vesta", "Bn254G1", "Bls381G1", "Bn254G2", "Bls381G2"):
        g = orc.curve_generator(cname)
        nl = g.shape[0] // 2
        bl = orc.field_nlimbs(orc.curve_base_field(cname))
        one = np.zeros(nl, dtype=np.uint64)
        one[:bl] = orc.to_mont(orc.curve_base_field(cname), orc.ints_to_array([1], bl))[0]   # (1, 0) on G2
        jac = np.concatenate([g, one])
        assert (zk.point_to_affine(cname, jac) == g).all()
        two = zk.point_add(cname, jac, jac)                      # doubling branch
        assert (zk.point_to_affine(cname, two) == orc.point_add(cname, g, g)).all()
        three = zk.point_add(cname, two, jac)
        assert (zk.point_to_affine(cname, three) == orc.scalar_mul(cname, g, orc.int_to_limbs(3, 4))).all()
        ident = np.zeros(3 * nl, dtype=np.uint64)
        assert (zk.point_to_affine(cname, zk.point_add(cname, ident, jac)) == g).all()
        assert zk.msm_window_count(cname, 1 << 20, 16) == 16
    zk._lib = None
