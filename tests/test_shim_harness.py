"""The marshalling contract of the Rust shims, executed without Rust: tests/shim_harness.c makes the calls of
rust/zkcp-amd-sys/src/lib.rs (init_once, SrsCache::get_or_upload with its content probe, jacobian_to_ark_uncompressed) and of the
ark-ec fork's VariableBaseMSM::multi_scalar_mul in the same order through the C ABI, on ark-serialize bytes written by the
pure-Python codec (oracle/pyref_ark.py).  CPU tier: against the emulator build; -m gpu: against libzkcp_amd.so."""
import os
import struct
import subprocess

import numpy as np
import pytest

import parity_suite as ps
from oracle import pyref, pyref_ark
from oracle import zk_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "shim_harness.c")
BIN = os.path.join(ROOT, "tests", "emu", "shim_harness")
CURVE_ID = {"Bn254G1": 2, "Bls381G1": 3, "Bn254G2": 4, "Bls381G2": 5}


def build_harness():
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O1", "-Wall", "-Wextra", "-std=gnu11", "-o", BIN, SRC, "-ldl"])
    return BIN


def to_tuples(cname, aff):
    """oracle affine rows (Montgomery limbs) -> pyref points (ints / int pairs, None = infinity)"""
    bf = pyref.CURVES[cname][0]
    L = orc.coord_limbs(cname)
    nl = pyref.FIELDS[bf][2]
    out = []
    for row in np.atleast_2d(aff):
        if not row.any():
            out.append(None)
            continue
        vals = [orc.limbs_to_int(orc.from_mont(bf, row[j * nl:(j + 1) * nl].reshape(1, nl))[0]) for j in range(2 * L // nl)]
        half = len(vals) // 2
        out.append((vals[0], vals[1]) if half == 1 else ((vals[0], vals[1]), (vals[2], vals[3])))
    return out


def run_case(lib, cname, tmp_path):
    n = 150
    pts_a, pts_b = ps.bases_for(cname, n, seed=61), ps.bases_for(cname, n, seed=62)
    pts_a = pts_a.copy()
    pts_a[3] = 0                                                         # an identity element: (0, 1) + flag on the wire
    sc = [ps.scalars_for(cname, n, 70 + i, realistic=(i == 1)) for i in range(4)]
    enc = lambda pts: b"".join(pyref_ark.encode_point(cname, P, False) for P in to_tuples(cname, pts))
    ea, eb = enc(pts_a), enc(pts_b)
    ps_bytes = pyref_ark.point_size(cname, False)
    assert len(ea) == n * ps_bytes
    # vector 0: key A ; vector 1: key A again at the same address (cache hit: no upload) ; vector 2: key B at A's address (probe
    # mismatch: free + upload) ; vector 3: key B somewhere else (second upload of the same contents at a new address)
    plan = [(ea, 0, sc[0], pts_a), (ea, 1, sc[1], pts_a), (eb, 1, sc[2], pts_b), (eb, 0, sc[3], pts_b)]
    blob = struct.pack("<II", CURVE_ID[cname], len(plan))
    for e, reuse, s, _ in plan:
        blob += struct.pack("<II", n, reuse) + e + np.ascontiguousarray(s, dtype=np.uint64).tobytes()
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(blob)
    r = subprocess.run([build_harness(), lib, str(fin), str(fout)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = fout.read_bytes()
    rec = ps_bytes + 8
    assert len(out) == rec * len(plan)
    stats = []
    for i, (_, _, s, pts) in enumerate(plan):
        got = out[i * rec:i * rec + ps_bytes]
        exp = orc.msm_ark(cname, pts, s, threads=4)
        assert got == pyref_ark.encode_point(cname, to_tuples(cname, exp)[0], False), (cname, i)
        stats.append(struct.unpack("<II", out[i * rec + ps_bytes:(i + 1) * rec]))
    assert stats == [(1, 0), (1, 0), (2, 1), (3, 1)], stats          # (uploads, frees) after each call


@pytest.mark.parametrize("cname", ["Bls381G1", "Bn254G2"])
def test_shim_contract_on_the_emulator(cname, tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("zk_build", os.path.join(ROOT, "contangle-zkcp_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    run_case(b.build_emu(), cname, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("cname", ["Bls381G1", "Bls381G2", "Bn254G1"])
def test_shim_contract_on_the_gpu(cname, tmp_path):
    lib = os.path.join(ROOT, "contangle-zkcp_amd", "libzkcp_amd.so")
    assert os.path.exists(lib)
    run_case(lib, cname, tmp_path)
