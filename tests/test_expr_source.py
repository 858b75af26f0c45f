"""CPU: the source of the quotient kernel specialised per program (zk_expr_specialised_source needs no device) -- its shape for the
bench's program, and that it cross-compiles for gfx950 against the headers that ship next to the library."""
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
sys.path.insert(0, %r)
import contangle_zkcp_amd as zk
from contangle_zkcp_amd import synth
zk.load()                                         # the product library: loading it needs no GPU
prog = synth.quotient_program(13, 8, 3)
src = zk.halo2.expr_specialised_source(%r, prog, 13 + 8 + 6 + 3, 5)
open(sys.argv[1], "w").write(src)
bad = 0
for p in ([("add",)], [("col", 40, 0)], [("col", 0, 0), ("col", 1, 0)]):
    try:
        zk.halo2.expr_specialised_source(%r, p, 30, 5)
    except zk.ZkError:
        bad += 1
print("REFUSED", bad)
"""


@pytest.mark.parametrize("field", ["PallasFp", "Bls381Fr"])
def test_specialised_quotient_kernel_source(field):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "zk_expr_jit.hip")
        r = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, field, field), path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert r.returncode == 0 and "REFUSED 3" in r.stdout, r.stdout[-2000:]
        src = open(path).read()
        assert "using F = %s;" % field in src and 'extern "C" __global__' in src and "zk_expr_jit" in src
        # the bench's program: 56 products + 25 scalings, 122 pushes, 44 additions; every intermediate a named value, no LDS
        assert src.count("fe29_mul(") == 81 and src.count("fe29_add(") == 44 and src.count("fe29_unpack(t") == 122
        assert "__shared__" not in src and "stack" not in src
        r = subprocess.run([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O1", "-std=c++17", "--cuda-device-only", "-include", "hip/hip_runtime.h",
                            "-I" + os.path.join(ROOT, "contangle-zkcp_amd", "csrc"), "-c", path, "-o", os.path.join(d, "k.o")],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
