"""CPU: UndefinedBehaviorSanitizer over the kernel sources (GPU sanitizers are not available on the pool; the
emulator build under tests/emu is how the kernels' indexing / shifts / integer arithmetic get sanitized).
-fno-sanitize-recover: any UB report aborts the process, so a pass means none was hit."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import importlib.util, os, sys
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("zk_build", os.path.join(ROOT, "contangle-zkcp_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
lib = b.build_emu(sanitize=True)
import contangle_zkcp_amd as zk
import parity_suite as ps
zk.load(path=lib); zk.init(0)
zk.ntt_configure(max_log_radix=3, log_tile=1)
ps.check_ntt_vs_oracle(zk, "Bls381Fr", 7)
zk.ntt_configure()
ps.check_ntt_vs_oracle(zk, "PallasFp", 11)
ps.check_msm_vs_oracle(zk, "Vesta", 300, 6, True)
ps.check_msm_vs_oracle(zk, "Bls381G2", 40, 4, False)
ps.check_msm_edges(zk, "Bn254G1")
ps.check_msm_big_buckets(zk, "Pallas", n=2300, window_bits=5)
ps.check_witness_map(zk, "Bn254Fr", 5)
zk.shutdown()
print("UBSAN-OK")
"""


def test_kernels_under_ubsan():
    env = dict(os.environ)
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       env=env, timeout=1500)
    assert r.returncode == 0 and "UBSAN-OK" in r.stdout, r.stdout[-3000:]
    assert "runtime error" not in r.stdout
