"""CPU tier: the single-process multi-device entry of the C ABI (zk_init_devices) under the test emulator,
which reports ZK_EMU_DEVICES "devices": per-device contexts, bases on every device, the window fan-out over one host
thread per device and the host-side addition of the partial sums all run; only the devices are fake.  A subprocess,
because the library's device list is fixed for the life of a process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import importlib.util, os, sys, threading
ROOT = %r
NDEV = %d
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("zk_build", os.path.join(ROOT, "contangle-zkcp_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
lib = b.build_emu()
import contangle_zkcp_amd as zk
import parity_suite as ps
zk.load(path=lib)
zk.init_devices(list(range(NDEV)))
try:
    zk.init_devices([0])
    raise SystemExit("a second, different device list must be refused")
except zk.ZkError:
    pass
ps.check_multi_device(zk, NDEV)
# the ABI is callable from any host thread (rayon workers upstream): same checks from a second thread
err = []
def worker():
    try:
        ps.check_msm_vs_oracle(zk, "Vesta", 200, 5, True)
        ps.check_ntt_vs_oracle(zk, "Bls381Fr", 8)
    except BaseException as e:
        err.append(e)
t = threading.Thread(target=worker); t.start(); t.join()
assert not err, err
zk.shutdown()
print("MULTI-DEVICE-OK")
"""


@pytest.mark.parametrize("ndev", [2, 3])
def test_multi_device_c_entry(ndev):
    env = dict(os.environ)
    env["ZK_EMU_DEVICES"] = str(ndev)
    r = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, ndev)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       env=env, timeout=1200)
    assert r.returncode == 0 and "MULTI-DEVICE-OK" in r.stdout, r.stdout[-3000:]
