import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd())
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import contangle_zkcp_amd as zk
import parity_suite as ps
zk.load(); zk.init(0)
for cname in ("Bn254G1", "Bls381G1", "Vesta"):
    try:
        ps.check_ipa(zk, cname, 3)
        print(cname, "ipa ok", flush=True)
    except AssertionError as e:
        print(cname, "ipa FAIL", str(e)[:200], flush=True)
    except Exception as e:
        print(cname, "ipa ERR", repr(e)[:300], flush=True)
