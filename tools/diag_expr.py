import os, sys, subprocess
code = r'''
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import torch; assert torch.cuda.is_available()
import contangle_zkcp_amd as zk
zk.load(); zk.init(0)
import parity_suite as ps
try:
    ps.check_expression(zk, "PallasFp", 8, ext=3)
    print("PASS")
except AssertionError as e:
    print("FAIL", e)
'''
for slots in ("1", "2", "4"):
    for hoist in ("0", "12"):
        env = dict(os.environ, ZK_EXPR29_SLOTS=slots, ZK_EXPR29_HOIST=hoist)
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print("slots", slots, "hoist", hoist, "->", r.stdout.strip().splitlines()[-1][:200], flush=True)
