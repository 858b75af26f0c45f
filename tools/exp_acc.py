import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import contangle_zkcp_amd as zk
import parity_suite as ps
curve="Vesta"
zk.load(); zk.init(0)
for logn in (20, 19, 18):
    n = 1 << logn
    ks = ps.scalars_for(curve, n, 0x5EED)
    d_pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    zk.fixed_base_mul_device(curve, torch.from_numpy(ks.view(np.int64)).cuda(), d_pts, n)
    torch.cuda.synchronize()
    bases = zk.Bases(curve, device_tensor=d_pts, n=n)
    d_sc = torch.from_numpy(ps.scalars_for(curve, n, 0xC0DE).view(np.int64)).cuda()
    for env in ({}, {"ZK_MSM_DEBUG_MASK": "0x3ff"}, {"ZK_MSM_DEBUG_MASK": "0xffff"}, {"ZK_MSM_WAVES": "8"}, {"ZK_MSM_WAVES":"8","ZK_MSM_DEBUG_MASK": "0x3ff"}):
        for k in ("ZK_MSM_DEBUG_MASK", "ZK_MSM_WAVES"): os.environ.pop(k, None)
        os.environ["ZK_MSM_C"] = "16"
        os.environ.update(env)
        for _ in range(2): zk.msm(bases, d_sc)
        acc = 0
        for _ in range(5):
            zk.msm(bases, d_sc); acc += zk.msm_last_profile()["accumulate_ms"] / 5
        print(logn, env, "accumulate_ms %.3f" % acc, "adds/s %.2f G" % (n * 16 / acc / 1e6), flush=True)
    bases.free()
