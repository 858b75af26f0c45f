#!/usr/bin/env python3
"""Per-rank MSM time for the window share a rank of an N-way window-sharded MSM owns (N = 1, 2, 4, 8), measured on one GPU:
the strong-scaling model of contangle-zkcp_amd/dist.py without the (tiny) all_gather.  Optional: ZK_MSM_SPLIT sweep."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import contangle_zkcp_amd as zk
import parity_suite as ps
curve = sys.argv[1] if len(sys.argv) > 1 else "Vesta"
n = 1 << 20
zk.load(); zk.init(0)
nl = zk.base_limbs(curve)
ks = ps.scalars_for(curve, n, 0x5EED)
d_pts = torch.empty((n, 2 * nl), dtype=torch.int64, device="cuda")
zk.fixed_base_mul_device(curve, torch.from_numpy(ks.view(np.int64)).cuda(), d_pts, n)
torch.cuda.synchronize()
bases = zk.Bases(curve, device_tensor=d_pts, n=n)
d_sc = torch.from_numpy(ps.scalars_for(curve, n, 0xC0DE).view(np.int64)).cuda()
splits = os.environ.get("SPLITS", "auto").split(",")
for N in (1, 2, 4, 8):
    W = 16 // N
    for sp in splits:
        if sp == "auto": os.environ.pop("ZK_MSM_SPLIT", None)
        else: os.environ["ZK_MSM_SPLIT"] = sp
        for _ in range(3): zk.msm(bases, d_sc, windows=(0, W))
        acc = {}
        R = 10
        t0 = time.perf_counter()
        for _ in range(R):
            zk.msm(bases, d_sc, windows=(0, W))
            p = zk.msm_last_profile()
            for k, v in p.items(): acc[k] = acc.get(k, 0) + v / R
        wall = (time.perf_counter() - t0) / R * 1e3
        print("ranks", N, "windows/rank", W, "split", sp, "wall %.3f ms" % wall, {k: round(v, 3) for k, v in acc.items() if k.endswith("_ms")}, flush=True)
