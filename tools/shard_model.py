#!/usr/bin/env python3
"""Per-rank MSM time for the window share a rank of an N-way window-sharded MSM owns (N = 1, 2, 4, 8), measured on one GPU:
the strong-scaling model of contangle-zkcp_amd/dist.py without the (tiny) all_gather.  Usage: shard_model.py [curve] [logn];
SPLITS=0,1,2 sweeps zk_msm_opts.split_log."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import contangle_zkcp_amd as zk
from contangle_zkcp_amd import synth as ps
curve = sys.argv[1] if len(sys.argv) > 1 else "Vesta"
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
zk.load(); zk.init(0)
nl = zk.base_limbs(curve)
ks = ps.scalars_for(curve, n, 0x5EED)
d_pts = torch.empty((n, 2 * nl), dtype=torch.int64, device="cuda")
zk.fixed_base_mul_device(curve, torch.from_numpy(ks.view(np.int64)).cuda(), d_pts, n)
torch.cuda.synchronize()
bases = zk.Bases(curve, device_tensor=d_pts, n=n)
d_sc = torch.from_numpy(ps.scalars_for(curve, n, 0xC0DE).view(np.int64)).cuda()
splits = os.environ.get("SPLITS", "auto").split(",")
WT = zk.msm_window_count(curve, n)
for N in (1, 2, 4, 8):
    W = -(-WT // N)          # the largest share of any rank
    for sp in splits:
        kw = {} if sp == "auto" else {"split_log": int(sp)}
        for _ in range(3): zk.msm(bases, d_sc, windows=(0, W), **kw)
        acc = {}
        R = 10
        t0 = time.perf_counter()
        for _ in range(R):
            zk.msm(bases, d_sc, windows=(0, W), **kw)
            p = zk.msm_last_profile()
            for k, v in p.items(): acc[k] = acc.get(k, 0) + v / R
        wall = (time.perf_counter() - t0) / R * 1e3
        print("ranks", N, "windows/rank", W, "split", sp, "wall %.3f ms" % wall, {k: round(v, 3) for k, v in acc.items() if k.endswith("_ms")}, flush=True)
