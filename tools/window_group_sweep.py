#!/usr/bin/env python3
"""A/B of zk_msm_opts.window_group on one large MSM: usage window_group_sweep.py [curve] [logn]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import contangle_zkcp_amd as zk
from contangle_zkcp_amd import synth
curve = sys.argv[1] if len(sys.argv) > 1 else "Bn254G1"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 22
n = 1 << logn
zk.load(); zk.init(0)
st = torch.cuda.current_stream().cuda_stream
ks = synth.scalars_for(curve, n, 1)
d_pts = torch.empty((n, 2 * zk.base_limbs(curve)), dtype=torch.int64, device="cuda")
zk.fixed_base_msm_device(curve, torch.from_numpy(ks.view(np.int64)).cuda(), d_pts, n, stream=st)
torch.cuda.synchronize()
bases = zk.Bases(curve, device_tensor=d_pts, n=n)
for realistic in (False, True):
    d_sc = torch.from_numpy(synth.scalars_for(curve, n, 7, realistic=realistic).view(np.int64)).cuda()
    W = zk.msm_window_count(curve, n)
    for gw in (W, 8, 4, 2, 0):
        for _ in range(2):
            zk.msm(bases, d_sc, window_group=gw, stream=st)
        zk.msm_profile_totals(reset=True)
        t0 = time.perf_counter()
        R = 5
        for _ in range(R):
            zk.msm(bases, d_sc, window_group=gw, stream=st)
        dt = (time.perf_counter() - t0) / R * 1e3
        t = zk.msm_profile_totals(reset=True)
        p = zk.msm_last_profile()
        print("%s 2^%d %s window_group %2d (groups %d): wall %.3f ms | accumulate kernel %.3f | device %.3f | host tail %.3f" % (
            curve, logn, "0/1-heavy" if realistic else "uniform", gw, p["groups"], dt, t["accumulate_kernel_ms"] / R, t["device_ms"] / R, t["host_tail_ms"] / R), flush=True)
