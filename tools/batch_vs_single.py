#!/usr/bin/env python3
"""One job of k scalar vectors (zk_msm_batch_device) against k single MSMs, same bases: wall time and accumulate-kernel time per MSM.
At 2^20 points a 4-vector job sorts 64 windows x 2^20 entries (268 MB of sorted entries + 64 MB of bases: past the 256 MiB
Infinity Cache), like a single 2^22 MSM.  usage: batch_vs_single.py [curve] [logn]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import contangle_zkcp_amd as zk
from contangle_zkcp_amd import synth

curve = sys.argv[1] if len(sys.argv) > 1 else "Vesta"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << logn
zk.load()
zk.init(0)
st = torch.cuda.current_stream().cuda_stream
ks = synth.scalars_for(curve, n, 1)
d_pts = torch.empty((n, 2 * zk.base_limbs(curve)), dtype=torch.int64, device="cuda")
zk.fixed_base_msm_device(curve, torch.from_numpy(ks.view(np.int64)).cuda(), d_pts, n, stream=st)
torch.cuda.synchronize()
bases = zk.Bases(curve, device_tensor=d_pts, n=n)
sf = synth.CURVE_SCALAR_FIELD[curve]
cols = torch.from_numpy(np.stack([synth.rand_field(sf, n, 100 + i) for i in range(8)]).view(np.int64)).cuda()


def run(label, fn, msms, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    zk.msm_profile_totals(reset=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t = zk.msm_profile_totals(reset=True)
    print("%-34s wall %.3f ms per MSM | accumulate kernel %.3f | sort %.3f | reduce %.3f | host tail %.3f  (per MSM)" % (
        label, dt * 1e3 / msms, t["accumulate_kernel_ms"] / t["msms"], t["sort_ms"] / t["msms"], t["reduce_ms"] / t["msms"], t["host_tail_ms"] / t["msms"]), flush=True)


run("single x4 (one at a time)", lambda: [zk.msm(bases, cols[i], montgomery=True, stream=st) for i in range(4)], 4)
for k in (1, 2, 3, 4, 8):
    run("batch of %d (jobs of <= 4)" % k, lambda k=k: zk.msm_batch(bases, cols[:k], montgomery=True, stream=st), k)
run("4 tickets in flight", lambda: [t.collect() for t in [zk.msm_submit(bases, cols[i], montgomery=True, stream=st) for i in range(4)]], 4)
