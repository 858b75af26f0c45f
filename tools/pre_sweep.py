#!/usr/bin/env python3
"""the one-bucket-set MSM over the table of window multiples (zk_bases_precompute): window bits x bucket split sweep against the
plain form.  usage: pre_sweep.py [curve] [logn]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import contangle_zkcp_amd as zk
from contangle_zkcp_amd import synth
curve = sys.argv[1] if len(sys.argv) > 1 else "Vesta"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << logn
zk.load(); zk.init(0)
st = torch.cuda.current_stream().cuda_stream
ks = synth.scalars_for(curve, n, 1)
d_pts = torch.empty((n, 2 * zk.base_limbs(curve)), dtype=torch.int64, device="cuda")
zk.fixed_base_msm_device(curve, torch.from_numpy(ks.view(np.int64)).cuda(), d_pts, n, stream=st)
torch.cuda.synchronize()
bases = zk.Bases(curve, device_tensor=d_pts, n=n)
d_sc = torch.from_numpy(synth.scalars_for(curve, n, 7).view(np.int64)).cuda()
def run(label, **kw):
    for _ in range(2):
        zk.msm(bases, d_sc, stream=st, **kw)
    zk.msm_profile_totals(reset=True)
    R = 6
    t0 = time.perf_counter()
    for _ in range(R):
        zk.msm(bases, d_sc, stream=st, **kw)
    dt = (time.perf_counter() - t0) / R * 1e3
    t = zk.msm_profile_totals(reset=True)
    p = zk.msm_last_profile()
    print("%s 2^%d %-28s wall %.3f ms | digits %.3f stage %.3f sort %.3f | accumulate kernel %.3f (+ combine %.3f) | reduce %.3f | host tail %.3f | windows %d x %d bits" % (
        curve, logn, label, dt, p["digits_ms"], p["hist_ms"], p["scatter_ms"], t["accumulate_kernel_ms"] / R, (t["accumulate_ms"] - t["accumulate_kernel_ms"]) / R, t["reduce_ms"] / R,
        t["host_tail_ms"] / R, p["windows_done"], p["window_bits"]), flush=True)
run("plain")
for wb in (16, 18, 19, 20):
    bases.precompute(wb)
    torch.cuda.synchronize()
    for sl in (-1, 0, 1, 2):
        run("table c=%d split %s" % (wb, "auto" if sl < 0 else str(sl)), precomputed=True, window_bits=wb, **({} if sl < 0 else {"split_log": sl}))
