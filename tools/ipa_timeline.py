#!/usr/bin/env python3
"""host wall-clock of every step of the fold-free inner-product argument at 2^20 (the bench's opening phase), one line per round"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import contangle_zkcp_amd as zk
from contangle_zkcp_amd import synth
H = zk.halo2
curve, k = "Vesta", 20
n = 1 << k
zk.load(); zk.init(0)
st = torch.cuda.current_stream().cuda_stream
sf = synth.CURVE_SCALAR_FIELD[curve]
ks = synth.scalars_for(curve, n, 1)
d_pts = torch.empty((n, 2 * zk.base_limbs(curve)), dtype=torch.int64, device="cuda")
zk.fixed_base_msm_device(curve, torch.from_numpy(ks.view(np.int64)).cuda(), d_pts, n, stream=st)
bases = zk.Bases(curve, device_tensor=d_pts, n=n)
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
p0, b0 = to_dev(synth.rand_field(sf, n, 5)), to_dev(synth.rand_field(sf, n, 6))
us = [synth.rand_field(sf, 1, 100 + j)[0] for j in range(k)]
d_S, d_W = torch.zeros((2, n, 4), dtype=torch.int64, device="cuda"), torch.empty((n, 4), dtype=torch.int64, device="cuda")
for rep in range(3):
    d_p, d_b = p0.clone(), b0.clone()
    torch.cuda.synchronize()
    ipa = H.IpaProverVirtual(curve, d_p, d_b, bases, lambda shape: torch.zeros(shape, dtype=torch.int64, device="cuda"), stream=st, buffers=(d_S, d_W))
    t0 = time.perf_counter(); line = []
    for j in range(k):
        a = time.perf_counter()
        ipa.round()
        b = time.perf_counter()
        if rep == 2 and j in (0, 10):
            prof = zk.msm_last_profile()
            print("round %d batch-MSM profile:" % j, {kk: round(v, 3) if isinstance(v, float) else v for kk, v in prof.items()})
        ipa.fold(us[j])
        c = time.perf_counter()
        extra = 0.0
        if j + 1 == 6:
            ipa.collapse()
            extra = time.perf_counter() - c
        line.append("%d:%.2f+%.2f%s" % (j, (b - a) * 1e3, (c - b) * 1e3, ("+collapse %.2f" % (extra * 1e3)) if extra else ""))
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) * 1e3
    ipa.free()
    if rep == 2:
        print("rounds (round ms + fold enqueue ms):", " ".join(line))
        print("total %.2f ms" % tot)
