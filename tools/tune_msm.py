#!/usr/bin/env python3
"""Sweep the MSM launch knobs (zk_msm_opts: window_bits, slice_len, waves_per_simd, split_log, big_threshold) on one GPU;
prints phase times."""
import itertools, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import contangle_zkcp_amd as zk
from contangle_zkcp_amd import synth as ps

curve = sys.argv[1] if len(sys.argv) > 1 else "Vesta"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << logn
zk.load(); zk.init(0)
nl = zk.base_limbs(curve)
ks = ps.scalars_for(curve, n, 0x5EED)
d_pts = torch.empty((n, 2 * nl), dtype=torch.int64, device="cuda")
zk.fixed_base_mul_device(curve, torch.from_numpy(ks.view(np.int64)).cuda(), d_pts, n)
torch.cuda.synchronize()
bases = zk.Bases(curve, device_tensor=d_pts, n=n)
d_sc = torch.from_numpy(ps.scalars_for(curve, n, 0xC0DE).view(np.int64)).cuda()
ref = None
grid = {"split_log": [0, 1], "window_bits": [16, 15], "slice_len": [1, 2, 4, 16], "waves_per_simd": [2, 3]}
if os.environ.get("TUNE_GRID"):   # e.g. TUNE_GRID='{"slice_len": [2, 4]}'
    import json
    grid = json.loads(os.environ["TUNE_GRID"])
base = {"window_bits": 16, "slice_len": 8, "waves_per_simd": 3, "split_log": 1}
configs = [dict(base)]
for k, vals in grid.items():
    for v in vals:
        c = dict(base); c[k] = v
        if c not in configs: configs.append(c)
for c in configs:
    for _ in range(2): out = zk.msm(bases, d_sc, **c)
    acc = {}
    R = 5
    for _ in range(R):
        out = zk.msm(bases, d_sc, **c)
        p = zk.msm_last_profile()
        for k, v in p.items(): acc[k] = acc.get(k, 0) + v / R
    aff = zk.point_to_affine(curve, out)
    if ref is None: ref = aff
    ok = bool((aff == ref).all())
    print(c, "ok" if ok else "MISMATCH", {k: round(v, 3) for k, v in acc.items() if k.endswith("_ms")}, flush=True)
