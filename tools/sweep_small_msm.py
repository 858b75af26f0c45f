#!/usr/bin/env python3
"""Phase times of MSMs of 2^lo .. 2^hi points (the IPA's later rounds, Groth16's small queries) for a few window sizes:
where the launch sequence, not the arithmetic, sets the time.  Prints one line per (log n, c)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import contangle_zkcp_amd as zk
from contangle_zkcp_amd import synth as ps

curve = sys.argv[1] if len(sys.argv) > 1 else "Vesta"
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 19
zk.load(); zk.init(0)
nl = zk.base_limbs(curve)
N = 1 << hi
ks = ps.scalars_for(curve, N, 0x5EED)
d_pts = torch.empty((N, 2 * nl), dtype=torch.int64, device="cuda")
zk.fixed_base_mul_device(curve, torch.from_numpy(ks.view(np.int64)).cuda(), d_pts, N)
torch.cuda.synchronize()
bases = zk.Bases(curve, device_tensor=d_pts, n=N)
d_all = torch.from_numpy(ps.scalars_for(curve, N, 0xC0DE).view(np.int64)).cuda()
for logn in range(lo, hi + 1):
    n = 1 << logn
    d_sc = d_all[:n]
    default_c = zk.load().zk_msm_window_bits(zk.curve_id(curve), n, 0)   # msm_pick_c
    ref = None
    cs = sorted({default_c, max(4, logn - 4), 8, 12, 16})
    for c, slice_reduce in [(c, s) for c in cs for s in (False, True)]:
        kw = {"window_bits": c, "slice_reduce": slice_reduce}
        for _ in range(2):
            out = zk.msm(bases, d_sc, **kw)
        acc = {}
        R = 6
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(R):
            out = zk.msm(bases, d_sc, **kw)
            p = zk.msm_last_profile()
            for k, v in p.items():
                acc[k] = acc.get(k, 0) + v / R
        wall = (time.perf_counter() - t0) / R * 1e3
        aff = zk.point_to_affine(curve, out)
        if ref is None:
            ref = aff
        ok = bool((aff == ref).all())
        print(f"logn={logn} c={c}{'*' if c == default_c else ' '} {'slice' if slice_reduce else 'axes '} {'ok' if ok else 'MISMATCH'} wall={wall:.3f} ",
              {k[:-3]: round(v, 3) for k, v in acc.items() if k.endswith("_ms")}, flush=True)
