#!/usr/bin/env python3
"""Sweep NTT plan knobs (zk_ntt_configure: max_log_radix, log_tile, block) on one GPU."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import contangle_zkcp_amd as zk
from contangle_zkcp_amd import synth as ps
name = sys.argv[1] if len(sys.argv) > 1 else "PallasFp"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
zk.load(); zk.init(0)
a = ps.rand_field(name, 1 << logn, 7)
w = zk.root_of_unity(name, logn)
ref = None
st = torch.cuda.current_stream().cuda_stream
for maxr, logt, blk in [(10, 2, 256), (10, 2, 512), (10, 2, 1024), (10, 1, 256), (10, 1, 512), (10, 0, 256), (7, 3, 256), (7, 4, 256), (7, 4, 512), (7, 3, 512), (8, 3, 512), (9, 2, 512)]:
    zk.ntt_configure(max_log_radix=maxr, log_tile=logt, block=blk)
    d = torch.from_numpy(a.view(np.int64)).cuda()
    zk.ntt(name, d, w, stream=st); torch.cuda.synchronize()
    out = d.cpu().numpy().view(np.uint64)
    if ref is None: ref = out
    ok = bool((out == ref).all())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): zk.ntt(name, d, w, stream=st)
    e1.record(); torch.cuda.synchronize()
    print("max_logr", maxr, "logt", logt, "block", blk, "ok" if ok else "MISMATCH", "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3), flush=True)
