import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import contangle_zkcp_amd as zk
from contangle_zkcp_amd import synth as ps
zk.load(); zk.init(0)
curve="Vesta"; nl=zk.base_limbs(curve)
for logn in (13,14,15):
    N=1<<logn
    ks=ps.scalars_for(curve,N,0x5EED)
    d_pts=torch.empty((N,2*nl),dtype=torch.int64,device="cuda")
    zk.fixed_base_mul_device(curve, torch.from_numpy(ks.view(np.int64)).cuda(), d_pts, N); torch.cuda.synchronize()
    bases=zk.Bases(curve, device_tensor=d_pts, n=N)
    sc=ps.scalars_for(curve,N,0xC0DE); sc[N//2:]=0      # half zero, as an IPA round's S_L
    d=torch.from_numpy(sc.view(np.int64)).cuda()
    S=torch.stack([d,d.flip(0).contiguous()]).contiguous()
    for c in (8,9,10,11,12,13,14,16):
        for _ in range(3): zk.msm_batch(bases,S,window_bits=c)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(10): zk.msm_batch(bases,S,window_bits=c)
        print(logn,c,round((time.perf_counter()-t0)/10*1e3,3),flush=True)
