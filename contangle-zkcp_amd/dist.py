"""Multi-GPU MSM: one process per GPU, window-range sharding, one tiny collective.

Rank g owns windows [W*g/G, W*(g+1)/G) of the signed-digit decomposition; every
rank holds all bases (the SRS is uploaded once per GPU) and all scalars.  Each rank returns
sum_{w in range} 2^(c*w) T_w as one Jacobian point; EC addition is not an RCCL reduction op, so
the "reduce" is an all_gather of 3*limbs u64 words per rank (96 B; 144 B for BLS12-381 G1)
followed by <= G-1 point additions on every rank.  NTT stays single-GPU (north_star).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import msm, msm_window_count, point_add


def window_range(n_windows, rank, world):
    return n_windows * rank // world, n_windows * (rank + 1) // world


def msm_sharded(bases, scalars, montgomery=False, window_bits=0, group=None, stream=0):
    """All ranks call with the same bases/scalars; returns the full Jacobian sum on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = int(scalars.shape[0])
    nwin = msm_window_count(bases.curve, n, window_bits)
    lo, hi = window_range(nwin, rank, world)
    part = msm(bases, scalars, montgomery=montgomery, window_bits=window_bits, windows=(lo, hi), stream=stream) \
        if hi > lo else _identity(part_len=None, bases=bases)
    if world == 1:
        return part
    backend = dist.get_backend(group)
    t = torch.from_numpy(part.view(np.int64).copy())
    if backend == "nccl":
        t = t.cuda()
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    acc = None
    for o in outs:
        p = o.cpu().numpy().view(np.uint64)
        acc = p if acc is None else point_add(bases.curve, acc, p)
    return acc


def _identity(part_len, bases):
    from . import base_limbs
    return np.zeros(3 * base_limbs(bases.curve), dtype=np.uint64)
