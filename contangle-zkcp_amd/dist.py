"""Multi-GPU MSM, one process per GPU: window-range sharding and one tiny collective.

Rank g owns windows [W*g/G, W*(g+1)/G) of the signed-digit decomposition; every
rank holds all bases (the SRS is uploaded once per GPU) and all scalars.  Each rank returns
sum_{w in range} 2^(c*w) T_w as one Jacobian point; EC addition is not an RCCL reduction op, so
the "reduce" is an all_gather of 3*limbs u64 words per rank and MSM (96 B; 144 B for BLS12-381 G1)
followed by <= G-1 point additions on every rank.  MSMs that are issued together (the column
commitments of a halo2 proof, the five MSMs of a Groth16 proof) share ONE all_gather.
NTT stays single-GPU (north_star).

(The other multi-GPU form -- one process driving several GPUs through zk_init_devices, the
fan-out and the additions inside the C ABI -- needs none of this: see include/zkcp_amd.h.)
"""
import numpy as np
import torch
import torch.distributed as dist

from . import base_limbs, msm, msm_batch, msm_submit, msm_window_count, point_add


# Below this many points an MSM is bound by its dependent launches, not by its additions (2^14 points
# 0.51 ms, 2^16 0.62 ms against 1.9 ms at 2^20): a window share is no faster than the whole, and the exchange would only add
# its latency -- every rank then computes the whole sum itself (identical on all ranks, no collective).
SHARD_MIN_POINTS = 1 << 17


def window_range(n_windows, rank, world):
    return n_windows * rank // world, n_windows * (rank + 1) // world


def _world(group):
    if not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _identity(curve):
    return np.zeros(3 * base_limbs(curve), dtype=np.uint64)


def _gather_add(curves, parts, group):
    """parts: this rank's Jacobian partial sums, one per MSM (curves[i] names the curve of parts[i]); ONE collective for the
    concatenation -- over RCCL: one H2D of the (few hundred bytes of) partials, all_gather_into_tensor on the device, one D2H of
    the gathered block -- then every rank adds the G partials of every MSM (zk_point_add: EC addition is not a reduction op)"""
    world, _ = _world(group)
    if world == 1:
        return parts
    flat = np.concatenate([np.ascontiguousarray(p, dtype=np.uint64).ravel() for p in parts])
    t = torch.from_numpy(flat.view(np.int64).copy())
    if dist.get_backend(group) == "nccl":
        dev = t.cuda()
        out = torch.empty((world, dev.numel()), dtype=dev.dtype, device=dev.device)
        dist.all_gather_into_tensor(out.view(-1), dev, group=group)
        host = out.cpu().numpy().view(np.uint64)          # one copy of the whole block (synchronises)
    else:
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        host = np.stack([o.numpy().view(np.uint64) for o in outs])
    res, off = [], 0
    for c, p in zip(curves, parts):
        ln = p.size
        acc = None
        for r in range(world):
            q = host[r, off:off + ln]
            acc = q.copy() if acc is None else point_add(c, acc, q)
        res.append(acc)
        off += ln
    return res


_gather_tmp = {}


def gather_parts(local, out, stream=None, group=None):
    """The sharded quotient's one exchange: rank j holds h on its sub-coset (local[i] = the extended evaluation i * world + j);
    afterwards every rank holds the whole extended vector, out[i * world + j] = rank j's local[i].  ONE all_gather of
    32 bytes per extended row over RCCL (device to device), then a strided copy; `local` and `out` are device buffers
    (numpy arrays under the CPU test emulator)."""
    world, rank = _world(group)
    m = int(local.shape[0])
    if world == 1:
        if out is not local:
            out[:] = local
        return out
    assert int(out.shape[0]) == m * world
    if isinstance(local, np.ndarray):                                  # CPU tests: "device" memory is host memory
        t = torch.from_numpy(np.ascontiguousarray(local).view(np.int64))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        stacked = np.stack([o.numpy().view(np.uint64).reshape(m, -1) for o in outs])
        out[:] = stacked.transpose(1, 0, 2).reshape(m * world, -1)
        return out
    width = int(local.shape[1])
    if dist.get_backend(group) == "nccl":
        key = (world, m, width, local.device)
        if key not in _gather_tmp:
            _gather_tmp.clear()
            _gather_tmp[key] = torch.empty((world, m, width), dtype=local.dtype, device=local.device)
        tmp = _gather_tmp[key]
        dist.all_gather_into_tensor(tmp.view(-1), local.reshape(-1), group=group)     # ordered with the current stream on both sides
    else:                                                              # gloo rehearsal on a GPU box: staged through the host
        t = local.cpu()
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        tmp = torch.stack(outs).to(local.device)
    out.view(m, world, width).copy_(tmp.permute(1, 0, 2))
    return out


def gather_stack(local, out, group=None):
    """device buffers: out[r] = rank r's `local` (out: [world, ...local.shape]); `local` may be out[rank] itself (in place).
    One all_gather over RCCL; gloo (CPU rehearsal) goes through the host."""
    world, rank = _world(group)
    if isinstance(local, np.ndarray):                                  # CPU tests: "device" memory is host memory
        if world == 1:
            out[0] = local
            return out
        t = torch.from_numpy(np.ascontiguousarray(local).view(np.int64))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        for r_, o in enumerate(outs):
            out[r_] = o.numpy().view(np.uint64).reshape(local.shape)
        return out
    if world == 1:
        if out[0].data_ptr() != local.data_ptr():
            out[0].copy_(local)
        return out
    if dist.get_backend(group) == "nccl":
        # (a private copy of the share: `local` may be a slice of `out`, and an aliased send / receive pair is not something to rely on)
        dist.all_gather_into_tensor(out.view(-1), local.reshape(-1).clone(), group=group)
    else:
        t = local.cpu()
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        out.copy_(torch.stack(outs).to(out.device))
    return out


def gather_rows(local, max_rows, group=None):
    """host rows (numpy [rows <= max_rows, w] u64: e.g. a rank's commitments) from every rank, padded to max_rows each:
    numpy [world * max_rows, w], rank r's rows at r * max_rows.  One all_gather."""
    world, rank = _world(group)
    local = np.ascontiguousarray(local, dtype=np.uint64)
    w = int(local.shape[1])
    pad = np.zeros((max_rows, w), dtype=np.uint64)
    pad[:local.shape[0]] = local
    if world == 1:
        return pad
    t = torch.from_numpy(pad.view(np.int64))
    on_gpu = dist.get_backend(group) == "nccl"
    if on_gpu:
        t = t.cuda()
    out = torch.empty((world * max_rows, w), dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out.cpu().numpy().view(np.uint64)


def msm_sharded(bases, scalars, montgomery=False, window_bits=0, group=None, stream=0):
    """All ranks call with the same bases/scalars; returns the full Jacobian sum on every rank."""
    world, rank = _world(group)
    n = int(scalars.shape[0])
    lo, hi = window_range(msm_window_count(bases.curve, n, window_bits), rank, world)
    if world == 1 or n < SHARD_MIN_POINTS:
        return msm(bases, scalars, montgomery=montgomery, window_bits=window_bits, stream=stream)
    part = msm(bases, scalars, montgomery=montgomery, window_bits=window_bits, windows=(lo, hi), stream=stream) \
        if hi > lo else _identity(bases.curve)
    return _gather_add([bases.curve], [part], group)[0]


def _gather_blocks(blocks, counts, width, group):
    """rank r contributes blocks [counts[r], width] (its whole results); every rank gets all of them in rank order -- one collective"""
    world, rank = _world(group)
    cap = max(counts)
    mine = np.zeros((cap, width), dtype=np.uint64)
    mine[:counts[rank]] = blocks
    t = torch.from_numpy(mine.view(np.int64).copy())
    if dist.get_backend(group) == "nccl":
        dev = t.cuda()
        out = torch.empty((world,) + tuple(dev.shape), dtype=dev.dtype, device=dev.device)
        dist.all_gather_into_tensor(out.view(-1), dev.view(-1), group=group)
        host = out.cpu().numpy().view(np.uint64)
    else:
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        host = np.stack([o.numpy().view(np.uint64) for o in outs])
    return np.concatenate([host[r, :counts[r]] for r in range(world)])


def msm_batch_sharded(bases, d_cols, montgomery=False, window_bits=0, group=None, stream=0):
    """count MSMs over the same bases (d_cols: device buffer [count, n, 4]); returns [count, 3 * limbs] on every rank.
    With at least one vector per rank every rank computes the WHOLE MSMs of its share of the vectors -- the fixed costs of an
    MSM (digit extraction over all 16 windows, sort launches, the latency-bound reduction, the host tail) are then paid once
    per vector, not once per vector and rank, and nothing has to be added afterwards -- and one all_gather hands the results
    round.  Fewer vectors than ranks (the three permutation products on 4 or 8 GPUs): each vector by scalar-window range
    (SURVEY 8e), one batched device call per rank and ONE all_gather of the partial sums for all of them."""
    world, rank = _world(group)
    count, n = int(d_cols.shape[0]), int(d_cols.shape[1])
    if world == 1 or n < SHARD_MIN_POINTS:
        return msm_batch(bases, d_cols, montgomery=montgomery, window_bits=window_bits, stream=stream)
    if count >= world:
        bounds = [count * r // world for r in range(world + 1)]
        lo, hi = bounds[rank], bounds[rank + 1]
        mine = msm_batch(bases, d_cols[lo:hi], montgomery=montgomery, window_bits=window_bits, stream=stream)
        return _gather_blocks(np.asarray(mine), [bounds[r + 1] - bounds[r] for r in range(world)], 3 * base_limbs(bases.curve), group)
    lo, hi = window_range(msm_window_count(bases.curve, n, window_bits), rank, world)
    if hi > lo:
        parts = msm_batch(bases, d_cols, montgomery=montgomery, window_bits=window_bits, windows=(lo, hi), stream=stream)
    else:
        parts = np.stack([_identity(bases.curve)] * count)
    return np.stack(_gather_add([bases.curve] * count, list(parts), group))


def msm_many_sharded(jobs, window_bits=0, group=None, stream=0):
    """jobs: [(bases, device scalars, montgomery[, base_offset])] with different bases (the five MSMs of a Groth16 proof): every MSM is
    submitted before the first is collected (deferred results), then one all_gather combines the ranks' partials"""
    world, rank = _world(group)
    tickets, curves = [], []
    for job in jobs:
        bases, sc, mont = job[:3]
        base_offset = job[3] if len(job) > 3 else 0
        n = int(sc.shape[0])
        lo, hi = window_range(msm_window_count(bases.curve, n, window_bits), rank, world)
        if n < SHARD_MIN_POINTS:            # whole on every rank (see SHARD_MIN_POINTS); mixed with shared MSMs below
            lo, hi = (0, msm_window_count(bases.curve, n, window_bits)) if rank == 0 else (0, 0)
        curves.append(bases.curve)
        if world > 1 and hi == lo:
            tickets.append(None)
            continue
        if len([t for t in tickets if t is not None and not isinstance(t, np.ndarray)]) >= 3:   # <= 4 MSMs in flight per device
            for i, t in enumerate(tickets):
                if t is not None and not isinstance(t, np.ndarray):
                    tickets[i] = t.collect()
                    break
        tickets.append(msm_submit(bases, sc, montgomery=mont, window_bits=window_bits, windows=(lo, hi) if world > 1 else None,
                                  stream=stream, base_offset=base_offset, own_stream=True))     # (the scalars stay untouched until the collects below)
    parts = []
    for c, t in zip(curves, tickets):
        parts.append(_identity(c) if t is None else (t if isinstance(t, np.ndarray) else t.collect()))
    return _gather_add(curves, parts, group)
