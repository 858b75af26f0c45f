"""CPU: the N > 1 MSM path (window-range sharding + all_gather + point adds, contangle-zkcp_amd/dist.py)
with world_size 2 and 3 over gloo.  Each rank drives the emulator build (tests/emu, TEST
INFRASTRUCTURE) through the same C ABI; the combined result must equal the oracle bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, emu_path, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contangle_zkcp_amd as zk
    from contangle_zkcp_amd import dist as zkdist
    import parity_suite as ps
    from oracle import zk_oracle as orc
    zk.load(path=emu_path)
    zk.init(0)
    ok = True
    zkdist.SHARD_MIN_POINTS = 0          # the test inputs are small: share them anyway
    for cname, n, wb in (("Vesta", 96, 6), ("Bls381G1", 48, 5)):
        pts = ps.bases_for(cname, n)
        sc = ps.scalars_for(cname, n, 31, realistic=(cname == "Vesta"))
        bases = zk.Bases(cname, pts)
        full = zkdist.msm_sharded(bases, sc, window_bits=wb)
        lo, hi = zkdist.window_range(zk.msm_window_count(cname, n, wb), rank, world)
        assert hi > lo
        exp = orc.msm_ark(cname, pts, sc, threads=2)
        ok &= bool((zk.point_to_affine(cname, full) == exp).all())
        # the batched form (halo2's column commitments): one device call and ONE all_gather for all columns
        cols = np.stack([sc, ps.scalars_for(cname, n, 32), ps.scalars_for(cname, n, 33, realistic=True)])
        got = zkdist.msm_batch_sharded(bases, ps.to_device(zk, cols), window_bits=wb)      # 3 vectors >= world: whole MSMs per rank
        for i in range(3):
            ok &= bool((zk.point_to_affine(cname, got[i]) == orc.msm_ark(cname, pts, cols[i], threads=2)).all())
        got2 = zkdist.msm_batch_sharded(bases, ps.to_device(zk, cols[:world - 1]), window_bits=wb)   # fewer vectors than ranks: window shares
        for i in range(world - 1):
            ok &= bool((zk.point_to_affine(cname, got2[i]) == orc.msm_ark(cname, pts, cols[i], threads=2)).all())
        # several MSMs over different bases (Groth16's five), submitted together, one all_gather
        pts2 = ps.bases_for(cname, n, seed=12)
        bases2 = zk.Bases(cname, pts2)
        res = zkdist.msm_many_sharded([(bases, ps.to_device(zk, sc), False), (bases2, ps.to_device(zk, cols[1]), False),
                                       (bases, ps.to_device(zk, cols[2]), False)], window_bits=wb)
        ok &= bool((zk.point_to_affine(cname, res[0]) == exp).all())
        ok &= bool((zk.point_to_affine(cname, res[1]) == orc.msm_ark(cname, pts2, cols[1], threads=2)).all())
        ok &= bool((zk.point_to_affine(cname, res[2]) == orc.msm_ark(cname, pts, cols[2], threads=2)).all())
        # below SHARD_MIN_POINTS (the default) nothing is shared: every rank computes the whole sum, no collective
        zkdist.SHARD_MIN_POINTS = 1 << 17
        ok &= bool((zk.point_to_affine(cname, zkdist.msm_sharded(bases, sc, window_bits=wb)) == exp).all())
        got = zkdist.msm_batch_sharded(bases, ps.to_device(zk, cols), window_bits=wb)
        ok &= bool((zk.point_to_affine(cname, got[2]) == orc.msm_ark(cname, pts, cols[2], threads=2)).all())
        res = zkdist.msm_many_sharded([(bases, ps.to_device(zk, sc), False), (bases2, ps.to_device(zk, cols[1]), False)], window_bits=wb)
        ok &= bool((zk.point_to_affine(cname, res[0]) == exp).all())
        ok &= bool((zk.point_to_affine(cname, res[1]) == orc.msm_ark(cname, pts2, cols[1], threads=2)).all())
        zkdist.SHARD_MIN_POINTS = 0
        bases.free()
        bases2.free()
    # the IPA's one-step generator collapse, every rank computing its share of the survivors + one all_gather
    from oracle import pyref
    cname, k = "Vesta", 9
    sf = pyref.CURVES[cname][1]
    n = 1 << k
    gens = ps.bases_for(cname, n, seed=31)
    srs = zk.Bases(cname, gens)
    new_buffer = lambda shape: np.zeros(shape, dtype=np.uint64)
    res = []
    for sharded in (False, True):
        vipa = zk.halo2.IpaProverVirtual(cname, ps.rand_field(sf, n, 5).copy(), ps.rand_field(sf, n, 6).copy(), srs, new_buffer)
        for j in range(2):
            vipa.round()
            vipa.fold(ps.rand_field(sf, 1, 40 + j)[0])
        res.append(vipa.collapse(sharded=sharded).copy())
        L, R, vl, vr = vipa.round()                     # ... and the next round runs over the gathered generators
        res.append(np.concatenate([L, R]))
        vipa.free()
    ok &= bool((res[0] == res[2]).all() and (res[1] == res[3]).all())
    srs.free()
    zk.shutdown()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def _quotient_worker(rank, world, port, emu_path, q):
    """the extended coset in `world` sub-cosets: every rank transforms every column onto ITS sub-coset (one transform of size
    extended_len / world per column), evaluates the quotient program there (rotations stay inside the sub-coset), divides by the
    vanishing polynomial's values on it, and ONE all_gather (dist.gather_parts) assembles h -- bit-equal to the unsharded path"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contangle_zkcp_amd as zk
    from contangle_zkcp_amd import dist as zkdist, synth
    import parity_suite as ps
    zk.load(path=emu_path)
    zk.init(0)
    name, k, j = "PallasFp", 5, 9
    dom = zk.halo2.EvaluationDomain(name, j, k)
    n, ne, ek = dom.n, dom.extended_len(), dom.extended_k
    n_adv, n_fix, n_inst = 13, 8, 3
    prog = synth.quotient_program(n_adv, n_fix, n_inst)
    ncols = n_adv + n_fix + 6 + n_inst
    coeffs = [ps.rand_field(name, n, 700 + c) for c in range(ncols)]
    consts = ps.rand_field(name, 5, 777)
    # unsharded reference (same library, whole coset)
    whole = []
    for c in coeffs:
        buf = np.zeros((ne, 4), dtype=np.uint64)
        dom.coeff_to_extended(buf, coeffs=c.copy())
        whole.append(buf)
    h_ref = np.zeros((ne, 4), dtype=np.uint64)
    zk.halo2.evaluate_expression(name, prog, whole, consts, ek, 1 << (ek - k), h_ref)
    dom.divide_by_vanishing_poly(h_ref)
    # this rank's sub-coset
    m = dom.part_len(world)
    mine = []
    for c in coeffs:
        buf = np.zeros((m, 4), dtype=np.uint64)
        dom.coeff_to_extended_part(c.copy(), buf, rank, world)
        mine.append(buf)
    h_part = np.zeros((m, 4), dtype=np.uint64)
    zk.halo2.evaluate_expression(name, prog, mine, consts, m.bit_length() - 1, dom.rot_scale_part(world), h_part)
    dom.divide_by_vanishing_poly_part(h_part, rank, world)
    h = np.zeros((ne, 4), dtype=np.uint64)
    zkdist.gather_parts(h_part, h)
    ok = bool((h == h_ref).all()) and bool((h_part == h_ref[rank::world]).all())
    # ... and the coefficients of h agree as well
    a, b = h.copy(), h_ref.copy()
    dom.extended_to_coeff(a)
    dom.extended_to_coeff(b)
    ok &= bool((a == b).all())
    zk.shutdown()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def _run(worker, world):
    import importlib.util
    spec = importlib.util.spec_from_file_location("zk_build", os.path.join(ROOT, "contangle-zkcp_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    emu = b.build_emu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, emu, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res == [(r, True) for r in range(world)]


@pytest.mark.parametrize("world", [2, 4])
def test_quotient_coset_sharded_gloo(world):
    _run(_quotient_worker, world)


@pytest.mark.parametrize("world", [2, 3])
def test_msm_sharded_gloo(world):
    import importlib.util
    spec = importlib.util.spec_from_file_location("zk_build", os.path.join(ROOT, "contangle-zkcp_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    emu = b.build_emu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, emu, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res == [(r, True) for r in range(world)]
