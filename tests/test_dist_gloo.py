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


def _quotient_parts_worker(rank, world, port, emu_path, q):
    """the quotient AND its commitments sharded by sub-coset (bench.py at N > 1): rank r owns sub-cosets j = r mod world of 8, brings each
    to its folded coefficients (part_to_coeff), commits them over the whole SRS, and the ranks exchange the commitments (gather_rows)
    and one folded vector each (gather_stack); the pieces' commitments and the folded h(X) equal upstream's order on every rank"""
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
    name, cname, k, QP = "PallasFp", "Vesta", 2, 8
    dom = zk.halo2.EvaluationDomain(name, 9, k)
    n, ne, p = dom.n, dom.extended_len(), dom._p
    m, r_sl = ne // QP, (ne // QP) // n
    h_ext = ps.rand_field(name, ne, 4242)                     # the divided numerator on the whole coset (the same on every rank)
    pts = ps.bases_for(cname, n)
    bases = zk.Bases(cname, pts)
    unmont = lambda a: orc.limbs_to_int(orc.from_mont(name, a.reshape(1, 4))[0])
    mont = lambda v: zk.halo2._mont_limbs(v % p, p)
    my = [j for j in range(QP) if j % world == rank]
    A = [dom.part_to_coeff(np.ascontiguousarray(h_ext[j::QP]), j, QP) for j in my]
    C_loc = np.stack([zk.msm(bases, A[jj][s_ * n:(s_ + 1) * n].copy(), montgomery=True) for jj in range(len(my)) for s_ in range(r_sl)])
    max_rows = -(-QP // world) * r_sl
    C_all = zkdist.gather_rows(C_loc, max_rows)
    order = {}
    for rk in range(world):
        for jj, j in enumerate(jx for jx in range(QP) if jx % world == rk):
            for s_ in range(r_sl):
                order[(j, s_)] = rk * max_rows + jj * r_sl + s_
    used = sorted(order.values())
    C_used, order = C_all[used], {kk: used.index(v) for kk, v in order.items()}
    rows = dom.piece_scalars(QP)
    got = zk.halo2.combine_commitments(cname, C_used, [[(order[(j, s_)], sc) for j, s_, sc in terms] for _, terms in rows])
    # the folded quotient: this rank's share, then the ranks' shares added
    xn = 0x1234567 * 0x89abcdef % p
    e_js = dom.fold_scalars(QP, xn)
    share = np.zeros((n, 4), dtype=np.uint64)
    first = True
    for jj, j in enumerate(my):
        for s_ in range(r_sl):
            sl = np.ascontiguousarray(A[jj][s_ * n:(s_ + 1) * n])
            if first:
                share = zk.vec_op(name, "scale", sl.copy(), scalar=mont(e_js[j][s_]))
                first = False
            else:
                zk.halo2.vec_muladd(name, sl, share, mont(e_js[j][s_]), out=share)
    shares = np.zeros((world, n, 4), dtype=np.uint64)
    zkdist.gather_stack(share, shares)
    folded = shares[0].copy()
    for rk in range(1, world):
        zk.vec_op(name, "add", folded, shares[rk])
    # upstream's order on every rank
    coeffs = dom.extended_to_coeff(h_ext.copy())
    ok = True
    for (qi, _), gq in zip(rows, got):
        direct = zk.msm(bases, np.ascontiguousarray(coeffs[qi * n:(qi + 1) * n]), montgomery=True)
        ok &= bool((zk.point_to_affine(cname, gq) == zk.point_to_affine(cname, direct)).all())
    pieces = np.ascontiguousarray(coeffs.reshape(ne // n, n, 4))
    exp = np.zeros((n, 4), dtype=np.uint64)
    zk.halo2.vec_fold_many(name, exp, pieces, mont(xn), reverse=True)
    ok &= bool((folded == exp).all())
    bases.free()
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


@pytest.mark.parametrize("world", [2, 3])      # 3: an uneven deal of the 8 sub-cosets
def test_quotient_by_parts_gloo(world):
    _run(_quotient_parts_worker, world)


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
