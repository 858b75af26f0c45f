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
    for cname, n, wb in (("Vesta", 96, 6), ("Bls381G1", 48, 5)):
        pts = ps.bases_for(cname, n)
        sc = ps.scalars_for(cname, n, 31, realistic=(cname == "Vesta"))
        bases = zk.Bases(cname, pts)
        full = zkdist.msm_sharded(bases, sc, window_bits=wb)
        lo, hi = zkdist.window_range(zk.msm_window_count(cname, n, wb), rank, world)
        assert hi > lo
        ok &= bool((zk.point_to_affine(cname, full) == orc.msm_ark(cname, pts, sc, threads=2)).all())
        bases.free()
    zk.shutdown()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


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
