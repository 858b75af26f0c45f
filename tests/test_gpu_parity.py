"""GPU tier (-m gpu): the HIP library on a real MI355X, through the C ABI, against the CPU oracle
and the pure-Python golden fixtures.  Bit-exact bar (integer arithmetic).  Nothing here reads
/root/reference.  Full-size (2^20) cases use size-independent properties plus the oracle's
threaded restatement where it finishes in seconds."""
import os

import numpy as np
import pytest

import parity_suite as ps
from oracle import zk_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    import torch
    assert torch.cuda.is_available(), "no GPU visible"
    import contangle_zkcp_amd as zk
    zk._lib = None
    zk.load()                       # the in-tree HIP library; raises if missing
    zk.init(0)
    info = zk.backend_info()
    assert info.startswith("hip gfx950"), info
    yield zk
    zk.shutdown()


def test_ntt_golden(zk):
    ps.check_ntt_golden(zk)


@pytest.mark.parametrize("name", ps.NTT_FIELDS)
@pytest.mark.parametrize("logn", [1, 2, 5, 9, 10, 11, 13, 14])
def test_ntt_vs_oracle(zk, name, logn):
    ps.check_ntt_vs_oracle(zk, name, logn)


@pytest.mark.parametrize("max_logr,logt", [("3", "1"), ("5", "2"), ("7", "3")])
def test_ntt_multipass_plans(zk, max_logr, logt):
    zk.ntt_configure(max_log_radix=int(max_logr), log_tile=int(logt))
    try:
        for name, logn in (("PallasFp", 12), ("Bls381Fr", 11), ("PallasFq", 9), ("Bn254Fr", 10)):
            if (logn + int(max_logr) - 1) // int(max_logr) > 4:
                continue
            ps.check_ntt_vs_oracle(zk, name, logn)
    finally:
        zk.ntt_configure()


@pytest.mark.parametrize("name,logn", [("PallasFp", 20), ("PallasFq", 20), ("Bls381Fr", 20), ("Bn254Fr", 18), ("PallasFp", 22)])
def test_ntt_full_size(zk, name, logn):
    """2^20 (BASELINE configs[1]) and 2^22: bit-exact vs the oracle's best_fft, ifft(fft(x)) = x,
    linearity: NTT(a + b) = NTT(a) + NTT(b) checked through a checksum of limbs."""
    ps.check_ntt_vs_oracle(zk, name, logn, threads=16)


@pytest.mark.parametrize("name,logn", [("Bls381Fr", 9), ("PallasFp", 13), ("Bn254Fr", 16), ("PallasFq", 20), ("Bls381Fr", 22)])
def test_ntt_fused_coset(zk, name, logn):
    ps.check_ntt_fused_coset(zk, name, logn, threads=16)


def test_ntt_device_tensor_path(zk):
    """torch-allocated HBM buffer + torch stream through zk_ntt_device (the bench path)."""
    import torch
    name, logn = "PallasFp", 16
    a = ps.rand_field(name, 1 << logn, 21)
    w = orc.root_of_unity(name, logn)
    d = torch.from_numpy(a.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    zk.ntt(name, d, w, stream=st)
    torch.cuda.synchronize()
    got = d.cpu().numpy().view(np.uint64)
    assert (got == orc.halo2_best_fft(name, a, w, logn, threads=8)).all()


@pytest.mark.parametrize("name", ps.NTT_FIELDS)
def test_vec_ops(zk, name):
    ps.check_vec_ops(zk, name, 5000)


@pytest.mark.parametrize("name,logm", [("Bls381Fr", 10), ("Bls381Fr", 16), ("Bn254Fr", 13), ("PallasFp", 12), ("Bls381Fr", 20)])
def test_groth16_witness_map(zk, name, logm):
    """ark-groth16 witness_map (7 NTTs + pointwise glue) entirely in HBM vs the oracle's restatement; 2^20 is the
    domain of the reference's own largest test (circuits-ark/src/encryption.rs:379)."""
    ps.check_witness_map(zk, name, logm, threads=16)


def test_msm_golden(zk):
    ps.check_msm_golden(zk)


@pytest.mark.parametrize("cname", ps.CURVES)
def test_msm_edges(zk, cname):
    ps.check_msm_edges(zk, cname)


@pytest.mark.parametrize("cname", ps.CURVES)
@pytest.mark.parametrize("n,wb,realistic", [(1000, 0, False), (4096, 0, True), (5000, 11, False), (1 << 14, 0, False),
                                            (1 << 14, 13, True), (100003, 0, False)])
def test_msm_vs_oracle(zk, cname, n, wb, realistic):
    ps.check_msm_vs_oracle(zk, cname, n, wb, realistic)


@pytest.mark.parametrize("cname,parts", [("Vesta", 2), ("Pallas", 4), ("Bls381G1", 8), ("Bn254G1", 4)])
def test_msm_window_sharding(zk, cname, parts):
    ps.check_msm_window_sharding(zk, cname, 1 << 12, 0, parts)
    ps.check_msm_window_sharding(zk, cname, 3000, 16, parts)


@pytest.mark.parametrize("cname", ps.CURVES)
def test_msm_big_buckets(zk, cname):
    ps.check_msm_big_buckets(zk, cname)
    ps.check_msm_big_buckets(zk, cname, n=9000, window_bits=13)


@pytest.mark.parametrize("cname", ["Vesta", "Bn254G1", "Bls381G1", "Bls381G2"])
def test_msm_sort_shapes(zk, cname):
    ps.check_msm_sort_shapes(zk, cname, 1 << 13, (0, 9))
    ps.check_msm_sort_shapes(zk, cname, 70000, (16, 14, 12))   # 16 / 4 / 1 bucket ranges per window, 69 scalar blocks


@pytest.mark.parametrize("cname", ["Vesta", "Bn254G1", "Bls381G1", "Bn254G2"])
def test_msm_bucket_splitting(zk, cname):
    ps.check_msm_split(zk, cname, 1 << 13, 0)
    ps.check_msm_split(zk, cname, 50000, 12, realistic=False)


def _device_bases(zk, cname, n, seed=77):
    """bases generated on the GPU: P_i = [k_i]G via zk_fixed_base_mul_device; spot-checked on the oracle."""
    import torch
    ks = ps.scalars_for(cname, n, seed)
    nl = zk.base_limbs(cname)
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    d_pts = torch.empty((n, 2 * nl), dtype=torch.int64, device="cuda")
    zk.fixed_base_mul_device(cname, d_k, d_pts, n)
    torch.cuda.synchronize()
    return ks, d_pts


@pytest.mark.parametrize("cname", ps.CURVES)
def test_fixed_base_mul(zk, cname):
    ks, d_pts = _device_bases(zk, cname, 512)
    got = d_pts.cpu().numpy().view(np.uint64)
    assert (got == orc.fixed_base_mul(cname, ks, threads=8)).all()


@pytest.mark.parametrize("cname", ["Vesta", "Bls381G1", "Bls381G2"])
def test_msm_slice_lengths(zk, cname):
    ps.check_msm_slice_lengths(zk, cname, 1 << 13, 12)


@pytest.mark.parametrize("cname,n,wb,count", [("Vesta", 1 << 16, 16, 3), ("Pallas", 1 << 18, 0, 0), ("Bls381G1", 1 << 14, 12, 2),
                                              ("Bn254G2", 1 << 13, 10, 0), ("Vesta", 1 << 20, 0, 5),
                                              # windows wider than 16 bits (32-bit digit codes; 0 = the form's own choice: 18 / 20 bits above)
                                              ("Vesta", 1 << 16, 17, 3), ("Bls381G1", 1 << 16, 18, 2), ("Bn254G1", 1 << 17, 20, 0),
                                              ("Bls381G2", 1 << 14, 19, 0)])
def test_msm_precomputed_table(zk, cname, n, wb, count):
    ps.check_msm_precomputed(zk, cname, n, wb, realistic=(cname == "Bls381G1"), count=count)


@pytest.mark.parametrize("cname,n,wb", [("Vesta", 1 << 15, 0), ("Bn254G1", 70000, 13), ("Bls381G2", 5000, 9)])
def test_msm_window_groups(zk, cname, n, wb):
    ps.check_msm_window_groups(zk, cname, n, wb)


@pytest.mark.parametrize("cname,n,wbs", [("Vesta", 1 << 14, [2, 3, 4, 7, 10, 13, 15, 16]), ("Bls381G1", 1 << 13, [5, 12, 16]),
                                         ("Bn254G2", 1 << 12, [3, 9, 14, 16]), ("Bls381G2", 1 << 12, [8, 15, 16])])
def test_msm_axis_reduce(zk, cname, n, wbs):
    ps.check_msm_axis_reduce(zk, cname, n, wbs)


@pytest.mark.parametrize("cname", ps.CURVES)
def test_msm_device_side_partial_conversion(zk, cname):
    """fe29_to_std on the device (one lane of the reduction's last kernel) against the host's, all six curves"""
    ps.check_msm_device_partials(zk, cname, n=5000)


@pytest.mark.parametrize("name,log_in,logn", [("PallasFp", 10, 13), ("Bls381Fr", 17, 20), ("Bn254Fr", 12, 12), ("PallasFq", 18, 21)])
def test_ntt_extend(zk, name, log_in, logn):
    ps.check_ntt_extend(zk, name, log_in, logn, threads=32)


@pytest.mark.parametrize("cname", ps.CURVES)
def test_fixed_base_msm(zk, cname):
    ps.check_fixed_base_msm(zk, cname, 1003)


def test_fixed_base_msm_matches_plain_kernel_2p16(zk):
    """windowed fixed-base path vs the plain double-and-add kernel at 2^16 (bit-exact affine output)"""
    import torch
    n = 1 << 16
    ks, d_plain = _device_bases(zk, "Bls381G1", n, seed=5)
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    d_win = torch.empty_like(d_plain)
    zk.fixed_base_msm_device("Bls381G1", d_k, d_win, n)
    torch.cuda.synchronize()
    assert torch.equal(d_win, d_plain)


@pytest.mark.parametrize("cname", ["Pallas", "Vesta"])
def test_msm_full_size_2p20(zk, cname):
    """BASELINE configs[1]: 2^20-point MSM.  Bit-exact vs the oracle's ark restatement (threads = host cores),
    plus the structural identity MSM(s, [k_i G]) = [sum s_i k_i] G."""
    import torch
    n = 1 << 20
    ks, d_pts = _device_bases(zk, cname, n)
    pts = d_pts.cpu().numpy().view(np.uint64)
    for i in (0, 1, n // 2, n - 1):
        assert orc.on_curve(cname, pts[i])
    sc = ps.scalars_for(cname, n, 123)
    bases = zk.Bases(cname, device_tensor=d_pts, n=n)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    got = zk.point_to_affine(cname, zk.msm(bases, d_sc, stream=torch.cuda.current_stream().cuda_stream))
    # identity: sum s_i k_i mod r, computed with Python ints
    from oracle import pyref
    r = pyref.FIELDS[pyref.CURVES[cname][1]][0]
    tot = sum(a * b for a, b in zip(orc.array_to_ints(sc), orc.array_to_ints(ks))) % r
    exp = orc.scalar_mul(cname, orc.curve_generator(cname), orc.int_to_limbs(tot, 4))
    assert (got == exp).all()
    # and the threaded oracle restatement of ark-ec's Pippenger on the same inputs
    exp2 = orc.msm_ark(cname, pts, sc, threads=os.cpu_count() or 8)
    assert (got == exp2).all()
    # realistic (0/1-heavy) witness mix
    sc2 = ps.scalars_for(cname, n, 124, realistic=True)
    got2 = zk.point_to_affine(cname, zk.msm(bases, torch.from_numpy(sc2.view(np.int64)).cuda()))
    tot2 = sum(a * b for a, b in zip(orc.array_to_ints(sc2), orc.array_to_ints(ks))) % r
    assert (got2 == orc.scalar_mul(cname, orc.curve_generator(cname), orc.int_to_limbs(tot2, 4))).all()
    bases.free()


def test_msm_2p24_identity(zk):
    """2^24 points (16x the headline size: u32 entry positions reach 2^28, 1 GB of bases): the size-independent identity
    MSM(s, [k_i G]) = [sum s_i k_i mod r] G, with the bases from the windowed fixed-base path; whole and window-sharded."""
    import torch
    from oracle import pyref
    cname, n = "Vesta", 1 << 24
    ks = ps.scalars_for(cname, n, 2024)
    sc = ps.scalars_for(cname, n, 2025)
    d_pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    zk.fixed_base_msm_device(cname, torch.from_numpy(ks.view(np.int64)).cuda(), d_pts, n)
    torch.cuda.synchronize()
    bases = zk.Bases(cname, device_tensor=d_pts, n=n)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    r = pyref.FIELDS[pyref.CURVES[cname][1]][0]
    # sum s_i k_i mod r on 64-bit limbs would overflow numpy: Python ints over object arrays, chunked
    a = sc.astype(object)
    b = ks.astype(object)
    sv = a[:, 0] + (a[:, 1] << 64) + (a[:, 2] << 128) + (a[:, 3] << 192)
    kv = b[:, 0] + (b[:, 1] << 64) + (b[:, 2] << 128) + (b[:, 3] << 192)
    tot = int((sv * kv).sum() % r)
    exp = orc.scalar_mul(cname, orc.curve_generator(cname), orc.int_to_limbs(tot, 4))
    got = zk.point_to_affine(cname, zk.msm(bases, d_sc))
    assert (got == exp).all()
    W = zk.msm_window_count(cname, n)
    lo = zk.msm(bases, d_sc, windows=(0, W // 4))
    hi = zk.msm(bases, d_sc, windows=(W // 4, W))
    assert (zk.point_to_affine(cname, zk.point_add(cname, lo, hi)) == exp).all()
    bases.free()


def test_msm_2p22_bn254(zk):
    """BASELINE configs[3] scale: 2^22-point BN254 G1 MSM (Groth16-sized), structural identity + the threaded oracle."""
    import torch
    from oracle import pyref
    cname, n = "Bn254G1", 1 << 22
    ks, d_pts = _device_bases(zk, cname, n)
    sc = ps.scalars_for(cname, n, 321)
    bases = zk.Bases(cname, device_tensor=d_pts, n=n)
    got = zk.point_to_affine(cname, zk.msm(bases, torch.from_numpy(sc.view(np.int64)).cuda()))
    assert zk.msm_last_profile()["limb_bits"] == 29
    r = pyref.FIELDS[pyref.CURVES[cname][1]][0]
    tot = sum(a * b for a, b in zip(orc.array_to_ints(sc), orc.array_to_ints(ks))) % r
    assert (got == orc.scalar_mul(cname, orc.curve_generator(cname), orc.int_to_limbs(tot, 4))).all()
    pts = d_pts.cpu().numpy().view(np.uint64)
    assert (got == orc.msm_ark(cname, pts, sc, threads=min(32, os.cpu_count() or 8))).all()
    bases.free()


def test_msm_full_size_bls12_381(zk):
    """the reference's curve (lib/src/lib.rs:21-24) at the domain size of its largest test (2^20): G1 on 14 x 28-bit lazy limbs"""
    import torch
    from oracle import pyref
    cname, n = "Bls381G1", 1 << 20
    ks, d_pts = _device_bases(zk, cname, n)
    sc = ps.scalars_for(cname, n, 654, realistic=True)
    bases = zk.Bases(cname, device_tensor=d_pts, n=n)
    got = zk.point_to_affine(cname, zk.msm(bases, torch.from_numpy(sc.view(np.int64)).cuda()))
    assert zk.msm_last_profile()["limb_bits"] == 29
    r = pyref.FIELDS[pyref.CURVES[cname][1]][0]
    tot = sum(a * b for a, b in zip(orc.array_to_ints(sc), orc.array_to_ints(ks))) % r
    assert (got == orc.scalar_mul(cname, orc.curve_generator(cname), orc.int_to_limbs(tot, 4))).all()
    bases.free()


# ------------------------------------------------------------------ deferred results, batches, streams, threads
@pytest.mark.parametrize("cname,n", [("Vesta", 1 << 16), ("Bls381G1", 5000), ("Bn254G2", 3000)])
def test_msm_deferred_results(zk, cname, n):
    ps.check_msm_async(zk, cname, n)


@pytest.mark.parametrize("cname,n,count", [("Vesta", 1 << 16, 6), ("Pallas", 3000, 3), ("Bls381G2", 2000, 2)])
def test_msm_batch(zk, cname, n, count):
    ps.check_msm_batch(zk, cname, n, count)


def test_ntt_two_streams_do_not_share_scratch(zk):
    """two multi-pass NTTs (and two witness maps) enqueued on two streams run concurrently: each stream owns its
    ping-pong buffer (the process-global scratch raced silently)"""
    import torch
    name, logn = "PallasFp", 18
    n = 1 << logn
    a, b = ps.rand_field(name, n, 61), ps.rand_field(name, n, 62)
    w = orc.root_of_unity(name, logn)
    exp_a = orc.halo2_best_fft(name, a, w, logn, threads=16)
    exp_b = orc.halo2_best_fft(name, b, w, logn, threads=16)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
        torch.cuda.synchronize()
        for _rep in range(1):
            zk.ntt(name, da, w, stream=s1.cuda_stream)
            zk.ntt(name, db, w, stream=s2.cuda_stream)
        torch.cuda.synchronize()
        assert (da.cpu().numpy().view(np.uint64) == exp_a).all()
        assert (db.cpu().numpy().view(np.uint64) == exp_b).all()


def test_abi_from_a_second_host_thread(zk):
    """the caller may arrive on any host thread (rayon workers upstream): every entry point binds the thread to its
    device first (hipSetDevice was only called in zk_init)"""
    import threading
    err = []

    def worker():
        try:
            ps.check_msm_vs_oracle(zk, "Vesta", 4096, 0, True)
            ps.check_ntt_vs_oracle(zk, "Bls381Fr", 13)
            ps.check_witness_map(zk, "Bn254Fr", 10)
        except BaseException as e:   # noqa: BLE001
            err.append(e)

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert not err, err


def test_invalid_arguments_are_refused(zk):
    """sizes are validated before anything is allocated from them (a bad log_n reached hipMalloc / a 64-bit shift)"""
    import ctypes
    import torch
    lib = zk.load()
    d = torch.zeros((16, 4), dtype=torch.int64, device="cuda")
    w = zk.root_of_unity("Bn254Fr", 4)
    g = zk.multiplicative_generator("Bn254Fr")
    vp = ctypes.c_void_p
    for bad in (29, 31, 40, 64, 200):   # Bn254Fr has two-adicity 28
        assert lib.zk_ntt_coset_device(2, vp(d.data_ptr()), bad, zk._ptr(w), 0, zk._ptr(g), None, None) == -1
        assert lib.zk_ntt_extend_device(2, vp(d.data_ptr()), bad, 2, zk._ptr(w), 0, zk._ptr(g), None, None) == -1
        assert lib.zk_ntt_device(2, vp(d.data_ptr()), bad, zk._ptr(w), 0, None) == -1
        assert lib.zk_groth16_witness_map_device(2, vp(d.data_ptr()), vp(d.data_ptr()), vp(d.data_ptr()), bad, None) == -1


@pytest.mark.parametrize("name,k,j", [("PallasFp", 10, 9), ("PallasFq", 11, 9), ("PallasFp", 9, 5), ("Bls381Fr", 8, 3)])
def test_halo2_domain(zk, name, k, j):
    ps.check_halo2_domain(zk, name, k, j)


@pytest.mark.parametrize("name,cname,k,parts", [("PallasFp", "Vesta", 7, 8), ("PallasFp", "Vesta", 6, 4), ("PallasFq", "Pallas", 6, 2), ("PallasFp", "Vesta", 5, 1),
                                                ("PallasFp", "Vesta", 11, 8)])
def test_quotient_by_parts(zk, name, cname, k, parts):
    ps.check_quotient_by_parts(zk, name, cname, k, parts)


# ------------------------------------------------------------------ BASELINE configs[2] / [3] / [4] at their full sizes
def test_halo2_extended_domain_2p23(zk):
    """configs[2]: halo2 0.2 EvaluationDomain on a 2^20-row circuit with a degree-9 gate set: coeff_to_extended 2^20 -> 2^23
    (zero-extension + coset shift + NTT, three-pass plan) and extended_to_coeff at 2^23, against the oracle's restatement"""
    import torch
    name, k, ext = "PallasFp", 20, 23
    n, ne = 1 << k, 1 << ext
    coeffs = ps.rand_field(name, n, 0xE17)
    w_ext = orc.root_of_unity(name, ext)
    g = orc.field_generator(name)      # the coset shift; halo2's ZETA shift is covered by test_halo2_domain below
    padded = np.zeros((ne, 4), dtype=np.uint64)
    padded[:n] = coeffs
    exp = orc.ark_fft(name, padded, "coset_fft", threads=os.cpu_count() or 8)
    d = torch.empty((ne, 4), dtype=torch.int64, device="cuda")
    d.fill_(-1)                                                   # the padding is never read
    d[:n].copy_(torch.from_numpy(coeffs.view(np.int64)))
    zk.halo2.coeff_to_extended(name, d, k, w_ext, g)
    torch.cuda.synchronize()
    got = d.cpu().numpy().view(np.uint64)
    assert (got == exp).all()
    # the same transform stored sub-coset by sub-coset (ZK_NTT_OUT_SUBCOSETS) and every sub-coset as its own transform of size 2^20,
    # out of place from the coefficients (what the sharded quotient and a rank of an 8-GPU run use), in both output radices
    dom = zk.halo2.EvaluationDomain(name, 9, k)
    d_co = torch.from_numpy(coeffs.view(np.int64)).cuda()
    d2 = torch.empty((ne, 4), dtype=torch.int64, device="cuda")
    d_nat = torch.empty((ne, 4), dtype=torch.int64, device="cuda")
    dom.coeff_to_extended(d_nat, coeffs=d_co)                      # halo2's own shift (ZETA), natural order
    for lazy in (False, True):
        if lazy:
            zk.halo2.to_lazy_form(name, d_nat)
        d2.fill_(-1)
        dom.coeff_to_extended(d2, coeffs=d_co, parts=8, lazy_out=lazy)
        assert bool((d2.view(8, n, 4) == d_nat.view(n, 8, 4).permute(1, 0, 2)).all()), ("parts=8", lazy)
        for part in (0, 5):
            d_pt = torch.empty((n, 4), dtype=torch.int64, device="cuda")
            dom.coeff_to_extended_part(d_co, d_pt, part, 8, lazy_out=lazy)
            assert bool((d_pt == d_nat.view(n, 8, 4)[:, part]).all()), ("part", part, lazy)
    # ... and back: part_to_coeff of a sub-coset's values of a polynomial of degree < n returns its coefficients
    d_pt = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    dom.coeff_to_extended_part(d_co, d_pt, 3, 8)
    dom.part_to_coeff(d_pt, 3, 8)
    assert bool((d_pt == d_co).all())
    del d2, d_nat, d_pt
    # extended_to_coeff: inverse transform + coset un-shift brings the padded coefficients back
    winv, ginv = orc.fe_op(name, "inv", w_ext), orc.fe_op(name, "inv", g)
    zk.ntt(name, d, winv, scale_by_n_inv=True, coset_post=ginv)
    torch.cuda.synchronize()
    back = d.cpu().numpy().view(np.uint64)
    assert (back == padded).all()
    # and on a dense 2^23 vector against the oracle's coset_ifft
    dense = ps.rand_field(name, ne, 0xE18)
    d.copy_(torch.from_numpy(dense.view(np.int64)))
    zk.ntt(name, d, winv, scale_by_n_inv=True, coset_post=ginv)
    torch.cuda.synchronize()
    assert (d.cpu().numpy().view(np.uint64) == orc.ark_fft(name, dense, "coset_ifft", threads=os.cpu_count() or 8)).all()


def test_quotient_by_parts_2p23(zk):
    """configs[2] at its own shape: a quotient of 2^23 extended values through its 8 sub-cosets (part_to_coeff + the mixing scalars)
    equals extended_to_coeff coefficient for coefficient (device-side comparison of all 2^23), and the combination of the sub-cosets'
    commitments equals the commitments of the pieces (2^20-point MSMs)"""
    import torch
    name, cname, k, parts = "PallasFp", "Vesta", 20, 8
    dom = zk.halo2.EvaluationDomain(name, 9, k)
    n, ne, p = dom.n, dom.extended_len(), dom._p
    h_ext = torch.from_numpy(ps.rand_field(name, ne, 0x51DE).view(np.int64)).cuda()
    direct = h_ext.clone()
    dom.extended_to_coeff(direct)                                              # upstream's path: [8 pieces, n]
    A = h_ext.view(n, parts, 4).permute(1, 0, 2).contiguous()                  # sub-coset j = the extended values i * 8 + j
    for j in range(parts):
        dom.part_to_coeff(A[j], j, parts)
    c = dom.part_mix(parts)
    mont = lambda v: zk.halo2._mont_limbs(v % p, p)
    acc = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    for i in range(parts):                                                     # h^(i) = sum_j c[i][j] A_j, on the device
        acc.copy_(A[0])
        zk.vec_op(name, "scale", acc, scalar=mont(c[i][0]))
        for j in range(1, parts):
            zk.halo2.vec_muladd(name, A[j], acc, mont(c[i][j]), out=acc)
        assert bool((acc == direct.view(parts, n, 4)[i]).all()), ("piece", i)
    # commitments: linear in the folded coefficients
    pts = ps.bases_for(cname, n)
    bases = zk.Bases(cname, pts)
    C = zk.msm_batch(bases, A, montgomery=True)
    rows = dom.piece_scalars(parts)
    got = zk.halo2.combine_commitments(cname, list(C), [[(j, sc) for j, s_, sc in terms] for _, terms in rows[:3]],
                                       to_device=lambda arr: torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)).cuda())
    exp = zk.msm_batch(bases, direct.view(parts, n, 4)[:3].contiguous(), montgomery=True)
    for q in range(3):
        assert (zk.point_to_affine(cname, got[q]) == zk.point_to_affine(cname, exp[q])).all(), ("commitment", q)
    bases.free()


def test_groth16_bn254_2p22(zk):
    """configs[3]: BN254 Fr at the 2^22 domain: the NTT and the whole witness map (7 NTTs + glue) vs the oracle"""
    thr = os.cpu_count() or 8
    ps.check_ntt_vs_oracle(zk, "Bn254Fr", 22, threads=thr)
    ps.check_ntt_fused_coset(zk, "Bn254Fr", 22, threads=thr)
    ps.check_witness_map(zk, "Bn254Fr", 22, threads=thr)


def _identity_msm(zk, cname, n, seed, realistic, oracle_too):
    """MSM(s, [k_i G]) = [sum s_i k_i mod r] G at any size; bases from the windowed fixed-base path on the GPU"""
    import torch
    from oracle import pyref
    ks = ps.scalars_for(cname, n, seed)
    sc = ps.scalars_for(cname, n, seed + 1, realistic=realistic)
    nl = zk.base_limbs(cname)
    d_pts = torch.empty((n, 2 * nl), dtype=torch.int64, device="cuda")
    zk.fixed_base_msm_device(cname, torch.from_numpy(ks.view(np.int64)).cuda(), d_pts, n)
    torch.cuda.synchronize()
    bases = zk.Bases(cname, device_tensor=d_pts, n=n)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    got = zk.point_to_affine(cname, zk.msm(bases, d_sc))
    assert zk.msm_last_profile()["limb_bits"] == 29
    r = pyref.FIELDS[pyref.CURVES[cname][1]][0]
    a, b = sc.astype(object), ks.astype(object)
    sv = a[:, 0] + (a[:, 1] << 64) + (a[:, 2] << 128) + (a[:, 3] << 192)
    kv = b[:, 0] + (b[:, 1] << 64) + (b[:, 2] << 128) + (b[:, 3] << 192)
    tot = int((sv * kv).sum() % r)
    exp = orc.scalar_mul(cname, orc.curve_generator(cname), orc.int_to_limbs(tot, 4))
    assert (got == exp).all(), (cname, n)
    if oracle_too:
        pts = d_pts.cpu().numpy().view(np.uint64)
        assert (got == orc.msm_ark(cname, pts, sc, threads=min(64, os.cpu_count() or 8))).all(), (cname, n, "oracle")
    return bases, d_sc, exp


@pytest.mark.parametrize("cname,logn,oracle_too", [("Bn254G2", 20, True), ("Bls381G2", 20, True), ("Bn254G2", 22, False),
                                                   ("Bls381G2", 22, False)])
def test_msm_g2_full_size(zk, cname, logn, oracle_too):
    """configs[3]: Groth16's b_g2_query MSM at 2^20 (the domain of the reference's largest test) and 2^22, realistic witness"""
    bases, _, _ = _identity_msm(zk, cname, 1 << logn, 4000 + logn, True, oracle_too)
    bases.free()


@pytest.mark.parametrize("cname", ["Vesta", "Pallas"])
def test_msm_pasta_2p22_window_shares(zk, cname):
    """configs[4]: 2^22-point Pasta MSM, whole and as the 8 window shares of an 8-GPU run added up (one GPU plays all
    ranks; the collective itself is covered by tests/test_dist_gloo.py and tests/test_multi_device_emu.py)"""
    n = 1 << 22
    bases, d_sc, exp = _identity_msm(zk, cname, n, 5100, False, False)
    W = zk.msm_window_count(cname, n)
    acc = None
    for g in range(8):
        lo, hi = W * g // 8, W * (g + 1) // 8
        part = zk.msm(bases, d_sc, windows=(lo, hi))
        acc = part if acc is None else zk.point_add(cname, acc, part)
    assert (zk.point_to_affine(cname, acc) == exp).all()
    bases.free()


def test_msm_bls12_381_g1_2p22(zk):
    """the reference's curve at configs[3]'s size"""
    bases, _, _ = _identity_msm(zk, "Bls381G1", 1 << 22, 5200, True, False)
    bases.free()


@pytest.mark.parametrize("pairing,g1,g2", [("Bls381", "Bls381G1", "Bls381G2"), ("Bn254", "Bn254G1", "Bn254G2")])
def test_proving_key_file_to_resident_bases(zk, pairing, g1, g2):
    """f3 -> a8: a ProvingKey in the reference's on-disk form (serialize_unchecked, lib/src/utils.rs:85-102) is indexed, its
    query vectors are decoded straight into resident bases handles, and MSMs over them match the oracle"""
    az = zk.ark_serialize
    n = 3000
    a_q, b2_q = ps.bases_for(g1, n, seed=21), ps.bases_for(g2, n, seed=22)
    one1, one2 = ps.bases_for(g1, 1, seed=23), ps.bases_for(g2, 1, seed=24)
    a_q[5] = 0                                                     # an identity element inside a query vector
    members = {"alpha_g1": one1, "beta_g2": one2, "gamma_g2": one2, "delta_g2": one2, "gamma_abc_g1": a_q[:3], "beta_g1": one1,
               "delta_g1": one1, "a_query": a_q, "b_g1_query": a_q[::-1].copy(), "b_g2_query": b2_q, "h_query": a_q[: n - 1], "l_query": a_q[:100]}
    buf = az.ProvingKey.serialize_unchecked(pairing, members)
    pk = az.ProvingKey.deserialize_unchecked(pairing, buf)
    sc = ps.scalars_for(g1, n, 77, realistic=True)
    for name, curve, pts in (("a_query", g1, a_q), ("b_g2_query", g2, b2_q)):
        bases = pk.upload(name)
        assert bases.n == n
        got = zk.point_to_affine(curve, zk.msm(bases, sc))
        assert (got == orc.msm_ark(curve, pts, sc, threads=8)).all(), name
        bases.free()
    # ark-groth16 uses a_query[1..] with the witness and adds a_query[0] (the constant-one wire) separately
    tail = pk.upload("a_query", skip_first=1)
    got = zk.point_to_affine(g1, zk.msm(tail, sc[1:]))
    assert (got == orc.msm_ark(g1, a_q[1:], sc[1:], threads=8)).all()
    tail.free()


@pytest.mark.parametrize("pairing,nc,long_rows", [("Bls381", 120, (70, 110)), ("Bn254", 60, (40,))])
def test_groth16_prove_end_to_end(zk, pairing, nc, long_rows):
    """a1 + a6: assignment -> proof bytes entirely through the library (the reference's call: lib/src/zk/encryption.rs:76),
    checked against Groth16 in the exponent; long rows (> 64 terms) take the workgroup-per-row mat-vec kernel"""
    ps.check_groth16_prove(zk, pairing, num_constraints=nc, long_rows=long_rows)


# ------------------------------------------------------------------ halo2 prover steps beyond commit / FFT (f4)
@pytest.mark.parametrize("name,n", [("PallasFp", 1 << 16), ("PallasFq", 70001), ("Bn254Fr", 4097), ("Bls381Fr", 1)])
def test_halo2_batch_invert_and_scan(zk, name, n):
    ps.check_batch_invert_and_scan(zk, name, n)


@pytest.mark.parametrize("name,k", [("PallasFp", 10), ("PallasFq", 7)])
def test_halo2_permutation_and_lookup_products(zk, name, k):
    ps.check_permutation_and_lookup_products(zk, name, k)


def test_halo2_permute_expression_pair(zk):
    ps.check_permute_expression_pair(zk, "PallasFp", n=5000, usable=4990)


@pytest.mark.parametrize("cname,k", [("Vesta", 8), ("Pallas", 12)])
def test_halo2_ipa_collapse_edges(zk, cname, k):
    ps.check_ipa_collapse_edges(zk, cname, k)


@pytest.mark.parametrize("name", ["PallasFp", "PallasFq", "Bn254Fr"])
def test_halo2_eval_polynomial(zk, name):
    ps.check_eval_polynomial(zk, name, sizes=(1, 17, 4099, (1 << 16) + 3))


@pytest.mark.parametrize("name", ["PallasFp", "PallasFq", "Bn254Fr"])
def test_halo2_kate_division(zk, name):
    ps.check_kate_division(zk, name, sizes=(1, 2, 15, 16, 17, 4095, 4096, 4097, 9000, 70001))


@pytest.mark.parametrize("name,k", [("PallasFp", 20), ("PallasFq", 17), ("PallasFp", 23)])
def test_halo2_kate_division_at_size(zk, name, k):
    ps.check_kate_division_at_size(zk, name, k)


@pytest.mark.parametrize("cname,k", [("Vesta", 6), ("Pallas", 5), ("Bn254G1", 4), ("Bls381G1", 3)])   # the argument is halo2's (Pasta); the entry points take every curve
def test_halo2_ipa(zk, cname, k):
    ps.check_ipa(zk, cname, k)


@pytest.mark.parametrize("name,k,ext", [("PallasFp", 8, 3), ("PallasFq", 6, 3), ("Bn254Fr", 6, 2), ("Bls381Fr", 5, 2)])
def test_halo2_expression(zk, name, k, ext):
    """saturated interpreter, lazy-limb interpreter and the kernel compiled for the program (one hiprtc build per field unit)"""
    ps.check_expression(zk, name, k, ext=ext)


def test_halo2_expression_kernel_falls_back_to_the_interpreter(zk):
    """the compiled quotient kernel needs the field headers next to the library: when they cannot be found the hiprtc build fails and
    the interpreter kernel runs instead -- same results (a program no other test uses: a failed build is remembered per program)"""
    name, k = "PallasFp", 7
    ne = 1 << k
    cols = [ps.rand_field(name, ne, 0xFA11 + c) for c in range(3)]
    consts = ps.rand_field(name, 2, 0xFA20)
    a, b, c = (("col", i, 0) for i in range(3))
    prog = [a, b, ("mul",), c, ("scale", 1), ("sub",), ("col", 0, 2), ("mul",), ("const", 0), ("add",), b, ("neg",), ("mul",)]
    d_cols = [ps.to_device(zk, x) for x in cols]
    ref = ps.to_device(zk, np.zeros((ne, 4), dtype=np.uint64))
    zk.halo2.evaluate_expression(name, prog, d_cols, consts, k, 1, ref)
    for d in d_cols:
        zk.halo2.to_lazy_form(name, d)
    saved = os.environ.get("ZKCP_AMD_CSRC")
    os.environ["ZKCP_AMD_CSRC"] = "/nonexistent/csrc"
    try:
        zk.halo2.expr_configure("always")
        out = ps.to_device(zk, np.zeros((ne, 4), dtype=np.uint64))
        zk.halo2.evaluate_expression(name, prog, d_cols, consts, k, 1, out, lazy=True)
        assert (ps.to_host(zk, out) == ps.to_host(zk, ref)).all()
    finally:
        zk.halo2.expr_configure("auto")
        if saved is None:
            del os.environ["ZKCP_AMD_CSRC"]
        else:
            os.environ["ZKCP_AMD_CSRC"] = saved


def test_halo2_expression_2p23(zk):
    """configs[2] at its own shape: the bench's 268-op quotient program over 30 columns of 2^23 extended rows (k = 20, degree 9),
    sampled rows (incl. the rows whose rotations wrap) against Python integers"""
    assert ps.check_expression_at_size(zk, "PallasFp", 20, 3, samples=96) >= 96
    assert ps.check_expression_at_size(zk, "PallasFp", 20, 3, samples=96, parts=8) >= 96     # one rank's sub-coset of an 8-GPU run


def test_halo2_ipa_2p20_with_collapse(zk):
    """configs[2] at its own shape: the 20-round opening argument over 2^20 generators, materialised after 6 rounds (2^14
    survivors x 64 shared scalars) as the bench runs it"""
    ps.check_ipa_at_size(zk, "Vesta", 20, 6, survivors=16)


def test_halo2_products_2p20(zk):
    """the same products at the circuit size of configs[2]: the scan's closing value against Python integers, batch inversion
    through a * a^-1 = 1, the IPA folds through the size-independent identity fold(fold(a, u), v) on a prefix"""
    import torch
    name, n = "PallasFp", 1 << 20
    from oracle import pyref
    p = pyref.FIELDS[name][0]
    f = ps.rand_field(name, n, 77)
    d = torch.from_numpy(f.view(np.int64)).cuda()
    out, total = zk.halo2.prefix_product(name, d, want_total=True)
    ints = orc.array_to_ints(orc.from_mont(name, f))
    acc = 1
    for v in ints:
        acc = acc * v % p
    assert orc.limbs_to_int(orc.from_mont(name, total.reshape(1, 4))[0]) == acc
    back = out.cpu().numpy().view(np.uint64)
    assert (back[0] == orc.to_mont(name, orc.ints_to_array([1], 4))[0]).all()
    pre = 1
    for i in (1, 2, 4097, n - 1):          # spot checks across block boundaries
        pre = 1
        for v in ints[:i]:
            pre = pre * v % p
        assert orc.limbs_to_int(orc.from_mont(name, back[i].reshape(1, 4))[0]) == pre
    a = torch.from_numpy(f.view(np.int64)).cuda()
    inv = a.clone()
    zk.halo2.batch_invert(name, inv)
    zk.vec_op(name, "mul", inv, a)
    torch.cuda.synchronize()
    one = orc.to_mont(name, orc.ints_to_array([1], 4))[0]
    assert (inv.cpu().numpy().view(np.uint64) == one).all()


def test_ntt_saturated_limbs_path(zk):
    """zk_ntt_opts limb_bits = 32: the saturated-word butterflies (the default is the lazy 29-bit form) against the oracle"""
    zk.ntt_configure(limb_bits=32)
    try:
        ps.check_ntt_vs_oracle(zk, "PallasFp", 20, threads=16)
        ps.check_ntt_vs_oracle(zk, "Bn254Fr", 13)
        ps.check_ntt_fused_coset(zk, "Bls381Fr", 16, threads=16)
        ps.check_ntt_extend(zk, "PallasFq", 10, 13)
        ps.check_witness_map(zk, "Bls381Fr", 12)
    finally:
        zk.ntt_configure()


def test_ntt_lazy_limbs_extremes(zk):
    """the lazy-limb tiles at the edge of their bounds: all-(p - 1) inputs (largest values on every sum path), all zeros,
    alternating 0 / p - 1, for the deepest tile (2^10 points per pass) of every field"""
    from oracle import pyref
    for name in ps.NTT_FIELDS:
        p = pyref.FIELDS[name][0]
        for logn in (10, 20):
            n = 1 << logn
            w = orc.root_of_unity(name, logn)
            top = orc.int_to_limbs(p - 1, 4)          # as stored words: the largest canonical value
            for pattern in ("max", "alt", "zero"):
                a = np.zeros((n, 4), dtype=np.uint64)
                if pattern == "max":
                    a[:] = top
                elif pattern == "alt":
                    a[::2] = top
                got = zk.halo2.best_fft(name, a, w, logn)
                assert (got == orc.halo2_best_fft(name, a, w, logn, threads=16)).all(), (name, logn, pattern)


def test_bench_work_lists_run(zk):
    """bench.py's three work-lists at a small size, one step each, as subprocesses: the JSON line carries the contract's keys
    (the full-size numbers come from the driver's own run)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in (["--logn", "12"], ["--logn", "12", "--serial"], ["--workload", "column", "--logn", "14"],
                  ["--workload", "groth16", "--curve", "Bls381G1", "--logn", "12"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline"] + extra,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                    "data", "config", "roofline"):
            assert key in line, (extra, key)
        assert line["value"] > 0 and "workload" in line["config"] and line["roofline"]["frac"] > 0
