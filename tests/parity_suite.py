"""Parity cases shared by the two tiers:
  * tests/test_gpu_parity.py  (-m gpu)      : the HIP library on a real MI355X, through the C ABI
  * tests/test_emu_parity.py  (-m "not gpu"): the same sources under tests/emu (kernel-logic check on CPU)
Every case compares against the CPU oracle (oracle/) and/or the pure-Python golden fixtures.
Bar: bit-exact (integer arithmetic) after normalising points to affine."""
import json
import os

import numpy as np

from oracle import pyref
from oracle import zk_oracle as orc

H = lambda s: int(s, 16)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NTT_FIELDS = ["PallasFp", "PallasFq", "Bn254Fr", "Bls381Fr"]
CURVES = ["Pallas", "Vesta", "Bn254G1", "Bls381G1", "Bn254G2", "Bls381G2"]


def coord_array(cname, c):
    """golden coordinate (hex string, or [c0, c1] hex pair on G2) -> Montgomery u64 limbs (c0 | c1)"""
    bf = pyref.CURVES[cname][0]
    nl = pyref.FIELDS[bf][2]
    parts = c if isinstance(c, list) else [c]
    return np.concatenate([orc.int_to_limbs(pyref.mont(bf, H(x)), nl) for x in parts])


def golden_msm_case(cname, case):
    """-> (points [n, 2L], scalars [n,4], expected affine [2L]) as uint64 arrays"""
    L = orc.coord_limbs(cname)
    n = case["n"]
    pts = np.zeros((n, 2 * L), dtype=np.uint64)
    for i, P in enumerate(case["points"]):
        if P is not None:
            pts[i, :L] = coord_array(cname, P[0])
            pts[i, L:] = coord_array(cname, P[1])
    sc = orc.ints_to_array([H(x) for x in case["scalars"]], 4)
    exp = np.zeros(2 * L, dtype=np.uint64)
    if case["result"] is not None:
        exp[:L] = coord_array(cname, case["result"][0])
        exp[L:] = coord_array(cname, case["result"][1])
    return pts, sc, exp


def rand_field(name, n, seed):
    """n uniform elements of the field, Montgomery form, uint64 [n,4] (the product's seeded generator, checked here against
    the independently typed modulus of oracle/pyref.py)"""
    from contangle_zkcp_amd import synth
    assert synth.modulus(name) == pyref.FIELDS[name][0]
    return synth.rand_field(name, n, seed)


def scalars_for(curve, n, seed, realistic=False):
    """canonical scalars [n,4] in [0, r); `realistic` = 40% zeros, 25% ones, 10% < 2^8."""
    from contangle_zkcp_amd import synth
    assert synth.CURVE_SCALAR_FIELD[curve] == pyref.CURVES[curve][1]
    return synth.scalars_for(curve, n, seed, realistic)


_bases_cache = {}


def bases_for(curve, n, seed=11):
    """P_i = [k_i]G from the oracle (cached per process)."""
    key = (curve, n, seed)
    if key not in _bases_cache:
        ks = scalars_for(curve, n, seed + 1000)
        _bases_cache[key] = orc.fixed_base_mul(curve, ks, threads=8)
    return _bases_cache[key]


# ------------------------------------------------------------------ NTT
def check_ntt_golden(zk):
    v = json.load(open(os.path.join(GOLD, "ntt_vectors.json")))["fields"]
    for name in NTT_FIELDS:
        g = orc.field_generator(name)
        assert (zk.multiplicative_generator(name) == g).all()
        for case in v[name]["cases"]:
            logn = case["logn"]
            a = orc.to_mont(name, orc.ints_to_array([H(x) for x in case["in"]], 4))
            want = lambda k: orc.to_mont(name, orc.ints_to_array([H(x) for x in case[k]], 4))
            w = zk.root_of_unity(name, logn)
            assert (w == orc.root_of_unity(name, logn)).all()
            assert (zk.halo2.best_fft(name, a, w, logn) == want("fft")).all(), (name, logn)
            dom = zk.ark.Radix2EvaluationDomain(name, 1 << logn)
            assert dom.size == 1 << logn
            assert (dom.fft_in_place(a) == want("fft")).all()
            assert (dom.ifft_in_place(a) == want("ifft")).all(), (name, logn)
            assert (dom.coset_fft_in_place(a) == want("coset_fft")).all()
            assert (dom.coset_ifft_in_place(a) == want("coset_ifft")).all()


def check_ntt_vs_oracle(zk, name, logn, seed=3, threads=8):
    a = rand_field(name, 1 << logn, seed)
    w = orc.root_of_unity(name, logn)
    got = zk.halo2.best_fft(name, a, w, logn)
    assert (got == orc.halo2_best_fft(name, a, w, logn, threads=threads)).all(), (name, logn, "best_fft")
    dom = zk.ark.Radix2EvaluationDomain(name, 1 << logn)
    assert (dom.ifft_in_place(got) == a).all(), (name, logn, "ifft(fft(x)) != x")
    if logn <= 14:
        for kind in ("ifft", "coset_fft", "coset_ifft"):
            r = getattr(dom, kind + "_in_place")(a)
            assert (r == orc.ark_fft(name, a, kind, threads=threads)).all(), (name, logn, kind)


# ------------------------------------------------------------------ MSM
def affine_of(zk, curve, jac):
    return zk.point_to_affine(curve, jac)


def check_msm_golden(zk):
    v = json.load(open(os.path.join(GOLD, "msm_vectors.json")))["curves"]
    for cname in CURVES:
        sf = pyref.CURVES[cname][1]
        for case in v[cname]["cases"]:
            n = case["n"]
            pts, sc, exp = golden_msm_case(cname, case)
            bases = zk.Bases(cname, pts)
            emu = zk.backend_info().startswith("emu")
            if emu and n > 33 and pyref.is_g2(cname):
                continue                                                    # (the emulator pays per launched lane: the GPU tier runs every case)
            for wb in ((0, 3, 7) if not emu else ((0, 6) if n <= 5 else (0,))):
                got = affine_of(zk, cname, zk.msm(bases, sc, window_bits=wb))
                assert (got == exp).all(), (cname, n, wb)
            if not emu or n <= 5:
                got = affine_of(zk, cname, zk.ark.VariableBaseMSM.multi_scalar_mul(bases, sc))
                assert (got == exp).all()
                got = affine_of(zk, cname, zk.halo2.best_multiexp(orc.to_mont(sf, sc), bases))
                assert (got == exp).all(), (cname, n, "halo2")
            bases.free()


def check_msm_vs_oracle(zk, cname, n, window_bits=0, realistic=False, seed=5, threads=8):
    sf = pyref.CURVES[cname][1]
    pts = bases_for(cname, n)
    sc = scalars_for(cname, n, seed, realistic)
    exp = orc.msm_ark(cname, pts, sc, threads=threads)
    bases = zk.Bases(cname, pts)
    got = affine_of(zk, cname, zk.msm(bases, sc, window_bits=window_bits))
    assert (got == exp).all(), (cname, n, window_bits, realistic)
    assert orc.on_curve(cname, got)
    # buckets run on lazy limbs (9 x 29 bits; 14 x 28 for BLS12-381; pairs of those on the G2 twists) ...
    assert zk.msm_last_profile()["limb_bits"] == 29
    # ... and must agree with the saturated 32-bit path (zk_msm_opts.limb_bits = 32)
    got32 = affine_of(zk, cname, zk.msm(bases, sc, window_bits=window_bits, limb_bits=32))
    assert zk.msm_last_profile()["limb_bits"] == 32
    assert (got32 == exp).all()
    # halo2 entry: Montgomery scalars
    got = affine_of(zk, cname, zk.halo2.best_multiexp(orc.to_mont(sf, sc), bases))
    assert (got == exp).all(), (cname, n, "montgomery scalars")
    bases.free()
    return exp


def check_msm_window_sharding(zk, cname, n, window_bits, parts):
    """window-range partial sums (one per 'GPU') added up = the full MSM."""
    pts = bases_for(cname, n)
    sc = scalars_for(cname, n, 9)
    bases = zk.Bases(cname, pts)
    full = zk.msm(bases, sc, window_bits=window_bits)
    W = zk.msm_window_count(cname, n, window_bits)
    acc = None
    for g in range(parts):
        lo, hi = W * g // parts, W * (g + 1) // parts
        if lo == hi:
            continue
        part = zk.msm(bases, sc, window_bits=window_bits, windows=(lo, hi))
        acc = part if acc is None else zk.point_add(cname, acc, part)
    assert (affine_of(zk, cname, acc) == affine_of(zk, cname, full)).all()
    assert (affine_of(zk, cname, full) == orc.msm_ark(cname, pts, sc, threads=8)).all()
    bases.free()


def check_msm_edges(zk, cname):
    nl = orc.coord_limbs(cname)
    g = orc.curve_generator(cname)
    r = pyref.FIELDS[pyref.CURVES[cname][1]][0]
    # empty input -> identity (Z = 0)
    bases = zk.Bases(cname, np.zeros((0, 2 * nl), dtype=np.uint64))
    out = zk.msm(bases, np.zeros((0, 4), dtype=np.uint64))
    assert not out[2 * nl:].any() and not affine_of(zk, cname, out).any()
    bases.free()
    # all-zero scalars, all-identity bases
    n = 40
    bases = zk.Bases(cname, np.tile(g, (n, 1)))
    assert not affine_of(zk, cname, zk.msm(bases, np.zeros((n, 4), dtype=np.uint64))).any()
    # every point equal (forces the doubling branch of the mixed add), scalars 1..n
    sc = np.zeros((n, 4), dtype=np.uint64)
    sc[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    exp = orc.scalar_mul(cname, g, orc.int_to_limbs(n * (n + 1) // 2, 4))
    assert (affine_of(zk, cname, zk.msm(bases, sc, window_bits=4)) == exp).all()
    # maximal scalars r-1 on all: sum = -n G
    sc = np.tile(orc.int_to_limbs(r - 1, 4), (n, 1))
    exp = orc.scalar_mul(cname, g, orc.int_to_limbs((r - n) % r, 4))
    for wb in (0, 5, 13):
        assert (affine_of(zk, cname, zk.msm(bases, sc, window_bits=wb)) == exp).all(), wb
    # ragged: fewer scalars than bases uses the common prefix (ark: min(len, len))
    out = zk.ark.VariableBaseMSM.multi_scalar_mul(bases, sc[:7])
    assert (affine_of(zk, cname, out) == orc.scalar_mul(cname, g, orc.int_to_limbs((7 * (r - 1)) % r, 4))).all()
    bases.free()
    # identity bases only
    bases = zk.Bases(cname, np.zeros((n, 2 * nl), dtype=np.uint64))
    assert not affine_of(zk, cname, zk.msm(bases, sc)).any()
    bases.free()


def check_msm_big_buckets(zk, cname, n=5200, window_bits=6):
    """buckets far above the oversize threshold (2x mean + 64) and above one segment (2048): 45% of the scalars are 1, 45% are 3 and the rest random --
    exercises the cooperative segment kernels."""
    pts = bases_for(cname, n)
    sc = scalars_for(cname, n, 77)
    sel = np.arange(n) % 20
    sc[sel < 9] = 0
    sc[sel < 9, 0] = 1
    sc[(sel >= 9) & (sel < 18)] = 0
    sc[(sel >= 9) & (sel < 18), 0] = 3
    bases = zk.Bases(cname, pts)
    got = affine_of(zk, cname, zk.msm(bases, sc, window_bits=window_bits))
    assert (got == orc.msm_ark(cname, pts, sc, threads=8)).all(), (cname, n)
    bases.free()


# ------------------------------------------------------------------ Groth16 witness map
def to_device(zk, arr):
    """device buffer for the *_device entry points: a torch tensor on the GPU, or (under the CPU test emulator,
    whose 'device memory' is host memory) a numpy copy"""
    if zk.backend_info().startswith("emu"):
        return np.ascontiguousarray(arr, dtype=np.uint64).copy()
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint64).view(np.int64)).cuda()


def to_host(zk, buf):
    if isinstance(buf, np.ndarray):
        return buf
    import torch
    torch.cuda.synchronize()
    return buf.cpu().numpy().view(np.uint64)


def check_vec_ops(zk, name, n=1000):
    a, b, c = rand_field(name, n, 1), rand_field(name, n, 2), rand_field(name, n, 3)
    s = rand_field(name, 1, 4)[0]
    fe = lambda op, x, y=None: np.stack([orc.fe_op(name, op, x[i], None if y is None else y[i]) for i in range(len(x))])
    assert (to_host(zk, zk.vec_op(name, "mul", to_device(zk, a), to_device(zk, b))) == fe("mul", a, b)).all()
    assert (to_host(zk, zk.vec_op(name, "sub", to_device(zk, a), to_device(zk, b))) == fe("sub", a, b)).all()
    assert (to_host(zk, zk.vec_op(name, "add", to_device(zk, a), to_device(zk, b))) == fe("add", a, b)).all()
    assert (to_host(zk, zk.vec_op(name, "scale", to_device(zk, a), scalar=s)) == fe("mul", a, np.tile(s, (n, 1)))).all()
    assert (to_host(zk, zk.vec_op(name, "into_repr", to_device(zk, a))) == orc.from_mont(name, a)).all()
    assert (to_host(zk, zk.vec_op(name, "from_repr", to_device(zk, orc.from_mont(name, a)))) == a).all()
    q = to_host(zk, zk.vec_op(name, "qap", to_device(zk, a), to_device(zk, b), to_device(zk, c), scalar=s))
    assert (q == fe("mul", fe("sub", fe("mul", a, b), c), np.tile(s, (n, 1)))).all()


def check_witness_map(zk, name, logm, threads=8):
    """a, b random evaluations; c = a.b on the first m-2 rows and random on the rest is NOT a valid witness, so
    use c = a.b pointwise (valid QAP): h must have degree <= m-2 and match the oracle's restatement bit for bit."""
    m = 1 << logm
    a, b = rand_field(name, m, 11), rand_field(name, m, 12)
    c = to_host(zk, zk.vec_op(name, "mul", to_device(zk, a), to_device(zk, b))).copy()
    exp = orc.groth16_witness_map(name, a, b, c, threads=threads)
    h = to_host(zk, zk.groth16_witness_map(name, to_device(zk, a), to_device(zk, b), to_device(zk, c)))
    assert (h == exp).all(), (name, logm)
    assert not h[m - 1].any()                    # deg h <= m - 2 for a satisfied system
    assert h[: m - 1].any()
    # an unsatisfied system (random c) still matches the oracle (no structural shortcut taken)
    c2 = rand_field(name, m, 13)
    exp2 = orc.groth16_witness_map(name, a, b, c2, threads=threads)
    h2 = to_host(zk, zk.groth16_witness_map(name, to_device(zk, a), to_device(zk, b), to_device(zk, c2)))
    assert (h2 == exp2).all()


def check_msm_sort_shapes(zk, cname, n, window_bits_list):
    """the staged sort (count / stage by bucket range / LDS counting sort per region) for several window widths --
    one range per window, several ranges, several scalar blocks -- on a skewed witness, whole and window-sharded"""
    pts = bases_for(cname, n)
    sc = scalars_for(cname, n, 41, realistic=True)
    exp = orc.msm_ark(cname, pts, sc, threads=8)
    bases = zk.Bases(cname, pts)
    for window_bits in window_bits_list:
        W = zk.msm_window_count(cname, n, window_bits)
        got = affine_of(zk, cname, zk.msm(bases, sc, window_bits=window_bits))
        assert (got == exp).all(), (cname, window_bits)
        assert zk.msm_last_profile()["groups"] == 1
        lo = zk.msm(bases, sc, window_bits=window_bits, windows=(0, W // 3))
        hi = zk.msm(bases, sc, window_bits=window_bits, windows=(W // 3, W))
        assert (affine_of(zk, cname, zk.point_add(cname, lo, hi)) == exp).all(), (cname, window_bits, "sharded")
    bases.free()

def check_msm_window_groups(zk, cname, n, window_bits, groups=(1, 3, 5, 100)):
    """zk_msm_opts.window_group: the windows of one job processed a few at a time (what a >= 2^21-point MSM does by itself so that
    sorted entries + bases stay inside the Infinity Cache) -- every group size gives the same point, whole and as a window share"""
    pts = bases_for(cname, n)
    sc = scalars_for(cname, n, 47, realistic=True)
    exp = orc.msm_ark(cname, pts, sc, threads=8)
    bases = zk.Bases(cname, pts)
    W = zk.msm_window_count(cname, n, window_bits)
    for gw in groups:
        got = affine_of(zk, cname, zk.msm(bases, sc, window_bits=window_bits, window_group=gw))
        assert (got == exp).all(), (cname, gw)
        assert zk.msm_last_profile()["groups"] == -(-W // min(gw, W)), (cname, gw, zk.msm_last_profile()["groups"])
        lo = zk.msm(bases, sc, window_bits=window_bits, windows=(0, W // 2), window_group=gw)
        hi = zk.msm(bases, sc, window_bits=window_bits, windows=(W // 2, W), window_group=gw)
        assert (affine_of(zk, cname, zk.point_add(cname, lo, hi)) == exp).all(), (cname, gw, "shares")
    t = zk.msm_submit(bases, to_device(zk, sc), window_bits=window_bits, window_group=2)
    assert (affine_of(zk, cname, t.collect()) == exp).all()
    bases.free()


def check_msm_slice_lengths(zk, cname, n, window_bits):
    """the bucket reduction for several slice lengths L (X_t = W_t + [t L] S_t with the 2-bit windowed multiplier):
    powers of two (the digits of t, then log2 L doublings), a non-power of two (digits of t L), L = all buckets"""
    pts = bases_for(cname, n)
    sc = scalars_for(cname, n, 23)
    exp = orc.msm_ark(cname, pts, sc, threads=8)
    bases = zk.Bases(cname, pts)
    for L in (1, 2, 3, 8, 64, 1024):
        got = affine_of(zk, cname, zk.msm(bases, sc, window_bits=window_bits, slice_len=L, slice_reduce=True))
        assert (got == exp).all(), (cname, L)
    bases.free()


def check_msm_precomputed(zk, cname, n, window_bits, seed=37, realistic=False, count=0):
    """ZK_MSM_FLAG_PRECOMPUTED: one bucket set over the table of window multiples [2^(c w)] P_i (zk_bases_precompute) --
    against the default form and the oracle; single MSMs and a batch"""
    sf = pyref.CURVES[cname][1]
    pts = bases_for(cname, n)
    sc = scalars_for(cname, n, seed, realistic)
    exp = orc.msm_ark(cname, pts, sc, threads=8)
    bases = zk.Bases(cname, pts)
    bases.precompute(window_bits)
    got = affine_of(zk, cname, zk.msm(bases, to_device(zk, sc), window_bits=window_bits, precomputed=True))
    assert (got == exp).all(), (cname, n, window_bits)
    got = affine_of(zk, cname, zk.msm(bases, to_device(zk, orc.to_mont(sf, sc)), montgomery=True, window_bits=window_bits, precomputed=True))
    assert (got == exp).all(), (cname, n, window_bits, "montgomery")
    if count:
        cols = np.stack([scalars_for(cname, n, seed + 1 + i, realistic=(i % 2 == 1)) for i in range(count)])
        got = zk.msm_batch(bases, to_device(zk, cols), window_bits=window_bits, precomputed=True)
        for i in range(count):
            assert (affine_of(zk, cname, got[i]) == orc.msm_ark(cname, pts, cols[i], threads=8)).all(), (cname, n, i, "batch")
    bases.free()


def check_msm_axis_reduce(zk, cname, n, window_bits_list, windows=None, seed=29):
    """the default bucket reduction (row / column sums, suffix scan + tree) over the shapes of its matrix: one row
    (c = 2, 3), rows = columns (odd c), rows = columns / 2 (even c), more columns than one workgroup holds (c = 16 on the
    G2 point types: two blocks per axis, combined on the host); against the slice form and the oracle"""
    pts = bases_for(cname, n)
    sc = scalars_for(cname, n, seed)
    exp = orc.msm_ark(cname, pts, sc, threads=8) if windows is None else None
    bases = zk.Bases(cname, pts)
    for wb in window_bits_list:
        got = zk.msm(bases, sc, window_bits=wb, windows=windows)
        ref = zk.msm(bases, sc, window_bits=wb, windows=windows, slice_reduce=True)
        assert (affine_of(zk, cname, got) == affine_of(zk, cname, ref)).all(), (cname, wb)
        if exp is not None:
            assert (affine_of(zk, cname, got) == exp).all(), (cname, wb)
    bases.free()


def check_msm_device_partials(zk, cname, n=700, window_bits_list=(0, 3, 7, 13)):
    """ZK_MSM_FLAG_DEVICE_PARTIALS: the per-window partial sums of the row / column reduction converted to the caller's limb form
    on the device (one lane, after the LDS scan) must be the host's conversion of the same lazy values, stage by stage -- the
    conversion every lazy-limb result passes through.  (Round 2 saw BN254-only wrong points from exactly this on the GPU.)"""
    pts = bases_for(cname, n)
    sc = scalars_for(cname, n, 55, realistic=True)
    exp = orc.msm_ark(cname, pts, sc, threads=8)
    bases = zk.Bases(cname, pts)
    for wb in window_bits_list:
        got = affine_of(zk, cname, zk.msm(bases, sc, window_bits=wb, device_partials=True))
        prof = zk.msm_last_profile()
        assert prof["reserved"] == 0, (cname, wb, "device / host conversions differ: count %d stage %d component %d"
                                       % (prof["reserved"] & 0xffffff, (prof["reserved"] >> 24) & 15, (prof["reserved"] >> 28) & 15))
        assert (got == exp).all(), (cname, wb)
    # the golden two-point case that failed in round 2
    v = json.load(open(os.path.join(GOLD, "msm_vectors.json")))["curves"][cname]["cases"]
    for case in v:
        gp, gs, gexp = golden_msm_case(cname, case)
        gb = zk.Bases(cname, gp)
        got = affine_of(zk, cname, zk.msm(gb, gs, device_partials=True))
        assert zk.msm_last_profile()["reserved"] == 0 and (got == gexp).all(), (cname, case["n"], "golden")
        gb.free()
    bases.free()


def check_fixed_base_msm(zk, cname, n, seed=91):
    """zk_fixed_base_msm_device (8-bit window table + batched normalisation; ark-ec FixedBaseMSM + batch_normalization)
    against the oracle's plain double-and-add: generator and an arbitrary base, canonical and Montgomery scalars,
    edge scalars 0 / 1 / r-1 / 2^k, a length that is not a multiple of the normalisation batch"""
    sf = pyref.CURVES[cname][1]
    r = pyref.FIELDS[sf][0]
    ks = scalars_for(cname, n, seed)
    edge = [0, 1, r - 1, 1 << 8, (1 << 8) - 1, 1 << 248, 255 << 16]
    ks[:len(edge)] = orc.ints_to_array(edge, 4)
    nl = zk.base_limbs(cname)
    exp = orc.fixed_base_mul(cname, ks, threads=8)
    out = to_device(zk, np.zeros((n, 2 * nl), dtype=np.uint64))
    zk.ark.FixedBaseMSM.multi_scalar_mul(cname, None, to_device(zk, ks), out)
    got = to_host(zk, out)
    assert (got == exp).all(), (cname, "generator")
    assert (got[0] == 0).all()                                   # [0] G = identity = (0, 0)
    # an arbitrary base B = [7919] G, Montgomery-form scalars (what a Rust caller holding Fr values would pass)
    seven = orc.fixed_base_mul(cname, orc.ints_to_array([7919], 4))[0]
    m = min(n, 96)
    exp_b = np.stack([orc.scalar_mul(cname, seven, ks[i]) for i in range(m)])
    out = to_device(zk, np.zeros((m, 2 * nl), dtype=np.uint64))
    zk.ark.FixedBaseMSM.multi_scalar_mul(cname, seven, to_device(zk, orc.to_mont(sf, ks[:m])), out, montgomery=True)
    assert (to_host(zk, out) == exp_b).all(), (cname, "base 7919 G")


def check_ntt_fused_coset(zk, name, logn, threads=8):
    """zk_ntt_coset_device: the coset shifts fused into the first / last NTT pass (on-the-fly powers from two small
    tables) against the oracle's ark-poly coset_fft / coset_ifft restatement."""
    a = rand_field(name, 1 << logn, 17)
    w = orc.root_of_unity(name, logn)
    g = orc.field_generator(name)
    winv, ginv = orc.fe_op(name, "inv", w), orc.fe_op(name, "inv", g)
    d = to_device(zk, a)
    got = to_host(zk, zk.ntt(name, d, w, coset_pre=g, device=True))
    assert (got == orc.ark_fft(name, a, "coset_fft", threads=threads)).all(), (name, logn, "coset_fft")
    d = to_device(zk, a)
    got = to_host(zk, zk.ntt(name, d, winv, scale_by_n_inv=True, coset_post=ginv, device=True))
    assert (got == orc.ark_fft(name, a, "coset_ifft", threads=threads)).all(), (name, logn, "coset_ifft")
    # both at once: x -> g^-k/n DFT^-1 ( g^i x )  has no upstream name, check it against the two oracle steps
    d = to_device(zk, a)
    got = to_host(zk, zk.ntt(name, d, w, coset_pre=g, coset_post=ginv, device=True))
    exp = orc.distribute_powers(name, orc.ark_fft(name, a, "coset_fft", threads=threads), ginv)
    assert (got == exp).all()
    # standalone coset_mul on a device buffer
    d = to_device(zk, a)
    assert (to_host(zk, zk.coset_mul(name, d, g)) == orc.distribute_powers(name, a, g)).all()


def check_ntt_extend(zk, name, log_in, logn, threads=8):
    """zk_ntt_extend_device: the transform of a zero-extended input (halo2 coeff_to_extended) never reads the padding --
    the tail of the buffer is filled with garbage here -- and equals the oracle's coset FFT of the padded vector"""
    n = 1 << logn
    a = np.zeros((n, 4), dtype=np.uint64)
    a[:1 << log_in] = rand_field(name, 1 << log_in, 29)
    w = orc.root_of_unity(name, logn)
    g = orc.field_generator(name)
    exp = orc.ark_fft(name, a, "coset_fft", threads=threads)
    dirty = a.copy()
    dirty[1 << log_in:] = rand_field(name, n - (1 << log_in), 31) if log_in < logn else dirty[1 << log_in:]
    got = to_host(zk, zk.halo2.coeff_to_extended(name, to_device(zk, dirty), log_in, w, g))
    assert (got == exp).all(), (name, log_in, logn)
    # without the coset shift, with 1/n scaling
    winv = orc.fe_op(name, "inv", w)
    got = to_host(zk, zk.ntt(name, to_device(zk, dirty), winv, scale_by_n_inv=True, in_log=log_in, device=True))
    assert (got == orc.ark_fft(name, a, "ifft", threads=threads)).all(), (name, log_in, logn, "ifft")


def check_msm_split(zk, cname, n, window_bits, realistic=True):
    """bucket splitting (every bucket's slice summed by 2^k lanes, then combined) for k = 0..3, including oversized
    buckets (realistic mix) and buckets shorter than the number of pieces"""
    pts = bases_for(cname, n)
    sc = scalars_for(cname, n, 43, realistic=realistic)
    exp = orc.msm_ark(cname, pts, sc, threads=8)
    bases = zk.Bases(cname, pts)
    for k in (0, 1, 2, 3):
        got = affine_of(zk, cname, zk.msm(bases, sc, window_bits=window_bits, split_log=k))
        assert (got == exp).all(), (cname, n, k)
    bases.free()


# ------------------------------------------------------------------ deferred results, batches, several devices
def check_msm_async(zk, cname, n, window_bits=0):
    """zk_msm_submit / zk_msm_collect: three MSMs in flight over the same bases, collected out of order; a fifth
    submission without a collect is refused with ZK_ERR_BUSY, not queued"""
    sf = pyref.CURVES[cname][1]
    pts = bases_for(cname, n)
    bases = zk.Bases(cname, pts)
    scs = [scalars_for(cname, n, 300 + i, realistic=(i == 1)) for i in range(3)]
    exp = [orc.msm_ark(cname, pts, s, threads=8) for s in scs]
    d = [to_device(zk, s) for s in scs[:2]] + [to_device(zk, orc.to_mont(sf, scs[2]))]
    tk = [zk.msm_submit(bases, d[0], window_bits=window_bits), zk.msm_submit(bases, d[1], window_bits=window_bits),
          zk.msm_submit(bases, d[2], montgomery=True, window_bits=window_bits)]
    for i in (2, 0, 1):
        assert (affine_of(zk, cname, tk[i].collect()) == exp[i]).all(), (cname, n, i)
    tk = [zk.msm_submit(bases, d[0]) for _ in range(4)]
    try:
        zk.msm_submit(bases, d[0])
        raise AssertionError("a fifth MSM in flight must be refused")
    except zk.ZkError as e:
        assert e.status == -8
    for t in tk:
        assert (affine_of(zk, cname, t.collect()) == exp[0]).all()
    try:
        tk[0].collect()
        raise AssertionError("a ticket can be collected once")
    except zk.ZkError as e:
        assert e.status == -7
    bases.free()


def check_msm_batch(zk, cname, n, count, window_bits=0):
    """zk_msm_batch_device: `count` scalar vectors against one bases handle (halo2's column commitments)"""
    sf = pyref.CURVES[cname][1]
    pts = bases_for(cname, n)
    bases = zk.Bases(cname, pts)
    cols = np.stack([scalars_for(cname, n, 500 + i, realistic=(i % 2 == 1)) for i in range(count)])
    exp = [orc.msm_ark(cname, pts, cols[i], threads=8) for i in range(count)]
    got = zk.msm_batch(bases, to_device(zk, cols), window_bits=window_bits)
    for i in range(count):
        assert (affine_of(zk, cname, got[i]) == exp[i]).all(), (cname, n, i)
    mont = np.stack([orc.to_mont(sf, cols[i]) for i in range(count)])
    got = zk.msm_batch(bases, to_device(zk, mont), montgomery=True, window_bits=window_bits)
    for i in range(count):
        assert (affine_of(zk, cname, got[i]) == exp[i]).all(), (cname, n, i, "montgomery")
    # vectors further apart than n (stride_elems > n), straight through the C entry
    import ctypes
    pad = 5
    padded = np.zeros((count, n + pad, 4), dtype=np.uint64)
    padded[:, :n] = cols
    padded[:, n:] = 0xFFFFFFFFFFFFFFFF           # garbage between the vectors must not be read
    d_pad = to_device(zk, padded)
    nl = zk.base_limbs(cname)
    out = np.zeros((count, 3 * nl), dtype=np.uint64)
    st = zk.load().zk_msm_batch_device(zk.curve_id(cname), bases.handle, zk._ptr(d_pad), n, count, n + pad, 0,
                                       ctypes.byref(zk.msm_opts(window_bits)), zk._ptr(out), ctypes.c_void_p(0))
    assert st == 0
    for i in range(count):
        assert (affine_of(zk, cname, out[i]) == exp[i]).all(), (cname, n, i, "stride")
    # a window share of every vector (the sharded form): up to 4 vectors ride in one job, each keeps its own windows
    W = zk.msm_window_count(cname, n, window_bits)
    if W >= 2:
        lo = zk.msm_batch(bases, to_device(zk, cols), window_bits=window_bits, windows=(0, W // 2))
        hi = zk.msm_batch(bases, to_device(zk, cols), window_bits=window_bits, windows=(W // 2, W))
        for i in range(count):
            assert (affine_of(zk, cname, zk.point_add(cname, lo[i], hi[i])) == exp[i]).all(), (cname, n, i, "window shares")
    bases.free()


def check_multi_device(zk, ndev):
    """one process, several devices (zk_init_devices): every whole MSM is split over the devices by scalar window and
    the partial sums are added on the host; host-pointer and device-pointer entries, more devices than windows, an
    explicit window range (stays on the home device), batches and deferred results on a multi-device process"""
    assert zk.device_count() == ndev
    for ci, (cname, n, wb) in enumerate((("Vesta", 500, 0), ("Bls381G1", 200, 7), ("Bn254G2", 60, 5), ("Pallas", 50, 13))):
        pts = bases_for(cname, n)
        sc = scalars_for(cname, n, 71, realistic=True)
        exp = orc.msm_ark(cname, pts, sc, threads=8)
        bases = zk.Bases(cname, pts)
        assert (affine_of(zk, cname, zk.msm(bases, sc, window_bits=wb)) == exp).all(), (cname, "host scalars")
        assert (affine_of(zk, cname, zk.msm(bases, to_device(zk, sc), window_bits=wb)) == exp).all(), (cname, "device scalars")
        if ci in (1, 2) and ndev == 2:
            bases.free()
            continue                                   # (CPU-emulator time: the 3-device case runs every form on every curve)
        W = zk.msm_window_count(cname, n, wb)
        lo = zk.msm(bases, sc, window_bits=wb, windows=(0, W // 2))
        hi = zk.msm(bases, sc, window_bits=wb, windows=(W // 2, W))
        assert (affine_of(zk, cname, zk.point_add(cname, lo, hi)) == exp).all()
        assert (affine_of(zk, cname, zk.msm_submit(bases, to_device(zk, sc), window_bits=wb).collect()) == exp).all()
        bases.free()
        adopted = zk.Bases(cname, device_tensor=to_device(zk, pts), n=n)      # resident on one device, copied to its peers
        assert (affine_of(zk, cname, zk.msm(adopted, sc, window_bits=wb)) == exp).all(), (cname, "adopted bases")
        adopted.free()
    check_msm_batch(zk, "Vesta", 200, 3)          # >= one vector per device: whole MSMs per device, only those vectors travel
    if ndev > 2:
        check_msm_batch(zk, "Pallas", 120, 7, 6)  # an uneven split of the vectors
        check_msm_batch(zk, "Vesta", 130, 2, 6)   # fewer vectors than devices: each one window-sharded over all devices
        check_ipa(zk, "Vesta", 3)                 # zk_ipa_round_device over the devices (the L / R pair), collapse, literal folds
    else:
        check_msm_async(zk, "Vesta", 150, 6)      # fanned-out tickets: a job per device behind each, collected out of order
        check_msm_edges(zk, "Bn254G1")
    check_ntt_vs_oracle(zk, "PallasFp", 9)


def check_quotient_by_parts(zk, name, cname, k, parts, seed=77, direct_pieces=None):
    """the quotient through its sub-cosets: part_to_coeff of every sub-coset's values + the mixing scalars reproduce upstream's
    extended_to_coeff (every coefficient, through Python integers), the folded h(X) = sum_q x^(n q) h_q, and -- commitments being
    linear -- the pieces' commitments as combinations of the sub-cosets' (against the MSM of the piece itself and the oracle)"""
    p = pyref.FIELDS[name][0]
    dom = zk.halo2.EvaluationDomain(name, 9, k)
    n, ne = 1 << k, dom.extended_len()
    m, r = ne // parts, (ne // parts) // n
    unmont = lambda a: orc.limbs_to_int(orc.from_mont(name, a.reshape(1, 4))[0])
    h_ext = rand_field(name, ne, seed)
    coeffs = to_host(zk, dom.extended_to_coeff(to_device(zk, h_ext.copy())))                     # upstream's path
    A = []
    for j in range(parts):
        A.append(to_host(zk, dom.part_to_coeff(to_device(zk, np.ascontiguousarray(h_ext[j::parts])), j, parts)))
    c = dom.part_mix(parts)
    Ai = [[unmont(a[t]) for t in range(m)] for a in A]
    ci = [unmont(coeffs[t]) for t in range(ne)]
    for i in range(parts):
        for t in range(m):
            assert sum(c[i][j] * Ai[j][t] for j in range(parts)) % p == ci[i * m + t], (name, k, parts, i, t)
    # the folded quotient
    xn = 0x1234567 * 0x89abcdef % p
    e = dom.fold_scalars(parts, xn)
    for t in (0, 1, n // 2, n - 1):
        exp = sum(pow(xn, q, p) * ci[q * n + t] for q in range(ne // n)) % p
        got = sum(e[j][s_] * Ai[j][s_ * n + t] for j in range(parts) for s_ in range(r)) % p
        assert got == exp, (name, k, parts, "fold", t)
    # the pieces' commitments
    pts = bases_for(cname, n)
    bases = zk.Bases(cname, pts)
    C, index = [], {}
    for j in range(parts):
        for s_ in range(r):
            index[(j, s_)] = len(C)
            C.append(zk.msm(bases, to_device(zk, np.ascontiguousarray(A[j][s_ * n:(s_ + 1) * n])), montgomery=True))
    rows = dom.piece_scalars(parts)
    got = zk.halo2.combine_commitments(cname, C, [[(index[(j, s_)], sc) for j, s_, sc in terms] for _, terms in rows])
    for (q, _), gq in list(zip(rows, got))[:direct_pieces]:
        direct = zk.msm(bases, to_device(zk, np.ascontiguousarray(coeffs[q * n:(q + 1) * n])), montgomery=True)
        assert (affine_of(zk, cname, gq) == affine_of(zk, cname, direct)).all(), (name, k, parts, "piece", q)
    q0 = rows[0][0]
    assert (affine_of(zk, cname, got[0]) == orc.msm_ark(cname, pts, orc.from_mont(name, coeffs[q0 * n:(q0 + 1) * n]), threads=4)).all()
    bases.free()


# ------------------------------------------------------------------ halo2 EvaluationDomain (poly/domain.rs)
PASTA_ZETA = {   # pasta_curves 0.4 FieldExt::ZETA (5^((p-1)/3))
    "PallasFp": 0x2d33357cb532458ed3552a23a8554e5005270d29d19fc7d27b7fd22f0201b547,
    "PallasFq": 0x06819a58283e528e511db4d81cf70f5a0fed467d47c033af2aa9d2e050aa0e4f,
}


def _fe_mul_rows(name, a, consts):
    """a[i] * consts[i mod len(consts)] elementwise through the oracle's field multiplication"""
    out = a.copy()
    m = len(consts)
    for i in range(a.shape[0]):
        if consts[i % m] is not None:
            out[i] = orc.fe_op(name, "mul", a[i], consts[i % m])
    return out


def check_halo2_domain(zk, name, k, j=9):
    """zk.halo2.EvaluationDomain against a restatement of halo2_proofs 0.2 poly/domain.rs built from the oracle's best_fft and
    field multiplication: constants (omega, extended_k, ZETA, t_evaluations), lagrange_to_coeff, coeff_to_extended (zero
    extension + distribute_powers_zeta: a[i] *= ZETA^(i mod 3)), extended_to_coeff, divide_by_vanishing_poly"""
    p = pyref.FIELDS[name][0]
    dom = zk.halo2.EvaluationDomain(name, j, k)
    n, ek = 1 << k, dom.extended_k
    ne = 1 << ek
    assert ne >= n * (j - 1) and (ne >> 1) < n * (j - 1)
    mont = lambda v: orc.int_to_limbs(pyref.mont(name, v % p), 4)
    unmont = lambda a: orc.limbs_to_int(orc.from_mont(name, a.reshape(1, 4))[0])
    zeta = unmont(dom.g_coset)
    assert zeta != 1 and pow(zeta, 3, p) == 1
    if name in PASTA_ZETA:
        assert zeta == PASTA_ZETA[name]
    assert (dom.omega == orc.root_of_unity(name, k)).all() and (dom.extended_omega == orc.root_of_unity(name, ek)).all()
    w_ext = unmont(dom.extended_omega)
    t_exp = [pow((pow(zeta * pow(w_ext, i, p) % p, n, p) - 1) % p, -1, p) for i in range(1 << (ek - k))]
    assert all(unmont(dom.t_evaluations[i]) == t_exp[i] for i in range(len(t_exp)))
    a = rand_field(name, n, 1234)
    # lagrange_to_coeff: best_fft(omega^-1) then times n^-1
    w_inv = orc.fe_op(name, "inv", dom.omega)
    exp = _fe_mul_rows(name, orc.halo2_best_fft(name, a, w_inv, k, threads=4), [mont(pow(n, -1, p))])
    got = to_host(zk, dom.lagrange_to_coeff(to_device(zk, a)))
    assert (got == exp).all(), (name, k, "lagrange_to_coeff")
    assert (to_host(zk, dom.coeff_to_lagrange(to_device(zk, got))) == a).all()
    # out of place (zk_ntt_oop_device): the Lagrange values stay, the coefficients land in `out`
    d_lag, d_out = to_device(zk, a), to_device(zk, np.full((n, 4), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))
    dom.lagrange_to_coeff(d_lag, out=d_out)
    assert (to_host(zk, d_out) == exp).all() and (to_host(zk, d_lag) == a).all(), (name, k, "lagrange_to_coeff out of place")
    # coeff_to_extended
    coeffs = got
    padded = np.zeros((ne, 4), dtype=np.uint64)
    padded[:n] = coeffs
    zp = [None, mont(zeta), mont(zeta * zeta)]
    exp_ext = orc.halo2_best_fft(name, _fe_mul_rows(name, padded, zp), dom.extended_omega, ek, threads=4)
    dirty = padded.copy()
    if ne > n:
        dirty[n:] = rand_field(name, ne - n, 99)        # the padding is implied, never read
    got_ext = to_host(zk, dom.coeff_to_extended(to_device(zk, dirty)))
    assert (got_ext == exp_ext).all(), (name, k, "coeff_to_extended")
    d_co, d_ex = to_device(zk, coeffs), to_device(zk, np.full((ne, 4), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))
    dom.coeff_to_extended(d_ex, coeffs=d_co)
    assert (to_host(zk, d_ex) == exp_ext).all() and (to_host(zk, d_co) == coeffs).all(), (name, k, "coeff_to_extended out of place")
    # the same coset in sub-cosets (the sharded quotient): part j of `parts` = the extended evaluations i * parts + j, as one
    # transform of size extended_len / parts each; with the periodic division applied per part
    tt_all = [mont(v) for v in t_exp]
    exp_div_all = _fe_mul_rows(name, exp_ext, tt_all)
    parts = 1
    while parts <= (1 << (ek - k)):
        for part in range(parts):
            d_co, d_pt = to_device(zk, coeffs), to_device(zk, np.full((ne // parts, 4), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))
            dom.coeff_to_extended_part(d_co, d_pt, part, parts, lazy_out=True)      # ZK_NTT_OUT_R29: the same values times 2^5
            lz = to_host(zk, d_pt).copy()
            dom.coeff_to_extended_part(d_co, d_pt, part, parts)
            assert (to_host(zk, d_pt) == exp_ext[part::parts]).all(), (name, k, part, parts, "coeff_to_extended_part")
            assert (lz == to_host(zk, zk.halo2.to_lazy_form(name, to_device(zk, exp_ext[part::parts])))).all(), (name, k, part, parts, "lazy_out")
            dom.divide_by_vanishing_poly_part(d_pt, part, parts)
            assert (to_host(zk, d_pt) == exp_div_all[part::parts]).all(), (name, k, part, parts, "divide_by_vanishing_poly_part")
        assert dom.rot_scale_part(parts) * parts == 1 << (ek - k)
        # ... and all of them from ONE transform of the whole coset that stores sub-coset by sub-coset (ZK_NTT_OUT_SUBCOSETS), out of
        # place from the coefficients and in place from the zero-padded buffer
        d_co, d_all = to_device(zk, coeffs), to_device(zk, np.full((ne, 4), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))
        dom.coeff_to_extended(d_all, coeffs=d_co, parts=parts)
        got_all = to_host(zk, d_all).reshape(parts, ne // parts, 4)
        for part in range(parts):
            assert (got_all[part] == exp_ext[part::parts]).all(), (name, k, part, parts, "coeff_to_extended(parts=)")
        got_all = to_host(zk, dom.coeff_to_extended(to_device(zk, dirty), parts=parts)).reshape(parts, ne // parts, 4)
        for part in range(parts):
            assert (got_all[part] == exp_ext[part::parts]).all(), (name, k, part, parts, "coeff_to_extended(parts=) in place")
        parts *= 2
    # divide_by_vanishing_poly
    tt = [mont(v) for v in t_exp]
    exp_div = _fe_mul_rows(name, exp_ext, tt)
    got_div = to_host(zk, dom.divide_by_vanishing_poly(to_device(zk, exp_ext)))
    assert (got_div == exp_div).all(), (name, k, "divide_by_vanishing_poly")
    # extended_to_coeff: best_fft(extended_omega^-1), times 2^-extended_k, ZETA^-(i mod 3)
    we_inv = orc.fe_op(name, "inv", dom.extended_omega)
    back = _fe_mul_rows(name, orc.halo2_best_fft(name, exp_ext, we_inv, ek, threads=4), [mont(pow(ne, -1, p))])
    back = _fe_mul_rows(name, back, [None, mont(zeta * zeta), mont(zeta)])
    got_back = to_host(zk, dom.extended_to_coeff(to_device(zk, exp_ext)))
    assert (got_back == back).all(), (name, k, "extended_to_coeff")
    assert (got_back == padded).all()


# ------------------------------------------------------------------ Groth16 end to end (a1 / a6): toy R1CS, known trapdoor
def check_groth16_prove(zk, pairing, seed=5, num_constraints=40, long_rows=(17,), num_inputs=3):
    """The whole device-side create_proof -- CSR mat-vecs, witness map, five MSMs over a key read from the reference's
    serialize_unchecked bytes, assembly, ark_to_bytes(proof) -- against the pure-Python Groth16-in-the-exponent reference
    (oracle/pyref_groth16.py): A, B, C must equal [a]G1, [b]G2, [c]G1 and satisfy the verification identity."""
    from oracle import pyref_groth16 as g16
    az = zk.ark_serialize
    field = "Bls381Fr" if pairing == "Bls381" else "Bn254Fr"
    g1, g2 = az.PAIRING_CURVES[az.pairing_id(pairing)]
    p = pyref.FIELDS[field][0]
    r1cs, z = g16.random_r1cs(field, seed, num_inputs=num_inputs, num_constraints=num_constraints, long_rows=long_rows)
    key = g16.setup(r1cs, seed + 100)
    m, nvars = key["m"], len(z)
    pts = lambda curve, logs: orc.fixed_base_mul(curve, orc.ints_to_array([v % p for v in logs], 4), threads=8)
    one = lambda curve, v: pts(curve, [v])
    members = {"alpha_g1": one(g1, key["alpha"]), "beta_g2": one(g2, key["beta"]), "gamma_g2": one(g2, key["gamma"]),
               "delta_g2": one(g2, key["delta"]), "gamma_abc_g1": pts(g1, key["gamma_abc"]), "beta_g1": one(g1, key["beta"]),
               "delta_g1": one(g1, key["delta"]), "a_query": pts(g1, key["a_query"]), "b_g1_query": pts(g1, key["b_query"]),
               "b_g2_query": pts(g2, key["b_query"]), "h_query": pts(g1, key["h_query"]), "l_query": pts(g1, key["l_query"])}
    pk = az.ProvingKey.deserialize_unchecked(pairing, az.ProvingKey.serialize_unchecked(pairing, members))   # through the wire format
    mont = lambda v: orc.int_to_limbs(pyref.mont(field, v % p), 4)
    mats = [zk.groth16.R1csMatrix(field, [[(mont(c), j) for c, j in row] for row in r1cs[k]], n_cols=nvars) for k in "ABC"]
    z_mont = np.stack([mont(v) for v in z])
    # the mat-vecs alone, then the witness map, against the reference's evaluations / quotient
    ea, eb, ec = g16.evaluations(r1cs, z)
    d_z = to_device(zk, z_mont)
    for mtx, exp in zip(mats, (ea, eb, ec)):
        got = to_host(zk, mtx.matvec(d_z, to_device(zk, np.ones((m, 4), dtype=np.uint64))))
        exp = list(exp)
        if mtx is mats[0]:
            for j in range(num_inputs):
                exp[len(r1cs["A"]) + j] = 0          # the input-consistency rows are added by the witness map, not by the matrix
        assert (got == np.stack([mont(v) for v in exp])).all()
    h_exp = np.stack([mont(v) for v in g16.h_coefficients(r1cs, z)])
    bufs = [to_device(zk, np.zeros((m, 4), dtype=np.uint64)) for _ in range(3)]
    h = to_host(zk, zk.groth16.witness_map(field, mats[0], mats[1], mats[2], d_z, num_inputs, *bufs))
    assert (h == h_exp).all(), "witness map"
    # the proof
    prover = zk.groth16.Prover(pairing, pk, mats[0], mats[1], mats[2], num_inputs, lambda a: to_device(zk, a))
    rng = pyref.Rng(seed + 200)
    for r, s in ((rng.below(p), rng.below(p)), (0, 0), (1, p - 1)):
        (A, B, C), proof_bytes = prover.prove(z_mont, mont(r), mont(s))
        a, b, c = g16.prove_logs(r1cs, key, z, r, s)
        assert g16.verify_logs(r1cs, key, z[:num_inputs], a, b, c)
        assert (A == one(g1, a)[0]).all() and (B == one(g2, b)[0]).all() and (C == one(g1, c)[0]).all(), (pairing, r, s)
        dA, dB, dC = az.proof_from_bytes(pairing, proof_bytes)            # what the reference's `buy` would deserialize
        assert (dA == A).all() and (dB == B).all() and (dC == C).all()
        assert len(proof_bytes) == (192 if pairing == "Bls381" else 128)
    # a wrong witness gives a proof that does NOT verify (the quotient no longer divides): the path takes no shortcut
    bad = list(z)
    bad[-1] = (bad[-1] + 1) % p
    (A2, _, _), _ = prover.prove(np.stack([mont(v) for v in bad]), mont(3), mont(4))
    assert not (A2 == A).all()
    prover.free()


# ------------------------------------------------------------------ halo2 prover steps beyond commit / FFT (f4)
def _ints(name, arr):
    return orc.array_to_ints(orc.from_mont(name, np.ascontiguousarray(arr)))


def _monts(name, ints):
    p = pyref.FIELDS[name][0]
    return orc.to_mont(name, orc.ints_to_array([v % p for v in ints], 4)) if len(ints) else np.zeros((0, 4), dtype=np.uint64)


def check_batch_invert_and_scan(zk, name, n, seed=3):
    from oracle import pyref_halo2 as h2
    p = pyref.FIELDS[name][0]
    a = rand_field(name, n, seed)
    if n > 5:
        a[3] = 0
        a[n - 1] = 0                                   # zeros stay zero
    ai = _ints(name, a)
    got = to_host(zk, zk.halo2.batch_invert(name, to_device(zk, a)))
    assert (got == _monts(name, h2.batch_invert(name, ai))).all(), (name, n, "batch_invert")
    f = rand_field(name, n, seed + 1)
    first = rand_field(name, 1, seed + 2)[0]
    z, tot = h2.prefix_product(name, _ints(name, f), _ints(name, first.reshape(1, 4))[0])
    out, total = zk.halo2.prefix_product(name, to_device(zk, f), first=first, want_total=True)
    assert (to_host(zk, out) == _monts(name, z)).all(), (name, n, "prefix_product")
    assert (total == _monts(name, [tot])[0]).all()
    out2 = to_device(zk, np.zeros((n, 4), dtype=np.uint64))
    zk.halo2.prefix_product(name, to_device(zk, f), out=out2)
    z1, _ = h2.prefix_product(name, _ints(name, f))
    assert (to_host(zk, out2) == _monts(name, z1)).all()


def check_permutation_and_lookup_products(zk, name, k, ncols=5, seed=9):
    """the grand products of halo2's permutation argument (two chunks, the second continuing from the first) and lookup
    argument against the pure-Python restatement; for a genuine permutation / lookup the final value is 1"""
    from oracle import pyref_halo2 as h2
    p = pyref.FIELDS[name][0]
    n = 1 << k
    rng = pyref.Rng(seed)
    omega = pyref.root_of_unity(name, k)
    delta = pow(pyref.FIELDS[name][1], 1 << pyref.two_adicity(p)[0], p)      # halo2's DELTA = generator^(2^S): order t, outside the 2^S subgroup
    # a genuine copy-constraint system: values constant on the cycles of a random permutation of the (column, row) cells
    cells = [(c, i) for c in range(ncols) for i in range(n)]
    perm = list(range(len(cells)))
    for i in range(len(perm) - 1, 0, -1):
        j = rng.below(i + 1)
        perm[i], perm[j] = perm[j], perm[i]
    val, seen = [None] * len(cells), [False] * len(cells)
    for s0 in range(len(cells)):
        if not seen[s0]:
            v, t = rng.below(p), s0
            while not seen[t]:
                seen[t], val[t] = True, v
                t = perm[t]
    cols = [[val[c * n + i] for i in range(n)] for c in range(ncols)]
    label = lambda c, i: pow(delta, c, p) * pow(omega, i, p) % p
    sig = [[label(*cells[perm[c * n + i]]) for i in range(n)] for c in range(ncols)]
    beta, gamma = rng.below(p), rng.below(p)
    chunks = [(0, 3), (3, ncols)]
    zf, d_cols, d_sig = None, [to_device(zk, _monts(name, c)) for c in cols], [to_device(zk, _monts(name, s)) for s in sig]
    last_exp = 1
    for lo, hi in chunks:
        f = h2.permutation_factors(name, cols[lo:hi], sig[lo:hi], beta, gamma, delta, omega, lo)
        z_exp, last_exp = h2.prefix_product(name, f, last_exp)
        z_out = to_device(zk, np.zeros((n, 4), dtype=np.uint64))
        zf = zk.halo2.permutation_product(name, d_cols[lo:hi], d_sig[lo:hi], _monts(name, [beta])[0], _monts(name, [gamma])[0],
                                          _monts(name, [delta])[0], k, z_out, first_column_index=lo, z_first=zf)
        assert (to_host(zk, z_out) == _monts(name, z_exp)).all(), (name, k, lo)
        assert (zf == _monts(name, [last_exp])[0]).all()
    assert last_exp == 1, "a genuine permutation closes the product"
    # lookup: A' a permutation of A, S' of S with A'_i in {S'_i, A'_(i-1)} -- here the products only need multiset equality
    A = [rng.below(64) for _ in range(n)]
    S = list(range(64)) + [rng.below(64) for _ in range(n - 64)] if n >= 64 else [rng.below(p) for _ in range(n)]
    Ap, Sp = sorted(A), sorted(S)
    fexp = h2.lookup_factors(name, A, S, Ap, Sp, beta, gamma)
    z_exp, last = h2.prefix_product(name, fexp)
    z_out = to_device(zk, np.zeros((n, 4), dtype=np.uint64))
    got_last = zk.halo2.lookup_product(name, *[to_device(zk, _monts(name, v)) for v in (A, S, Ap, Sp)], _monts(name, [beta])[0],
                                       _monts(name, [gamma])[0], z_out)
    assert (to_host(zk, z_out) == _monts(name, z_exp)).all() and (got_last == _monts(name, [last])[0]).all()
    assert last == 1


def check_permute_expression_pair(zk, name, n=600, usable=590, seed=5):
    """halo2 lookup argument, permute_expression_pair (a CPU step upstream and in the mirror): against the literal restatement,
    then through zk_halo2_lookup_product_device -- for the permuted pair of a genuine lookup the grand product closes to 1"""
    from oracle import pyref_halo2 as h2
    p = pyref.FIELDS[name][0]
    rng = pyref.Rng(seed)
    table = [rng.below(p) for _ in range(40)] + [3, 3, 7]
    table = (table * (usable // len(table) + 1))[:usable]
    inputs = [table[rng.below(len(table))] for _ in range(usable)]
    pad = lambda v: v + [rng.below(p) for _ in range(n - usable)]
    a_exp, s_exp = h2.permute_expression_pair(name, inputs, table, usable)
    a_got, s_got = zk.halo2.permute_expression_pair(name, _monts(name, pad(list(inputs))), _monts(name, pad(list(table))), usable)
    assert (a_got == _monts(name, a_exp)).all() and (s_got == _monts(name, s_exp)).all(), name
    assert sorted(s_exp) == sorted(table) and all(a_exp[i] == s_exp[i] or a_exp[i] == a_exp[i - 1] for i in range(usable))
    beta, gamma = rng.below(p), rng.below(p)
    z_out = to_device(zk, np.zeros((usable, 4), dtype=np.uint64))
    last = zk.halo2.lookup_product(name, to_device(zk, _monts(name, inputs)), to_device(zk, _monts(name, table)), to_device(zk, a_got), to_device(zk, s_got),
                                   _monts(name, [beta])[0], _monts(name, [gamma])[0], z_out)
    assert (last == _monts(name, [1])[0]).all(), "the lookup product of a permuted pair closes to 1"
    try:
        zk.halo2.permute_expression_pair(name, _monts(name, [5, 6] + inputs[2:]), _monts(name, table), usable)
        raise AssertionError("an input outside the table must be refused")
    except ValueError:
        pass


def check_eval_polynomial(zk, name, sizes=(1, 2, 15, 16, 17, 1000, 4099), seed=77):
    """halo2 arithmetic.rs eval_polynomial on the device against Python integers (Horner); edge points 0, 1, p - 1"""
    p = pyref.FIELDS[name][0]
    rng = pyref.Rng(seed)
    for n in sizes:
        coeffs = [rng.below(p) for _ in range(n)]
        d = to_device(zk, _monts(name, coeffs))
        for x in (0, 1, p - 1, rng.below(p), rng.below(p)):
            exp = 0
            for c in reversed(coeffs):
                exp = (exp * x + c) % p
            got = zk.halo2.eval_polynomial(name, d, _monts(name, [x])[0])
            assert (got == _monts(name, [exp])[0]).all(), (name, n, x)
    # the multiopen Horner step a = a s + b
    n = 1000
    a, b, sc = [rng.below(p) for _ in range(n)], [rng.below(p) for _ in range(n)], rng.below(p)
    got = to_host(zk, zk.halo2.vec_muladd(name, to_device(zk, _monts(name, a)), to_device(zk, _monts(name, b)), _monts(name, [sc])[0]))
    assert (got == _monts(name, [(x_ * sc + y_) % p for x_, y_ in zip(a, b)])).all(), (name, "muladd")
    d_a, d_b, d_o = to_device(zk, _monts(name, a)), to_device(zk, _monts(name, b)), to_device(zk, np.zeros((n, 4), dtype=np.uint64))
    zk.halo2.vec_muladd(name, d_a, d_b, _monts(name, [sc])[0], out=d_o)
    assert (to_host(zk, d_o) == got).all() and (to_host(zk, d_a) == _monts(name, a)).all(), (name, "muladd into a third buffer")
    # the whole Horner fold of several polynomials in one pass, forwards (a point set with x_1) and backwards (h's pieces with x^n)
    n, count = 700, 6
    polys = [[rng.below(p) for _ in range(n)] for _ in range(count)]
    d_polys = to_device(zk, np.stack([_monts(name, c) for c in polys]))
    for reverse in (False, True):
        order = polys[::-1] if reverse else polys
        exp = list(order[0])
        for q in order[1:]:
            exp = [(e_ * sc + v_) % p for e_, v_ in zip(exp, q)]
        got = to_host(zk, zk.halo2.vec_fold_many(name, to_device(zk, np.zeros((n, 4), dtype=np.uint64)), d_polys, _monts(name, [sc])[0], reverse=reverse))
        assert (got == _monts(name, exp)).all(), (name, "fold_many", reverse)
    one = zk.halo2.vec_fold_many(name, to_device(zk, np.zeros((n, 4), dtype=np.uint64)), d_polys[:1], _monts(name, [sc])[0])
    assert (to_host(zk, one) == _monts(name, polys[0])).all()
    # several polynomials at one point in one launch
    n, count = 300, 5
    polys = [[rng.below(p) for _ in range(n)] for _ in range(count)]
    x = rng.below(p)
    got = zk.halo2.eval_polynomials(name, to_device(zk, np.stack([_monts(name, c) for c in polys])), _monts(name, [x])[0])
    for q in range(count):
        exp = 0
        for c in reversed(polys[q]):
            exp = (exp * x + c) % p
        assert (got[q] == _monts(name, [exp])[0]).all(), (name, "batch", q)


def check_kate_division(zk, name, sizes=(1, 2, 15, 16, 17, 4095, 4096, 4097, 9000), seed=83):
    """halo2 arithmetic.rs kate_division on the device (three-phase suffix scan) against the published loop on Python integers:
    sizes around the lane (16) and workgroup (4096) boundaries, edge points 0 / 1 / p - 1, in place and out of place; then the
    multiopen use: a point set's polynomial divided by each of its points in turn, the sets folded with x_2"""
    from oracle import pyref_halo2 as h2
    p = pyref.FIELDS[name][0]
    rng = pyref.Rng(seed)
    for n in sizes:
        a = [rng.below(p) for _ in range(n)]
        for x in (rng.below(p), 0, 1, p - 1)[: 4 if n <= 4097 else 1]:
            exp = h2.kate_division(name, a, x) + [0]
            d = to_device(zk, _monts(name, a))
            out = to_device(zk, np.full((n, 4), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))
            zk.halo2.kate_division(name, d, _monts(name, [x])[0], out=out)
            assert (to_host(zk, out) == _monts(name, exp)).all(), (name, n, x, "out of place")
            assert (to_host(zk, d) == _monts(name, a)).all()
            zk.halo2.kate_division(name, d, _monts(name, [x])[0])
            assert (to_host(zk, d) == _monts(name, exp)).all(), (name, n, x, "in place")
    # the powers of a point (the argument's vector b)
    for n in (1, 2, 1023, 1025, 5000):
        x = rng.below(p)
        got = to_host(zk, zk.halo2.vec_powers(name, to_device(zk, np.zeros((n, 4), dtype=np.uint64)), _monts(name, [x])[0]))
        assert (got == _monts(name, [pow(x, i, p) for i in range(n)])).all(), (name, n, "powers")
    # the multiopen quotient: three point sets of 1, 2 and 3 points
    n = 5000
    polys = [[rng.below(p) for _ in range(n)] for _ in range(3)]
    sets = [[rng.below(p)], [rng.below(p), rng.below(p)], [rng.below(p), rng.below(p), rng.below(p)]]
    x2 = rng.below(p)
    exp = h2.multiopen_quotient(name, polys, sets, x2, n)
    acc = None
    for poly, pts in zip(polys, sets):
        d = to_device(zk, _monts(name, poly))
        for pt in pts:
            zk.halo2.kate_division(name, d, _monts(name, [pt])[0])
        acc = d if acc is None else zk.halo2.vec_muladd(name, acc, d, _monts(name, [x2])[0])
    assert (to_host(zk, acc) == _monts(name, exp)).all(), (name, "multiopen quotient")


def check_kate_division_at_size(zk, name, k, seed=87):
    """2^k coefficients: q(X) (X - x) + a(x) = a(X), checked coefficient by coefficient on sampled positions (a[j] = q[j - 1] - x q[j])
    across lane / workgroup boundaries and at both ends, with a(x) from zk_poly_eval_device"""
    p = pyref.FIELDS[name][0]
    n = 1 << k
    a = rand_field(name, n, seed)
    x = pyref.Rng(seed).below(p)
    q = to_host(zk, zk.halo2.kate_division(name, to_device(zk, a), _monts(name, [x])[0], out=to_device(zk, np.zeros((n, 4), dtype=np.uint64))))
    rng = pyref.Rng(seed + 1)
    pos = sorted({1, 2, 15, 16, 17, 4095, 4096, 4097, n // 2, n - 4097, n - 4096, n - 2, n - 1} | {1 + rng.below(n - 1) for _ in range(200)})
    pos = [j for j in pos if 1 <= j < n]
    ai, qi, qm = _ints(name, a[pos]), _ints(name, q[pos]), _ints(name, q[[j - 1 for j in pos]])
    assert all((qm[t] - x * qi[t] - ai[t]) % p == 0 for t in range(len(pos))), (name, k)
    assert not q[n - 1].any()
    ax = zk.halo2.eval_polynomial(name, to_device(zk, a), _monts(name, [x])[0])
    a0, q0 = _ints(name, a[:1])[0], _ints(name, q[:1])[0]
    assert (_ints(name, ax.reshape(1, 4))[0] - x * q0 - a0) % p == 0, (name, k, "remainder")


def check_ipa(zk, cname, k, seed=13):
    """halo2_proofs 0.2 inner-product argument rounds on the device (two MSMs, two inner products, three folds per round)
    against the pure-Python restatement; the folded generator equals <s, G> with s_i = prod_j u_j^(bit_j(i))"""
    from oracle import pyref_halo2 as h2
    sf = pyref.CURVES[cname][1]
    r = pyref.FIELDS[sf][0]
    n = 1 << k
    rng = pyref.Rng(seed)
    gens = bases_for(cname, n, seed=31)
    L = orc.coord_limbs(cname)
    bf = pyref.CURVES[cname][0]
    unm = lambda row: tuple(orc.limbs_to_int(orc.from_mont(bf, row[j * L:(j + 1) * L].reshape(1, L))[0]) for j in range(2))
    g_py = [unm(gens[i]) for i in range(n)]
    pp = [rng.below(r) for _ in range(n)]
    x3 = rng.below(r)
    b = [pow(x3, i, r) for i in range(n)]
    us = [1 + rng.below(r - 1) for _ in range(k)]
    for j, edge in zip(range(k - 1, -1, -1), (r - 1, 1, 0xFFFF, (1 << 200) + 12345)):   # edge challenges on the small rounds
        us[j] = edge
    rounds, c_fin, b_fin, g_fin = h2.ipa_argument(cname, pp, b, g_py, us)
    ipa = zk.halo2.IpaProver(cname, to_device(zk, _monts(sf, pp)), to_device(zk, _monts(sf, b)), to_device(zk, gens.copy()))
    aff = lambda P: np.zeros(2 * L, dtype=np.uint64) if P is None else np.concatenate([orc.int_to_limbs(pyref.mont(bf, c), L) for c in P])
    g_after = {}
    for j in range(k):
        Lj, Rj, vl, vr = ipa.round()
        eL, eR, evl, evr = rounds[j]
        assert (zk.point_to_affine(cname, Lj) == aff(eL)).all() and (zk.point_to_affine(cname, Rj) == aff(eR)).all(), (cname, k, j)
        assert (vl == _monts(sf, [evl])[0]).all() and (vr == _monts(sf, [evr])[0]).all()
        ipa.fold(_monts(sf, [us[j]])[0])
        g_after[j + 1] = to_host(zk, ipa.g)[:n >> (j + 1)].copy()
    assert (to_host(zk, ipa.p)[0] == _monts(sf, [c_fin])[0]).all() and (to_host(zk, ipa.b)[0] == _monts(sf, [b_fin])[0]).all()
    assert (to_host(zk, ipa.g)[0] == aff(g_fin)).all()
    ipa.free()
    # the same argument without folding the generators: every L, R over the ORIGINAL generators, the challenges collected in W
    srs = zk.Bases(cname, gens)
    new_buffer = lambda shape: to_device(zk, np.zeros(shape, dtype=np.uint64))
    vipa = zk.halo2.IpaProverVirtual(cname, to_device(zk, _monts(sf, pp)), to_device(zk, _monts(sf, b)), srs, new_buffer)
    for j in range(k):
        Lj, Rj, vl, vr = vipa.round()
        eL, eR, evl, evr = rounds[j]
        assert (zk.point_to_affine(cname, Lj) == aff(eL)).all() and (zk.point_to_affine(cname, Rj) == aff(eR)).all(), (cname, k, j, "virtual")
        assert (vl == _monts(sf, [evl])[0]).all() and (vr == _monts(sf, [evr])[0]).all()
        vipa.fold(_monts(sf, [us[j]])[0])
    assert (to_host(zk, vipa.p)[0] == _monts(sf, [c_fin])[0]).all()
    assert (zk.point_to_affine(cname, vipa.folded_generator()) == aff(g_fin)).all()
    vipa.free()
    # ... and leaving the fold-free form on the way: after some rounds the generators are materialised in one step
    # (zk_ipa_collapse_device) -- they must be what the literal folds left -- and the later rounds run over them
    stops = sorted({min(3, k), min(5, k), k})
    vipa = zk.halo2.IpaProverVirtual(cname, to_device(zk, _monts(sf, pp)), to_device(zk, _monts(sf, b)), srs, new_buffer)
    for j in range(k):
        Lj, Rj, vl, vr = vipa.round()
        eL, eR, evl, evr = rounds[j]
        assert (zk.point_to_affine(cname, Lj) == aff(eL)).all() and (zk.point_to_affine(cname, Rj) == aff(eR)).all(), (cname, k, j, "collapsed")
        vipa.fold(_monts(sf, [us[j]])[0])
        if j + 1 in stops:
            g = to_host(zk, vipa.collapse())
            assert (g == g_after[j + 1]).all(), (cname, k, j + 1, "collapse")
    assert (g[0] == aff(g_fin)).all()
    assert (to_host(zk, vipa.p)[0] == _monts(sf, [c_fin])[0]).all()
    vipa.free()
    srs.free()


def check_ipa_collapse_edges(zk, cname, k=6, seed=19):
    """zk_ipa_collapse_device at its edges: no round done yet (T = 1: a copy of the generators), every round done (one
    survivor, T = 2^k scalars), and zk_ipa_collapse_range_device shares that tile the output -- all against the literal folds"""
    import ctypes
    from oracle import pyref_halo2 as h2
    sf = pyref.CURVES[cname][1]
    r = pyref.FIELDS[sf][0]
    n = 1 << k
    rng = pyref.Rng(seed)
    gens = bases_for(cname, n, seed=33)
    L = orc.coord_limbs(cname)
    us = [1 + rng.below(r - 1) for _ in range(k)]
    # literal folds on the device (IpaProver) give the reference generators after every round
    ipa = zk.halo2.IpaProver(cname, to_device(zk, rand_field(sf, n, 1)), to_device(zk, rand_field(sf, n, 2)), to_device(zk, gens.copy()))
    g_after = {0: gens.copy()}
    for j in range(k):
        ipa.fold(_monts(sf, [us[j]])[0])
        g_after[j + 1] = to_host(zk, ipa.g)[:n >> (j + 1)].copy()
    ipa.free()
    srs = zk.Bases(cname, gens)
    new_buffer = lambda shape: to_device(zk, np.zeros(shape, dtype=np.uint64))
    plib = zk.halo2._plib()
    for rounds in (0, 2, k):
        v = zk.halo2.IpaProverVirtual(cname, to_device(zk, rand_field(sf, n, 1)), to_device(zk, rand_field(sf, n, 2)), srs, new_buffer)
        for j in range(rounds):
            v.fold(_monts(sf, [us[j]])[0])
        cur = n >> rounds
        # shares [0, a), [a, cur) through the range entry, then the whole thing
        a = max(1, cur // 3)
        parts = []
        for first, count in ((0, a), (a, cur - a)):
            if count == 0:
                continue
            out = new_buffer((count, 2 * L))
            st = plib.zk_ipa_collapse_range_device(zk.curve_id(cname), srs.handle, zk._ptr(v.W), n, cur, first, count, zk._ptr(out), ctypes.c_void_p(0))
            assert st == 0, (cname, rounds, first, count, st)
            parts.append(to_host(zk, out).reshape(count, 2 * L))
        assert (np.concatenate(parts) == g_after[rounds]).all(), (cname, rounds, "ranges")
        assert (to_host(zk, v.collapse()) == g_after[rounds]).all(), (cname, rounds, "whole")
        v.free()
    srs.free()


def check_expression(zk, name, k, ext=2, seed=21):
    """the quotient-numerator evaluator: a small gate set in the style of the reference's circuit (a multiplication gate
    behind a selector, a Pow5-style S-box with a rotation, a boolean check), folded with y, over extended-domain columns"""
    from oracle import pyref_halo2 as h2
    p = pyref.FIELDS[name][0]
    ek = k + ext
    ne, scale = 1 << ek, 1 << ext
    cols = [rand_field(name, ne, seed + c) for c in range(5)]          # a, b, c, q_mul, q_pow on the extended domain
    ci = [_ints(name, c) for c in cols]
    y = 0x1234567 % p
    consts = [y, 5, 1]
    a, b, c, qm, qp = (("col", i, 0) for i in range(5))
    prog = [qm, a, b, ("mul",), c, ("sub",), ("mul",),                                 # q_mul (a b - c)
            ("scale", 0),                                                               # * y
            qp, a, a, ("mul",), a, ("mul",), a, ("mul",), a, ("mul",), ("col", 1, 1), ("sub",), ("const", 1), ("add",), ("mul",),   # q_pow (a^5 - b(omega X) + 5)
            ("add",), ("scale", 0),
            c, c, ("const", 2), ("sub",), ("mul",), ("add",),                          # c (c - 1)
            ("col", 0, -3), ("neg",), ("add",)]                                         # - a(omega^-3 X)
    exp = [h2.eval_program(name, prog, ci, consts, ne, scale, i) for i in range(ne)]
    out = to_device(zk, np.zeros((ne, 4), dtype=np.uint64))
    zk.halo2.evaluate_expression(name, prog, [to_device(zk, c) for c in cols], _monts(name, consts), ek, scale, out)
    assert (to_host(zk, out) == _monts(name, exp)).all(), (name, k)
    # the lazy-limb evaluator: columns in the R' = 2^261 radix (times 2^5), constants and output in the usual form
    lazy_cols = [zk.halo2.to_lazy_form(name, to_device(zk, c)) for c in cols]
    for mode in ("never", "always"):      # the interpreter kernel / the kernel specialised for this program (hiprtc; GPU build only)
        zk.halo2.expr_configure(mode)
        out2 = to_device(zk, np.zeros((ne, 4), dtype=np.uint64))
        zk.halo2.evaluate_expression(name, prog, lazy_cols, _monts(name, consts), ek, scale, out2, lazy=True)
        assert (to_host(zk, out2) == _monts(name, exp)).all(), (name, k, "lazy limbs", mode)
    # adversarial bounds for the host's bound walk: long chains of additions / subtractions / negations before a product, at the
    # extreme stored values (all columns p - 1, or 0)
    top = np.tile(orc.int_to_limbs(p - 1, 4), (ne, 1))       # as stored words: x R' = p - 1
    zero = np.zeros((ne, 4), dtype=np.uint64)
    chain = [a] + [a, ("add",)] * 40 + [b] + [b, ("sub",)] * 9 + [("neg",), ("mul",), c, ("neg",), ("neg",), ("sub",)] + [c, ("add",)] * 70 + [qm, ("mul",), ("neg",)]
    for fill in (top, zero):
        ccols = [to_device(zk, fill) for _ in range(5)]
        xval = (p - 1 if fill is top else 0) * pow(1 << 261, -1, p) % p      # the value x behind the stored word x R'
        icol = [[xval] * ne for _ in range(5)]
        e0 = h2.eval_program(name, chain, icol, consts, ne, scale, 0)
        for mode in ("never", "always"):
            zk.halo2.expr_configure(mode)
            out3 = to_device(zk, np.zeros((ne, 4), dtype=np.uint64))
            zk.halo2.evaluate_expression(name, chain, ccols, _monts(name, consts), ek, scale, out3, lazy=True)
            assert (to_host(zk, out3) == _monts(name, [e0])[0]).all(), (name, k, "lazy limbs at the bounds", mode)
    zk.halo2.expr_configure("auto")
    for bad in ([("add",)], [a, b], [("col", 9, 0)], [("const", 7)], [a] * 9 + [("add",)] * 8):   # malformed programs are refused on the host
        try:
            zk.halo2.evaluate_expression(name, bad, [to_device(zk, c) for c in cols], _monts(name, consts), ek, scale, out)
            raise AssertionError("accepted " + repr(bad))
        except zk.ZkError:
            pass


# ------------------------------------------------------------------ BASELINE configs[2] at its own shapes (k = 20)
class _IntColumn:
    """a Montgomery uint64 [n, 4] column seen as Python ints, converted on access (pyref_halo2.eval_program indexes it)"""

    def __init__(self, name, arr):
        self.name, self.arr = name, arr

    def __getitem__(self, i):
        return orc.limbs_to_int(orc.from_mont(self.name, np.ascontiguousarray(self.arr[i]).reshape(1, 4))[0])


def check_expression_at_size(zk, name, k, ext, n_adv=13, n_fix=8, n_inst=3, samples=64, seed=0x9000, parts=1):
    """zk_expr_eval_device with the bench's own quotient program (contangle-zkcp_amd/synth.py quotient_program(13, 8, 3): 268 ops
    over 30 extended columns) at 2^(k + ext) / parts rows (parts > 1: one sub-coset of a sharded quotient, rotations scaled down
    accordingly), checked on sampled rows -- the first and last rows, whose rotations wrap around, and seeded random ones --
    against pyref_halo2.eval_program on Python integers"""
    from contangle_zkcp_amd import synth
    from oracle import pyref_halo2 as h2
    ne, scale = (1 << (k + ext)) // parts, (1 << ext) // parts
    ek = ne.bit_length() - 1
    prog = synth.quotient_program(n_adv, n_fix, n_inst)
    ncols = n_adv + n_fix + 6 + n_inst
    cols = [rand_field(name, ne, seed + c) for c in range(ncols)]
    consts = rand_field(name, 5, seed + 99)
    d_cols = [to_device(zk, c) for c in cols]
    out = to_device(zk, np.zeros((ne, 4), dtype=np.uint64))
    zk.halo2.evaluate_expression(name, prog, d_cols, consts, ek, scale, out)
    got = to_host(zk, out)
    # ... and the lazy-limb evaluator on the same columns in the R' radix
    for d in d_cols:
        zk.halo2.to_lazy_form(name, d)
    # ... by the interpreter kernel and by the kernel specialised for this program (hiprtc; the CPU emulator has only the former)
    for mode in ("never", "always"):
        zk.halo2.expr_configure(mode)
        out.zero_() if hasattr(out, "zero_") else out.fill(0)
        zk.halo2.evaluate_expression(name, prog, d_cols, consts, ek, scale, out, lazy=True)
        assert (to_host(zk, out) == got).all(), (name, k, ext, "lazy limbs differ from the saturated evaluator", mode)
    zk.halo2.expr_configure("auto")
    rng = pyref.Rng(seed)
    rows = sorted({0, 1, scale - 1, scale, scale + 1, ne - 1, ne - 2, ne - scale, ne - scale - 1, ne // 2, ne // 2 - 1}
                  | {rng.below(ne) for _ in range(samples)})
    rows = [r for r in rows if 0 <= r < ne]
    icols = [_IntColumn(name, c) for c in cols]
    iconsts = _ints(name, consts)
    exp = [h2.eval_program(name, prog, icols, iconsts, ne, scale, i) for i in rows]
    assert (got[rows] == _monts(name, exp)).all(), (name, k, ext, [r for r, g_, e_ in zip(rows, got[rows], _monts(name, exp)) if (g_ != e_).any()][:4])
    return len(rows)


def _dot_mod(a, b, r):
    """sum a_i b_i mod r over numpy object arrays of Python ints"""
    return int((a * b).sum() % r)


def check_ipa_at_size(zk, cname, k, collapse_after, seed=0x1A, survivors=16, gens_from_device=True):
    """The bench's opening argument at its own shape: IpaProverVirtual (zk_ipa_round_device) for k rounds over 2^k generators
    G_i = [g_i] G with KNOWN logarithms, the generators materialised after `collapse_after` rounds (zk_ipa_collapse_device,
    m0 = 2^k -> 2^(k - collapse_after) survivors that share 2^collapse_after scalars).  Checked:
      * every round's L_j, R_j against [<p'_hi, g_lo>] G, [<p'_lo, g_hi>] G (the argument in the exponent, Python integers),
        value_l / value_r against the inner products;
      * L_0, R_0 and the first round after the collapse also against the oracle's Pippenger on host-built scalars / the
        collapsed generators read back from the device;
      * `survivors` sampled outputs of the collapse, each against an oracle MSM over its 2^collapse_after original points with
        W_t from Python integers, and against [g'_i] G;
      * the final p', b, and the folded generator against MSM(G0, s), s_i = prod_j u_j^(bit_j(i))."""
    sf = pyref.CURVES[cname][1]
    r = pyref.FIELDS[sf][0]
    n = 1 << k
    L = orc.coord_limbs(cname)
    ks = scalars_for(cname, n, seed)
    d_pts = to_device(zk, np.zeros((n, 2 * L), dtype=np.uint64))
    zk.fixed_base_msm_device(cname, to_device(zk, ks), d_pts, n)
    pts = to_host(zk, d_pts).reshape(n, 2 * L).copy()
    G = orc.curve_generator(cname)
    for i in (0, n // 3, n - 1):
        assert (pts[i] == orc.scalar_mul(cname, G, ks[i])).all()
    srs = zk.Bases(cname, device_tensor=d_pts, n=n) if not isinstance(d_pts, np.ndarray) else zk.Bases(cname, pts)
    as_obj = lambda arr: np.array(orc.array_to_ints(arr), dtype=object)
    pp_m, b_m = rand_field(sf, n, seed + 1), rand_field(sf, n, seed + 2)
    pp, bb, gg = as_obj(orc.from_mont(sf, pp_m)), as_obj(orc.from_mont(sf, b_m)), as_obj(ks)
    rng = pyref.Rng(seed + 3)
    us = [1 + rng.below(r - 1) for _ in range(k)]
    new_buffer = lambda shape: to_device(zk, np.zeros(shape, dtype=np.uint64))
    vipa = zk.halo2.IpaProverVirtual(cname, to_device(zk, pp_m), to_device(zk, b_m), srs, new_buffer)
    point = lambda e: orc.scalar_mul(cname, G, orc.int_to_limbs(e % r, 4))
    aff = lambda jac: zk.point_to_affine(cname, jac)
    g_dev = None
    for j in range(k):
        cur = n >> j
        half = cur // 2
        Lj, Rj, vl, vr = vipa.round()
        assert (aff(Lj) == point(_dot_mod(pp[half:cur], gg[:half], r))).all(), (cname, k, j, "L")
        assert (aff(Rj) == point(_dot_mod(pp[:half], gg[half:cur], r))).all(), (cname, k, j, "R")
        assert (vl == _monts(sf, [_dot_mod(pp[half:cur], bb[:half], r)])[0]).all() and (vr == _monts(sf, [_dot_mod(pp[:half], bb[half:cur], r)])[0]).all(), (cname, k, j, "values")
        if j == 0 or j == collapse_after:      # the same two sums through the oracle's Pippenger on the points themselves
            base_pts = pts if j == 0 else g_dev
            sc = orc.ints_to_array([int(v) for v in pp[:cur]], 4)
            assert (aff(Lj) == orc.msm_ark(cname, base_pts[:half], sc[half:cur], threads=os.cpu_count() or 8)).all(), (cname, k, j, "L vs oracle")
            assert (aff(Rj) == orc.msm_ark(cname, base_pts[half:cur], sc[:half], threads=os.cpu_count() or 8)).all(), (cname, k, j, "R vs oracle")
        u, ui = us[j], pow(us[j], -1, r)
        pp = (pp[:half] + pp[half:cur] * ui) % r
        bb = (bb[:half] + bb[half:cur] * u) % r
        gg = (gg[:half] + gg[half:cur] * u) % r
        vipa.fold(_monts(sf, [u])[0])
        if j + 1 == collapse_after and j + 1 < k:
            g_dev = to_host(zk, vipa.collapse()).reshape(half, 2 * L).copy()
            T = n // half
            # W_t = the product of the challenges that t's bits select: fold a (0-based) halves the vector at bit (k - 1 - a)
            Wt = []
            for t in range(T):
                w = 1
                for a in range(collapse_after):
                    if (t >> (collapse_after - 1 - a)) & 1:
                        w = w * us[a] % r
                Wt.append(w)
            wsc = orc.ints_to_array(Wt, 4)
            picks = sorted({0, 1, half - 1, half // 2} | {rng.below(half) for _ in range(survivors)})
            for i in picks:
                sub = np.ascontiguousarray(pts[i::half][:T])
                assert (g_dev[i] == orc.msm_ark(cname, sub, wsc, threads=4)).all(), (cname, k, i, "collapsed generator vs oracle MSM")
                assert (g_dev[i] == point(int(gg[i]))).all(), (cname, k, i, "collapsed generator in the exponent")
    assert (to_host(zk, vipa.p)[0] == _monts(sf, [int(pp[0])])[0]).all() and (to_host(zk, vipa.b)[0] == _monts(sf, [int(bb[0])])[0]).all()
    # s_i = prod_j u_j^(bit_j(i)) over ALL k rounds: <s, g> = the folded generator's logarithm
    s = np.array([1], dtype=object)
    for u in reversed(us):                      # the last challenge pairs neighbours (bit 0), the first halves the vector
        s = np.concatenate([s, (s * u) % r])
    assert _dot_mod(s, as_obj(ks), r) == int(gg[0]) % r
    assert (aff(vipa.folded_generator()) == point(int(gg[0]))).all(), (cname, k, "folded generator")
    vipa.free()
    srs.free()
