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
    """n uniform elements of the field, Montgomery form, uint64 [n,4] (vectorised splitmix64 + rejection)."""
    p = pyref.FIELDS[name][0]
    out = np.zeros((n, 4), dtype=np.uint64)
    todo = np.arange(n)
    ctr = np.uint64(seed)
    top_mask = np.uint64((1 << (p.bit_length() - 192)) - 1)
    pl = [np.uint64((p >> (64 * i)) & 0xFFFFFFFFFFFFFFFF) for i in range(4)]
    rnd = 0
    while todo.size:
        m = todo.size
        with np.errstate(over="ignore"):
            idx = (np.arange(m * 4, dtype=np.uint64) + np.uint64(rnd * 0x1000003) * np.uint64(n * 4 + 1)
                   + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15))
            z = idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        c = z.reshape(m, 4).copy()
        c[:, 3] &= top_mask
        lt = np.zeros(m, dtype=bool)
        eq = np.ones(m, dtype=bool)
        for i in (3, 2, 1, 0):
            lt |= eq & (c[:, i] < pl[i])
            eq &= c[:, i] == pl[i]
        out[todo[lt]] = c[lt]
        todo = todo[~lt]
        rnd += 1
    return out


def scalars_for(curve, n, seed, realistic=False):
    """canonical scalars [n,4] in [0, r); `realistic` = 40% zeros, 25% ones, 10% < 2^8."""
    sf = pyref.CURVES[curve][1]
    s = rand_field(sf, n, seed)   # uniform in [0, r): read as canonical integers
    if realistic:
        u = rand_field(sf, n, seed + 1)[:, 0] % np.uint64(100)
        z, o, sm = u < 40, (u >= 40) & (u < 65), (u >= 65) & (u < 75)
        s[z] = 0
        s[o] = 0
        s[o, 0] = 1
        s[sm, 1:] = 0
        s[sm, 0] &= np.uint64(0xFF)
    return s


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
            for wb in (0, 3, 7):
                got = affine_of(zk, cname, zk.msm(bases, sc, window_bits=wb))
                assert (got == exp).all(), (cname, n, wb)
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
    lazy = True   # buckets run on lazy limbs (9 x 29 bits; 14 x 28 for BLS12-381; pairs of those on the G2 twists)
    assert zk.msm_last_profile()["limb_bits"] == (29 if lazy and os.environ.get("ZK_MSM_F29") != "0" else 32)
    if lazy:                                                  # and must agree with the saturated 32-bit path
        os.environ["ZK_MSM_F29"] = "0"
        try:
            got32 = affine_of(zk, cname, zk.msm(bases, sc, window_bits=window_bits))
            assert zk.msm_last_profile()["limb_bits"] == 32
        finally:
            os.environ.pop("ZK_MSM_F29")
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

def check_msm_slice_lengths(zk, cname, n, window_bits):
    """the bucket reduction for several slice lengths L (X_t = W_t + [t L] S_t with the 2-bit windowed multiplier):
    powers of two (the digits of t, then log2 L doublings), a non-power of two (digits of t L), L = all buckets"""
    pts = bases_for(cname, n)
    sc = scalars_for(cname, n, 23)
    exp = orc.msm_ark(cname, pts, sc, threads=8)
    bases = zk.Bases(cname, pts)
    try:
        for L in (1, 2, 3, 8, 64, 1024):
            os.environ["ZK_MSM_SLICE"] = str(L)
            got = affine_of(zk, cname, zk.msm(bases, sc, window_bits=window_bits))
            assert (got == exp).all(), (cname, L)
    finally:
        os.environ.pop("ZK_MSM_SLICE", None)
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
    try:
        for k in (0, 1, 2, 3):
            os.environ["ZK_MSM_SPLIT"] = str(k)
            got = affine_of(zk, cname, zk.msm(bases, sc, window_bits=window_bits))
            assert (got == exp).all(), (cname, n, k)
    finally:
        os.environ.pop("ZK_MSM_SPLIT", None)
    bases.free()
