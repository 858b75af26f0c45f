"""TEST INFRASTRUCTURE -- pure-Python big-integer reference.

The independent pin for the C oracle and the HIP path:
Python `int` arithmetic only (`%`, `pow`), naive double-and-add MSM and O(n^2) DFT.
Nothing here shares code or constant tables with oracle/zk_oracle.c or with the
product (`contangle-zkcp_amd/csrc`): the moduli below are typed from SURVEY.md
Appendix A (pasta_curves 0.4, ark-bn254 0.3, ark-bls12-381 0.3).

PARITY UNPINNED: no reference outputs exist for this path (the reference's MSM/NTT
live in un-vendored crates and its tests record no vectors); these fixtures pin the
mathematics, which determines MSM/NTT outputs uniquely once normalised.

`python oracle/pyref.py` regenerates tests/golden/*.json (seeds recorded inside).
"""
import json
import os

FIELDS = {
    # name: (modulus, multiplicative generator, limbs64)
    "PallasFp": (0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001, 5, 4),
    "PallasFq": (0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001, 5, 4),
    "Bn254Fr": (21888242871839275222246405745257275088548364400416034343698204186575808495617, 5, 4),
    "Bls381Fr": (0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001, 7, 4),
    "Bn254Fq": (21888242871839275222246405745257275088696311157297823662689037894645226208583, 3, 4),
    "Bls381Fq": (0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, 2, 6),
}
FIELD_IDS = ["PallasFp", "PallasFq", "Bn254Fr", "Bls381Fr", "Bn254Fq", "Bls381Fq"]

CURVES = {
    # name: (base field, scalar field, b, gx, gy)
    "Pallas": ("PallasFp", "PallasFq", 5, FIELDS["PallasFp"][0] - 1, 2),
    "Vesta": ("PallasFq", "PallasFp", 5, FIELDS["PallasFq"][0] - 1, 2),
    "Bn254G1": ("Bn254Fq", "Bn254Fr", 3, 1, 2),
    "Bls381G1": ("Bls381Fq", "Bls381Fr", 4,
                 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
                 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
}
# G2 twists: coordinates in Fq2 = Fq[u]/(u^2+1), written as (c0, c1) tuples
CURVES["Bn254G2"] = ("Bn254Fq", "Bn254Fr",
                     (19485874751759354771024239261021720505790618469301721065564631296452457478373,
                      266929791119991161246907387137283842545076965332900288569378510910307636690),
                     (10857046999023057135944570762232829481370756359578518086990519993285655852781,
                      11559732032986387107991004021392285783925812861821192530917403151452391805634),
                     (8495653923123431417604973247489272438418190587263600148770280649306958101930,
                      4082367875863433681332203403145435568316851327593401208105741076214120093531))
CURVES["Bls381G2"] = ("Bls381Fq", "Bls381Fr", (4, 4),
                      (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
                       0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
                      (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
                       0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))
CURVE_IDS = ["Pallas", "Vesta", "Bn254G1", "Bls381G1", "Bn254G2", "Bls381G2"]


def is_g2(curve):
    return isinstance(CURVES[curve][2], tuple)

MASK64 = (1 << 64) - 1


def splitmix64(state):
    """One step of splitmix64; returns (new_state, output)."""
    state = (state + 0x9E3779B97F4A7C15) & MASK64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return state, z ^ (z >> 31)


class Rng:
    def __init__(self, seed):
        self.s = seed & MASK64

    def u64(self):
        self.s, z = splitmix64(self.s)
        return z

    def below(self, m):
        """Uniform in [0, m) by rejection on bit_length(m) bits."""
        nb = m.bit_length()
        nl = (nb + 63) // 64
        while True:
            v = 0
            for i in range(nl):
                v |= self.u64() << (64 * i)
            v &= (1 << nb) - 1
            if v < m:
                return v


def two_adicity(p):
    s, t = 0, p - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    return s, t


def root_of_unity(field, logn):
    p, g, _ = FIELDS[field]
    s, t = two_adicity(p)
    return pow(pow(g, t, p), 1 << (s - logn), p)


def mont(field, x):
    p, _, nl = FIELDS[field]
    return x * (1 << (64 * nl)) % p


def unmont(field, x):
    p, _, nl = FIELDS[field]
    return x * pow(1 << (64 * nl), -1, p) % p


# ------------------------------------------------------------------ coordinate-field arithmetic: ints (Fq) or (c0, c1) tuples (Fq2)
class K:
    """Arithmetic in the coordinate field of `curve`."""

    def __init__(self, curve):
        self.p = FIELDS[CURVES[curve][0]][0]
        self.ext = 2 if is_g2(curve) else 1

    def add(self, a, b):
        return (a + b) % self.p if self.ext == 1 else ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)

    def sub(self, a, b):
        return (a - b) % self.p if self.ext == 1 else ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def mul(self, a, b):
        if self.ext == 1:
            return a * b % self.p
        return ((a[0] * b[0] - a[1] * b[1]) % self.p, (a[0] * b[1] + a[1] * b[0]) % self.p)

    def small(self, k, a):
        return k * a % self.p if self.ext == 1 else (k * a[0] % self.p, k * a[1] % self.p)

    def inv(self, a):
        if self.ext == 1:
            return pow(a, -1, self.p)
        n = pow(a[0] * a[0] + a[1] * a[1], -1, self.p)
        return (a[0] * n % self.p, -a[1] * n % self.p)

    def zero(self, a):
        return a == 0 if self.ext == 1 else a == (0, 0)

    def neg(self, a):
        return (-a) % self.p if self.ext == 1 else ((-a[0]) % self.p, (-a[1]) % self.p)


# ------------------------------------------------------------------ affine curve arithmetic (None = infinity)
def ec_add(curve, P, Q):
    k = K(curve)
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if k.zero(k.add(y1, y2)):
            return None
        lam = k.mul(k.small(3, k.mul(x1, x1)), k.inv(k.small(2, y1)))
    else:
        lam = k.mul(k.sub(y2, y1), k.inv(k.sub(x2, x1)))
    x3 = k.sub(k.sub(k.mul(lam, lam), x1), x2)
    return x3, k.sub(k.mul(lam, k.sub(x1, x3)), y1)


def ec_neg(curve, P):
    return None if P is None else (P[0], K(curve).neg(P[1]))


def ec_mul(curve, k, P):
    R = None
    while k:
        if k & 1:
            R = ec_add(curve, R, P)
        P = ec_add(curve, P, P)
        k >>= 1
    return R


def ec_on_curve(curve, P):
    if P is None:
        return True
    k = K(curve)
    b = CURVES[curve][2]
    return k.zero(k.sub(k.mul(P[1], P[1]), k.add(k.mul(k.mul(P[0], P[0]), P[0]), b)))


def msm_naive(curve, scalars, points):
    acc = None
    for k, P in zip(scalars, points):
        acc = ec_add(curve, acc, ec_mul(curve, k, P))
    return acc


def dft_naive(field, a, omega):
    p = FIELDS[field][0]
    n = len(a)
    return [sum(a[j] * pow(omega, j * k, p) for j in range(n)) % p for k in range(n)]


# ------------------------------------------------------------------ fixture generation
def hexs(x):
    return "%x" % x


def gen_field_vectors(seed=0x5EEDC0DE):
    out = {"seed": seed, "fields": {}}
    for name in FIELD_IDS:
        p, g, nl = FIELDS[name]
        rng = Rng(seed ^ hash_name(name))
        cases = []
        specials = [0, 1, 2, p - 1, p - 2, (p + 1) // 2]
        for i in range(24):
            a = specials[i % len(specials)] if i < 6 else rng.below(p)
            b = specials[(i * 5 + 1) % len(specials)] if i < 3 else rng.below(p)
            cases.append({
                "a": hexs(a), "b": hexs(b), "add": hexs((a + b) % p), "sub": hexs((a - b) % p),
                "mul": hexs(a * b % p), "inv_a": hexs(pow(a, -1, p) if a else 0),
                "a_mont": hexs(mont(name, a)),
            })
        s, _ = two_adicity(p)
        out["fields"][name] = {
            "modulus": hexs(p), "limbs64": nl, "two_adicity": s, "generator": g,
            "root_of_unity": hexs(root_of_unity(name, s)), "cases": cases,
        }
    return out


def hash_name(name):
    h = 0xCBF29CE484222325
    for ch in name.encode():
        h = ((h ^ ch) * 0x100000001B3) & MASK64
    return h


def coord_hex(c):
    return hexs(c) if not isinstance(c, tuple) else [hexs(c[0]), hexs(c[1])]


def gen_msm_vectors(seed=0x5EEDC0DE, sizes=(1, 2, 5, 33, 64)):
    out = {"seed": seed, "curves": {}}
    for cname in CURVE_IDS:
        bf, sf, b, gx, gy = CURVES[cname]
        r = FIELDS[sf][0]
        G = (gx, gy)
        assert ec_on_curve(cname, G)
        assert ec_mul(cname, r, G) is None, "generator order"
        rng = Rng(seed ^ hash_name(cname))
        cases = []
        for n in sizes:
            ks = [rng.below(r) for _ in range(n)]
            pts = [ec_mul(cname, k, G) for k in ks]
            sc = [rng.below(r) for _ in range(n)]
            # exercise the edge digits: zero, one, r-1, a small and a duplicate point pair
            if n >= 5:
                sc[0], sc[1], sc[2], sc[3] = 0, 1, r - 1, 0xFF
                pts[4] = pts[3]
            if n >= 33:
                pts[7] = ec_neg(cname, pts[6])                       # P and -P
                sc[7] = sc[6]                                        # cancels exactly
                pts[9] = None                                        # identity base
            res = msm_naive(cname, sc, pts)
            cases.append({
                "n": n,
                "scalars": [hexs(s) for s in sc],
                "points": [None if P is None else [coord_hex(P[0]), coord_hex(P[1])] for P in pts],
                "result": None if res is None else [coord_hex(res[0]), coord_hex(res[1])],
            })
        out["curves"][cname] = {"base_field": bf, "scalar_field": sf, "b": b if not isinstance(b, tuple) else list(b),
                                "generator": [coord_hex(gx), coord_hex(gy)], "cases": cases}
    return out


def gen_ntt_vectors(seed=0x5EEDC0DE, logns=(0, 1, 2, 3, 5, 8)):
    out = {"seed": seed, "fields": {}}
    for name in ["PallasFp", "PallasFq", "Bn254Fr", "Bls381Fr"]:
        p, g, _ = FIELDS[name]
        rng = Rng(seed ^ hash_name(name) ^ 0x4E5454)
        cases = []
        for logn in logns:
            n = 1 << logn
            a = [rng.below(p) for _ in range(n)]
            w = root_of_unity(name, logn)
            fwd = dft_naive(name, a, w)
            ninv = pow(n, -1, p)
            inv = [x * ninv % p for x in dft_naive(name, a, pow(w, -1, p))]
            coset = dft_naive(name, [x * pow(g, i, p) % p for i, x in enumerate(a)], w)
            ginv = pow(g, -1, p)
            coset_inv = [x * pow(ginv, i, p) % p for i, x in enumerate(inv)]
            cases.append({"logn": logn, "omega": hexs(w), "in": [hexs(x) for x in a],
                          "fft": [hexs(x) for x in fwd], "ifft": [hexs(x) for x in inv],
                          "coset_fft": [hexs(x) for x in coset], "coset_ifft": [hexs(x) for x in coset_inv]})
        out["fields"][name] = {"modulus": hexs(p), "coset_generator": g, "cases": cases}
    return out


def main():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    os.makedirs(root, exist_ok=True)
    for fn, data in (("field_vectors.json", gen_field_vectors()),
                     ("msm_vectors.json", gen_msm_vectors()),
                     ("ntt_vectors.json", gen_ntt_vectors())):
        with open(os.path.join(root, fn), "w") as f:
            json.dump(data, f, indent=0, separators=(",", ":"))
        print("wrote", fn, os.path.getsize(os.path.join(root, fn)), "bytes")


if __name__ == "__main__":
    main()
