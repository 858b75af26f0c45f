"""Seeded synthetic inputs of the benchmarks and tests: uniform field elements / scalars from a vectorised
splitmix64 stream with rejection sampling, and the 0/1-heavy "realistic witness" mix.  Product-side code: bench.py and the
tools use it directly, tests import it from here."""
import numpy as np

from . import FIELD_NAMES, field_modulus

# scalar field of each curve (zk_curve_scalar_field)
CURVE_SCALAR_FIELD = {"Pallas": "PallasFq", "Vesta": "PallasFp", "Bn254G1": "Bn254Fr", "Bn254G2": "Bn254Fr",
                      "Bls381G1": "Bls381Fr", "Bls381G2": "Bls381Fr"}
_moduli = {}


def modulus(field):
    if field not in _moduli:
        assert field in FIELD_NAMES, field
        _moduli[field] = field_modulus(field)
    return _moduli[field]


def rand_field(name, n, seed):
    """n uniform elements of the field (as stored words: read them as Montgomery residues or as canonical integers),
    uint64 [n, 4]"""
    p = modulus(name)
    out = np.zeros((n, 4), dtype=np.uint64)
    todo = np.arange(n)
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
    sf = CURVE_SCALAR_FIELD[curve]
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
