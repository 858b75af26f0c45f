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


def quotient_program(n_adv, n_fix, n_inst=0):
    """A gate set in the style of the reference's circuit (circuits-halo2/src/encryption.rs:83-161: ECC-style multiplication
    gates, a Pow5 S-box with a rotation, boolean / range checks behind fixed selectors, the lookup and permutation argument
    constraints), folded with y: columns [0, n_adv) advice, [n_adv, n_adv + n_fix) fixed, then A', S', Z_lookup, Z_perm x3, then
    the n_inst instance columns (equality-enabled in the reference's circuit: they enter through the last permutation chunk).
    consts: [y, 5, 1, beta, gamma]"""
    A = lambda i, r=0: ("col", i % n_adv, r)
    Fx = lambda i, r=0: ("col", n_adv + i % n_fix, r)
    X = n_adv + n_fix
    I = lambda i: ("col", X + 6 + i % n_inst, 0)
    ap, sp, zl = ("col", X, 0), ("col", X + 1, 0), ("col", X + 2, 0)
    zp = [("col", X + 3 + c, 0) for c in range(3)]
    prog = []
    # every term after the first is folded in as  acc = acc * y + term
    # multiplication gates  q (a b - c), eight of them
    first = True
    for g in range(8):
        term = [Fx(g), A(g), A(g + 1), ("mul",), A(g + 2), ("sub",), ("mul",)]
        prog.extend(term if first else [("scale", 0)] + term + [("add",)])
        first = False
    # Pow5 with a rotation  q (a^5 + 5 - b(omega X)), three of them
    for g in range(3):
        a = A(3 * g + 1)
        prog.extend([("scale", 0), Fx(g + 3), a, a, ("mul",), a, ("mul",), a, ("mul",), a, ("mul",), ("const", 1), ("add",),
                     ("col", (3 * g + 2) % n_adv, 1), ("sub",), ("mul",), ("add",)])
    # boolean checks  q a (a - 1), four of them
    for g in range(4):
        a = A(g + 9)
        prog.extend([("scale", 0), Fx(g + 4), a, a, ("const", 2), ("sub",), ("mul",), ("mul",), ("add",)])
    # lookup: Z(omega X)(A' + beta)(S' + gamma) - Z(X)(A + beta)(S + gamma), and (A' - S')(A' - A'(omega^-1 X))
    prog.extend([("scale", 0), ("col", X + 2, 1), ap, ("const", 3), ("add",), ("mul",), sp, ("const", 4), ("add",), ("mul",),
                 zl, A(0), ("const", 3), ("add",), ("mul",), Fx(7), ("const", 4), ("add",), ("mul",), ("sub",), ("add",)])
    prog.extend([("scale", 0), ap, sp, ("sub",), ap, ("col", X, -1), ("sub",), ("mul",), ("add",)])
    # permutation, per chunk: Z(omega X) prod (v + beta s + gamma) - Z(X) prod (v + delta-term + gamma), two columns of each chunk written out
    for c in range(3):
        v0, v1 = (I(0), I(1)) if (n_inst and c == 2) else (A(2 * c), A(2 * c + 1))
        prog.extend([("scale", 0), ("col", X + 3 + c, 1), v0, Fx(c), ("scale", 3), ("add",), ("const", 4), ("add",), ("mul",),
                     v1, Fx(c + 1), ("scale", 3), ("add",), ("const", 4), ("add",), ("mul",),
                     zp[c], v0, ("const", 4), ("add",), ("mul",), v1, ("const", 3), ("add",), ("mul",), ("sub",), ("add",)])
    return prog


def rotated_advice(n_adv):
    """the advice columns quotient_program reads at the next row (the Pow5 gates): they are opened at x AND omega x"""
    return sorted({(3 * g + 2) % n_adv for g in range(3)})
