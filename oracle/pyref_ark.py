"""TEST INFRASTRUCTURE -- pure-Python restatement of the ark-serialize 0.3 wire formats the reference reads and writes
(SURVEY 8f f3): short-Weierstrass points compressed / uncompressed, Vec<T>, ark-groth16 0.3 VerifyingKey / ProvingKey /
Proof; plus the reference's own JSON envelope.

Reference call sites (the formats themselves live in un-vendored crates -- PARITY UNPINNED, like the rest of oracle/):
  lib/src/utils.rs:85-102     write_circuit_artifacts: pk.serialize_unchecked (uncompressed, no checks) ; ark_to_bytes(vk) (compressed)
  lib/src/utils.rs:104-118    read_proving_key: ProvingKey::deserialize_unchecked ; read_verifying_key: ark_from_bytes
  circuits-ark/src/utils.rs:12-22   ark_from_bytes / ark_to_bytes = CanonicalDeserialize::deserialize / CanonicalSerialize::serialize
  lib/src/zk/verifiable_encryption.rs:23-34   VerifiableEncryption { ciphertext, proof_of_encryption, proofs_of_property } (serde_json)

ark-serialize 0.3 rules restated here:
  * Fp: canonical (non-Montgomery) integer, little-endian, ceil((MODULUS_BITS + flag bits) / 8) bytes; flags are OR-ed into
    the top bits of the LAST byte.  Fp2: c0 then c1, flags on c1.
  * SWFlags (2 bits): bit 7 = "y is the larger of (y, -y)", bit 6 = infinity; both set is invalid.
  * GroupAffine::serialize (compressed): x with flags; infinity = zero x with the infinity flag.
    serialize_uncompressed / serialize_unchecked: x plain, then y with flags (only the infinity flag is ever set);
    infinity is written as x = 0, y = 1 (GroupAffine::zero()) with the flag.
  * "larger": Fp compares canonical integers; Fp2 compares c1 first, then c0.
  * Vec<T>: u64 length (8 bytes LE) then the items.
  * ark-groth16 0.3 field order: VerifyingKey {alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1: Vec};
    ProvingKey {vk, beta_g1, delta_g1, a_query, b_g1_query, b_g2_query: Vec<G2>, h_query, l_query}; Proof {a: G1, b: G2, c: G1}.
`python oracle/pyref_ark.py` regenerates tests/golden/wire_vectors.json.
"""
import json
import os

try:
    from . import pyref
except ImportError:   # run as a script
    import pyref

PAIRINGS = {"Bn254": ("Bn254G1", "Bn254G2"), "Bls381": ("Bls381G1", "Bls381G2")}


def fq_bytes(curve):
    p = pyref.FIELDS[pyref.CURVES[curve][0]][0]
    return (p.bit_length() + 2 + 7) // 8        # with the two SWFlags bits; equals the flag-less size for both families


def _fp_to_bytes(v, nb, flags=0):
    b = bytearray(v.to_bytes(nb, "little"))
    b[-1] |= flags
    return bytes(b)


def _coord_to_bytes(curve, c, flags=0):
    nb = fq_bytes(curve)
    if pyref.is_g2(curve):
        return _fp_to_bytes(c[0], nb) + _fp_to_bytes(c[1], nb, flags)
    return _fp_to_bytes(c, nb, flags)


def _is_larger(curve, y):
    """y > -y in ark-ff's ordering"""
    k = pyref.K(curve)
    ny = k.neg(y)
    if pyref.is_g2(curve):
        return (y[1], y[0]) > (ny[1], ny[0])
    return y > ny


def point_size(curve, compressed):
    return fq_bytes(curve) * (2 if pyref.is_g2(curve) else 1) * (1 if compressed else 2)


def encode_point(curve, P, compressed):
    zero = (0, 0) if pyref.is_g2(curve) else 0
    one = (1, 0) if pyref.is_g2(curve) else 1
    if compressed:
        if P is None:
            return _coord_to_bytes(curve, zero, 1 << 6)
        return _coord_to_bytes(curve, P[0], (1 << 7) if _is_larger(curve, P[1]) else 0)
    if P is None:
        return _coord_to_bytes(curve, zero) + _coord_to_bytes(curve, one, 1 << 6)
    return _coord_to_bytes(curve, P[0]) + _coord_to_bytes(curve, P[1])


def _fp_sqrt(a, p):
    assert p % 4 == 3
    r = pow(a, (p + 1) // 4, p)
    return r if r * r % p == a % p else None


def _coord_sqrt(curve, a):
    p = pyref.FIELDS[pyref.CURVES[curve][0]][0]
    if not pyref.is_g2(curve):
        return _fp_sqrt(a, p)
    a0, a1 = a
    if a1 == 0:
        r = _fp_sqrt(a0, p)
        if r is not None:
            return (r, 0)
        r = _fp_sqrt((-a0) % p, p)
        return None if r is None else (0, r)
    alpha = _fp_sqrt((a0 * a0 + a1 * a1) % p, p)
    if alpha is None:
        return None
    inv2 = pow(2, -1, p)
    delta = (a0 + alpha) * inv2 % p
    c0 = _fp_sqrt(delta, p)
    if c0 is None:
        delta = (a0 - alpha) * inv2 % p
        c0 = _fp_sqrt(delta, p)
        if c0 is None:
            return None
    c1 = a1 * pow(2 * c0, -1, p) % p
    return (c0, c1)


def _read_coord(curve, buf, off, with_flags):
    nb = fq_bytes(curve)
    p = pyref.FIELDS[pyref.CURVES[curve][0]][0]
    parts = []
    ncomp = 2 if pyref.is_g2(curve) else 1
    flags = 0
    for i in range(ncomp):
        raw = bytearray(buf[off + i * nb: off + (i + 1) * nb])
        if with_flags and i == ncomp - 1:
            flags = raw[-1] & 0xC0
            raw[-1] &= 0x3F
        v = int.from_bytes(bytes(raw), "little")
        if v >= p:
            raise ValueError("non-canonical field element")
        parts.append(v)
    return (tuple(parts) if ncomp == 2 else parts[0]), flags, off + ncomp * nb


def decode_point(curve, buf, off, compressed):
    """-> (point or None, new offset).  Compressed: recovers y (ValueError if x is not on the curve).  Uncompressed:
    deserialize_unchecked semantics -- no curve check."""
    k = pyref.K(curve)
    if compressed:
        x, flags, off = _read_coord(curve, buf, off, True)
        if flags == 0xC0:
            raise ValueError("invalid flags")
        if flags & 0x40:
            return None, off
        b = pyref.CURVES[curve][2]
        y = _coord_sqrt(curve, k.add(k.mul(k.mul(x, x), x), b))
        if y is None:
            raise ValueError("x is not on the curve")
        if _is_larger(curve, y) != bool(flags & 0x80):
            y = k.neg(y)
        return (x, y), off
    x, _, off = _read_coord(curve, buf, off, False)
    y, flags, off = _read_coord(curve, buf, off, True)
    if flags == 0xC0:
        raise ValueError("invalid flags")
    return (None if flags & 0x40 else (x, y)), off


def encode_vec(curve, pts, compressed):
    return len(pts).to_bytes(8, "little") + b"".join(encode_point(curve, P, compressed) for P in pts)


def decode_vec(curve, buf, off, compressed):
    n = int.from_bytes(buf[off:off + 8], "little")
    off += 8
    out = []
    for _ in range(n):
        P, off = decode_point(curve, buf, off, compressed)
        out.append(P)
    return out, off


VK_FIELDS = [("alpha_g1", 1, False), ("beta_g2", 2, False), ("gamma_g2", 2, False), ("delta_g2", 2, False), ("gamma_abc_g1", 1, True)]
PK_FIELDS = [("beta_g1", 1, False), ("delta_g1", 1, False), ("a_query", 1, True), ("b_g1_query", 1, True), ("b_g2_query", 2, True),
             ("h_query", 1, True), ("l_query", 1, True)]


def _encode_fields(pairing, obj, fields, compressed):
    g = PAIRINGS[pairing]
    out = b""
    for name, grp, is_vec in fields:
        c = g[grp - 1]
        out += encode_vec(c, obj[name], compressed) if is_vec else encode_point(c, obj[name], compressed)
    return out


def _decode_fields(pairing, buf, off, fields, compressed):
    g = PAIRINGS[pairing]
    obj = {}
    for name, grp, is_vec in fields:
        c = g[grp - 1]
        obj[name], off = decode_vec(c, buf, off, compressed) if is_vec else decode_point(c, buf, off, compressed)
    return obj, off


def encode_vk(pairing, vk, compressed=True):
    return _encode_fields(pairing, vk, VK_FIELDS, compressed)


def decode_vk(pairing, buf, compressed=True):
    vk, off = _decode_fields(pairing, buf, 0, VK_FIELDS, compressed)
    assert off == len(buf)
    return vk


def encode_pk_unchecked(pairing, pk):
    """ProvingKey::serialize_unchecked (lib/src/utils.rs:90-91): everything uncompressed"""
    return _encode_fields(pairing, pk["vk"], VK_FIELDS, False) + _encode_fields(pairing, pk, PK_FIELDS, False)


def decode_pk_unchecked(pairing, buf):
    vk, off = _decode_fields(pairing, buf, 0, VK_FIELDS, False)
    pk, off = _decode_fields(pairing, buf, off, PK_FIELDS, False)
    assert off == len(buf)
    pk["vk"] = vk
    return pk


def encode_proof(pairing, a, b, c):
    """ark_to_bytes(proof): compressed a (G1), b (G2), c (G1) -- 48 + 96 + 48 bytes on BLS12-381"""
    g1, g2 = PAIRINGS[pairing]
    return encode_point(g1, a, True) + encode_point(g2, b, True) + encode_point(g1, c, True)


def decode_proof(pairing, buf):
    g1, g2 = PAIRINGS[pairing]
    a, off = decode_point(g1, buf, 0, True)
    b, off = decode_point(g2, buf, off, True)
    c, off = decode_point(g1, buf, off, True)
    assert off == len(buf)
    return a, b, c


def encode_fr(field, v):
    """Fr::serialize: canonical little-endian, 32 bytes"""
    return v.to_bytes((pyref.FIELDS[field][0].bit_length() + 7) // 8, "little")


def verifiable_encryption_json(ciphertext, proof_of_encryption, proofs_of_property):
    """serde_json of lib/src/zk/verifiable_encryption.rs:23-34: Vec<u8> fields are JSON arrays of numbers, the
    (String, Vec<u8>) argument pairs are two-element arrays"""
    return json.dumps({"ciphertext": list(ciphertext), "proof_of_encryption": list(proof_of_encryption),
                       "proofs_of_property": [{"proof": list(p), "arguments": [[n, list(v)] for n, v in args]}
                                              for p, args in proofs_of_property]}, separators=(",", ":"))


# ------------------------------------------------------------------ fixtures
def _rand_points(curve, rng, n, with_inf):
    r = pyref.FIELDS[pyref.CURVES[curve][1]][0]
    G = (pyref.CURVES[curve][3], pyref.CURVES[curve][4])
    pts = [pyref.ec_mul(curve, rng.below(r), G) for _ in range(n)]
    if with_inf and n >= 3:
        pts[1] = None
    return pts


def synth_pk(pairing, seed, n_a=5, n_l=3, n_pub=2):
    """a structurally valid (not cryptographically meaningful) key: random subgroup points in every slot"""
    g1, g2 = PAIRINGS[pairing]
    rng = pyref.Rng(seed ^ pyref.hash_name(pairing))
    vk = {"alpha_g1": _rand_points(g1, rng, 1, False)[0], "beta_g2": _rand_points(g2, rng, 1, False)[0],
          "gamma_g2": _rand_points(g2, rng, 1, False)[0], "delta_g2": _rand_points(g2, rng, 1, False)[0],
          "gamma_abc_g1": _rand_points(g1, rng, n_pub + 1, False)}
    pk = {"vk": vk, "beta_g1": _rand_points(g1, rng, 1, False)[0], "delta_g1": _rand_points(g1, rng, 1, False)[0],
          "a_query": _rand_points(g1, rng, n_a, True), "b_g1_query": _rand_points(g1, rng, n_a, True),
          "b_g2_query": _rand_points(g2, rng, n_a, True), "h_query": _rand_points(g1, rng, n_a - 1, False),
          "l_query": _rand_points(g1, rng, n_l, False)}
    return pk


def _pt_hex(P):
    return None if P is None else [pyref.coord_hex(P[0]), pyref.coord_hex(P[1])]


def _obj_hex(o):
    if isinstance(o, dict):
        return {k: _obj_hex(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_pt_hex(P) for P in o]
    return _pt_hex(o)


def gen_wire_vectors(seed=0x5EEDC0DE):
    out = {"seed": seed, "pairings": {}}
    for pairing in ("Bls381", "Bn254"):
        g1, g2 = PAIRINGS[pairing]
        pk = synth_pk(pairing, seed)
        pk_bytes = encode_pk_unchecked(pairing, pk)
        assert decode_pk_unchecked(pairing, pk_bytes) == pk
        vk_bytes = encode_vk(pairing, pk["vk"])
        assert decode_vk(pairing, vk_bytes) == pk["vk"]
        rng = pyref.Rng(seed + 7)
        a, c = _rand_points(g1, rng, 2, False)
        b = _rand_points(g2, rng, 1, False)[0]
        proof = encode_proof(pairing, a, b, c)
        assert decode_proof(pairing, proof) == (a, b, c)
        pts = {}
        for curve in (g1, g2):
            cases = []
            ps = _rand_points(curve, pyref.Rng(seed + 11), 6, False) + [None]
            ps.append(pyref.ec_neg(curve, ps[0]))       # the other y for the same x: the sign flag must flip
            for P in ps:
                cases.append({"point": _pt_hex(P), "compressed": encode_point(curve, P, True).hex(),
                              "uncompressed": encode_point(curve, P, False).hex()})
            pts[curve] = cases
        out["pairings"][pairing] = {"proving_key": _obj_hex(pk), "proving_key_unchecked_bytes": pk_bytes.hex(),
                                    "verifying_key_bytes": vk_bytes.hex(), "proof": [_pt_hex(a), _pt_hex(b), _pt_hex(c)],
                                    "proof_bytes": proof.hex(), "points": pts}
    out["verifiable_encryption_json"] = verifiable_encryption_json(b"\x01\x02\xff", b"\x00\x10", [(b"\x07", [("leaf", b"\x05\x06")])])
    return out


def main():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    data = gen_wire_vectors()
    fn = os.path.join(root, "wire_vectors.json")
    with open(fn, "w") as f:
        json.dump(data, f, indent=0, separators=(",", ":"))
    print("wrote wire_vectors.json", os.path.getsize(fn), "bytes")


if __name__ == "__main__":
    main()
