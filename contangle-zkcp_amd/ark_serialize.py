"""Mirror of the ark-serialize 0.3 surface the reference uses for keys and proofs, over the C ABI codecs of
include/zkcp_amd_prover.h:

  ark_to_bytes / ark_from_bytes            circuits-ark/src/utils.rs:12-22  (CanonicalSerialize::serialize = compressed)
  ProvingKey.deserialize_unchecked / serialize_unchecked    lib/src/utils.rs:85-110 (the key file `compile` writes, `sell` reads)
  VerifyingKey (compressed)                lib/src/utils.rs:93,112-118
  Proof (compressed a, b, c)               lib/src/zk/encryption.rs:80 `ark_to_bytes(proof)`
  VerifiableEncryption JSON                lib/src/zk/verifiable_encryption.rs:23-34, cipher_host.rs:25-42 (serde_json)

Points are numpy uint64 arrays of Montgomery limbs, [n, 2 * limbs] with infinity = all zero -- the layout zk.Bases takes.
"""
import ctypes
import json

import numpy as np

from . import Bases, _check, _np64, _ptr, base_limbs, curve_id, load

BN254, BLS12_381 = 0, 1
PAIRING_NAMES = {"Bn254": BN254, "Bls381": BLS12_381}
PAIRING_CURVES = {BN254: ("Bn254G1", "Bn254G2"), BLS12_381: ("Bls381G1", "Bls381G2")}

PROVER_EXPORTS = ["zk_ark_point_size", "zk_ark_points_encode", "zk_ark_points_decode", "zk_ark_scalars_encode", "zk_ark_scalars_decode",
                  "zk_ark_proving_key_index", "zk_bases_upload_ark", "zk_ark_proof_size", "zk_ark_proof_encode", "zk_ark_proof_decode"]


class Span(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint64), ("count", ctypes.c_uint64)]


PK_MEMBERS = ["alpha_g1", "beta_g2", "gamma_g2", "delta_g2", "gamma_abc_g1", "beta_g1", "delta_g1", "a_query", "b_g1_query", "b_g2_query",
              "h_query", "l_query"]
G2_MEMBERS = {"beta_g2", "gamma_g2", "delta_g2", "b_g2_query"}
VEC_MEMBERS = {"gamma_abc_g1", "a_query", "b_g1_query", "b_g2_query", "h_query", "l_query"}


class PkIndex(ctypes.Structure):
    _fields_ = [(m, Span) for m in PK_MEMBERS] + [("total_bytes", ctypes.c_uint64)]


def pairing_id(p):
    return PAIRING_NAMES[p] if isinstance(p, str) else p


def _lib():
    lib = load()
    u8p, u64, vp, i32 = ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int
    lib.zk_ark_points_encode.argtypes = [i32, vp, u64, i32, vp]
    lib.zk_ark_points_decode.argtypes = [i32, u8p, u64, i32, i32, vp]
    lib.zk_ark_scalars_encode.argtypes = [i32, vp, u64, vp]
    lib.zk_ark_scalars_decode.argtypes = [i32, u8p, u64, vp]
    lib.zk_ark_proving_key_index.argtypes = [i32, u8p, u64, ctypes.POINTER(PkIndex)]
    lib.zk_bases_upload_ark.argtypes = [i32, u8p, u64, ctypes.POINTER(u64)]
    lib.zk_ark_proof_encode.argtypes = [i32, vp, vp, vp, vp]
    lib.zk_ark_proof_decode.argtypes = [i32, u8p, vp, vp, vp]
    return lib


def point_size(curve, compressed=True):
    return _lib().zk_ark_point_size(curve_id(curve), int(compressed))


def points_to_bytes(curve, pts, compressed=True):
    """GroupAffine::serialize (compressed) / serialize_uncompressed for each point, concatenated (no length prefix)"""
    pts = _np64(pts).reshape(-1, 2 * base_limbs(curve))
    out = np.zeros(pts.shape[0] * point_size(curve, compressed), dtype=np.uint8)
    _check(_lib().zk_ark_points_encode(curve_id(curve), _ptr(pts), pts.shape[0], int(compressed), _ptr(out)), "zk_ark_points_encode")
    return out.tobytes()


def points_from_bytes(curve, buf, n, compressed=True, check_on_curve=False):
    out = np.zeros((n, 2 * base_limbs(curve)), dtype=np.uint64)
    _check(_lib().zk_ark_points_decode(curve_id(curve), bytes(buf), n, int(compressed), int(check_on_curve), _ptr(out)), "zk_ark_points_decode")
    return out


def vec_to_bytes(curve, pts, compressed=True):
    """Vec<GroupAffine>: u64 length, then the items"""
    pts = _np64(pts).reshape(-1, 2 * base_limbs(curve))
    return int(pts.shape[0]).to_bytes(8, "little") + points_to_bytes(curve, pts, compressed)


def scalars_to_bytes(field, a):
    from . import field_id
    a = _np64(a).reshape(-1, 4)
    out = np.zeros(a.shape[0] * 32, dtype=np.uint8)
    _check(_lib().zk_ark_scalars_encode(field_id(field), _ptr(a), a.shape[0], _ptr(out)), "zk_ark_scalars_encode")
    return out.tobytes()


def scalars_from_bytes(field, buf, n):
    from . import field_id
    out = np.zeros((n, 4), dtype=np.uint64)
    _check(_lib().zk_ark_scalars_decode(field_id(field), bytes(buf), n, _ptr(out)), "zk_ark_scalars_decode")
    return out


class ProvingKey:
    """ark-groth16 0.3 ProvingKey<E>.  `deserialize_unchecked(buf)` indexes the key file; `.points(name)` decodes a member
    to host limbs; `.upload(name)` puts a query vector straight into a resident `Bases` handle (uploaded once)."""

    def __init__(self, pairing, buf, index):
        self.pairing, self.buf, self.index = pairing_id(pairing), bytes(buf), index

    @classmethod
    def deserialize_unchecked(cls, pairing, buf):
        idx = PkIndex()
        _check(_lib().zk_ark_proving_key_index(pairing_id(pairing), bytes(buf), len(buf), ctypes.byref(idx)), "zk_ark_proving_key_index")
        if idx.total_bytes != len(buf):
            raise ValueError("trailing bytes after the proving key")
        return cls(pairing, buf, idx)

    def _curve(self, name):
        return PAIRING_CURVES[self.pairing][1 if name in G2_MEMBERS else 0]

    def count(self, name):
        return int(getattr(self.index, name).count)

    def points(self, name):
        sp = getattr(self.index, name)
        c = self._curve(name)
        n = int(sp.count)
        return points_from_bytes(c, self.buf[sp.offset: sp.offset + n * point_size(c, False)], n, compressed=False)

    def upload(self, name, skip_first=0):
        """-> Bases over member `name`; skip_first = 1 drops the first point (ark-groth16 uses a_query[1..] etc. for the
        witness part and handles the constant-one wire separately)"""
        sp = getattr(self.index, name)
        c = self._curve(name)
        ps = point_size(c, False)
        n = int(sp.count) - skip_first
        h = ctypes.c_uint64(0)
        start = sp.offset + skip_first * ps
        _check(_lib().zk_bases_upload_ark(curve_id(c), self.buf[start: start + n * ps], n, ctypes.byref(h)), "zk_bases_upload_ark")
        b = Bases.__new__(Bases)
        b.curve, b.n, b.handle, b._keep = curve_id(c), n, h.value, None
        return b

    @staticmethod
    def serialize_unchecked(pairing, members):
        """members: dict name -> points (single members as [1, 2L] arrays); the bytes `pk.serialize_unchecked` writes"""
        pairing = pairing_id(pairing)
        out = b""
        for name in PK_MEMBERS:
            c = PAIRING_CURVES[pairing][1 if name in G2_MEMBERS else 0]
            out += vec_to_bytes(c, members[name], False) if name in VEC_MEMBERS else points_to_bytes(c, members[name], False)
        return out


VK_MEMBERS = PK_MEMBERS[:5]


def verifying_key_to_bytes(pairing, members):
    """ark_to_bytes(vk): compressed"""
    pairing = pairing_id(pairing)
    out = b""
    for name in VK_MEMBERS:
        c = PAIRING_CURVES[pairing][1 if name in G2_MEMBERS else 0]
        out += vec_to_bytes(c, members[name], True) if name in VEC_MEMBERS else points_to_bytes(c, members[name], True)
    return out


def verifying_key_from_bytes(pairing, buf):
    pairing = pairing_id(pairing)
    off, out = 0, {}
    for name in VK_MEMBERS:
        c = PAIRING_CURVES[pairing][1 if name in G2_MEMBERS else 0]
        n = 1
        if name in VEC_MEMBERS:
            n = int.from_bytes(buf[off:off + 8], "little")
            off += 8
        ps = point_size(c, True)
        out[name] = points_from_bytes(c, buf[off: off + n * ps], n, compressed=True)
        off += n * ps
    if off != len(buf):
        raise ValueError("trailing bytes after the verifying key")
    return out


def proof_to_bytes(pairing, a, b, c):
    """ark_to_bytes(Proof { a, b, c }): 192 bytes on BLS12-381"""
    pairing = pairing_id(pairing)
    lib = _lib()
    out = np.zeros(lib.zk_ark_proof_size(pairing), dtype=np.uint8)
    _check(lib.zk_ark_proof_encode(pairing, _ptr(_np64(a)), _ptr(_np64(b)), _ptr(_np64(c)), _ptr(out)), "zk_ark_proof_encode")
    return out.tobytes()


def proof_from_bytes(pairing, buf):
    pairing = pairing_id(pairing)
    g1, g2 = PAIRING_CURVES[pairing]
    a, b, c = (np.zeros(2 * base_limbs(g1), dtype=np.uint64), np.zeros(2 * base_limbs(g2), dtype=np.uint64),
               np.zeros(2 * base_limbs(g1), dtype=np.uint64))
    if len(buf) != _lib().zk_ark_proof_size(pairing):
        raise ValueError("wrong proof length")
    _check(_lib().zk_ark_proof_decode(pairing, bytes(buf), _ptr(a), _ptr(b), _ptr(c)), "zk_ark_proof_decode")
    return a, b, c


class VerifiableEncryption:
    """lib/src/zk/verifiable_encryption.rs:23-34 -- the JSON the seller hosts (cipher_host.rs:25-42): byte vectors are JSON
    arrays of numbers, `arguments` a list of [name, bytes] pairs (serde's encoding of Vec<(String, Vec<u8>)>)"""

    def __init__(self, ciphertext, proof_of_encryption, proofs_of_property=()):
        self.ciphertext, self.proof_of_encryption = bytes(ciphertext), bytes(proof_of_encryption)
        self.proofs_of_property = [(bytes(p), [(n, bytes(v)) for n, v in args]) for p, args in proofs_of_property]

    def to_json(self):
        return json.dumps({"ciphertext": list(self.ciphertext), "proof_of_encryption": list(self.proof_of_encryption),
                           "proofs_of_property": [{"proof": list(p), "arguments": [[n, list(v)] for n, v in args]}
                                                  for p, args in self.proofs_of_property]}, separators=(",", ":"))

    @classmethod
    def from_json(cls, s):
        d = json.loads(s)
        return cls(bytes(d["ciphertext"]), bytes(d["proof_of_encryption"]),
                   [(bytes(p["proof"]), [(n, bytes(v)) for n, v in p["arguments"]]) for p in d["proofs_of_property"]])
