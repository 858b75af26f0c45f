"""Mirror of ark-groth16 0.3 `create_proof` around the device path (call sites
lib/src/zk/encryption.rs:76, verifiable_encryption.rs:92, sample_entries.rs:86, property.rs:133):

  R1csMatrix                 ark-relations 0.3 ConstraintMatrices rows, resident in CSR form (fixed per circuit)
  witness_map                r1cs_to_qap.rs R1CStoQAP::witness_map from the full assignment: 3 sparse mat-vecs + 7 NTTs + glue
  Prover.prove(z, r, s)      prover.rs create_proof_with_reduction_and_matrices: witness map -> h ; the five MSMs
                             (submitted back to back, collected afterwards) ; assembly of A, B, C ; ark_to_bytes(proof)

The blinding scalars r, s are arguments: upstream draws them from the caller's RNG, so a proof is reproducible bit for
bit only when the unmodified Rust prover drives the FFI (SURVEY 7 "hard parts"); everything before them is deterministic.
"""
import ctypes

import numpy as np

from . import (_check, _np64, _ptr, ark_serialize, base_limbs, field_id, load, msm_submit, vec_op)

PROVER_EXPORTS = ["zk_r1cs_matrix_upload", "zk_r1cs_matrix_free", "zk_r1cs_matvec_device", "zk_groth16_witness_map_r1cs_device",
                  "zk_groth16_assemble_proof"]


class Assembly(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "a_query0", "b_g1_query0",
                                                "b_g2_query0", "a_acc", "b_g1_acc", "l_acc", "h_acc", "b_g2_acc", "r", "s")]


def _lib():
    lib = load()
    u64, vp, i32 = ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int
    lib.zk_r1cs_matrix_upload.argtypes = [i32, vp, vp, vp, u64, u64, ctypes.POINTER(u64)]
    lib.zk_r1cs_matrix_free.argtypes = [u64]
    lib.zk_r1cs_matvec_device.argtypes = [u64, vp, vp, u64, vp]
    lib.zk_groth16_witness_map_r1cs_device.argtypes = [i32, u64, u64, u64, vp, u64, ctypes.c_uint32, vp, vp, vp, vp]
    lib.zk_groth16_assemble_proof.argtypes = [i32, ctypes.POINTER(Assembly), vp, vp, vp]
    return lib


class R1csMatrix:
    """rows: list of rows, a row = list of (coefficient as Montgomery u64[4], variable index)  -- or CSR arrays directly"""

    def __init__(self, field, rows=None, n_cols=None, csr=None):
        self.field = field_id(field)
        if csr is None:
            row_ptr = np.zeros(len(rows) + 1, dtype=np.uint64)
            for i, row in enumerate(rows):
                row_ptr[i + 1] = row_ptr[i] + len(row)
            nnz = int(row_ptr[-1])
            col = np.zeros(max(nnz, 1), dtype=np.uint32)
            val = np.zeros((max(nnz, 1), 4), dtype=np.uint64)
            k = 0
            for row in rows:
                for c, j in row:
                    val[k] = c
                    col[k] = j
                    k += 1
        else:
            row_ptr, col, val = (np.ascontiguousarray(csr[0], dtype=np.uint64), np.ascontiguousarray(csr[1], dtype=np.uint32),
                                 _np64(csr[2]))
        self.n_rows, self.n_cols = int(row_ptr.shape[0]) - 1, int(n_cols)
        h = ctypes.c_uint64(0)
        _check(_lib().zk_r1cs_matrix_upload(self.field, _ptr(row_ptr), _ptr(col), _ptr(val), self.n_rows, self.n_cols, ctypes.byref(h)),
               "zk_r1cs_matrix_upload")
        self.handle = h.value

    def matvec(self, d_z, d_out, stream=0):
        _check(_lib().zk_r1cs_matvec_device(self.handle, _ptr(d_z), _ptr(d_out), int(d_out.shape[0]), ctypes.c_void_p(stream)),
               "zk_r1cs_matvec_device")
        return d_out

    def free(self):
        if self.handle:
            _check(_lib().zk_r1cs_matrix_free(self.handle), "zk_r1cs_matrix_free")
            self.handle = 0


def witness_map(field, A, B, C, d_z, num_inputs, d_a, d_b, d_c, stream=0):
    """h = R1CStoQAP::witness_map(z) in HBM; d_a / d_b / d_c: device buffers of the domain size m (a power of two >=
    num_constraints + num_inputs); on return d_a holds the m coefficients of h (Montgomery), the last one zero"""
    m = int(d_a.shape[0])
    log_m = m.bit_length() - 1
    assert m == 1 << log_m and int(d_b.shape[0]) == m and int(d_c.shape[0]) == m
    _check(_lib().zk_groth16_witness_map_r1cs_device(field_id(field), A.handle, B.handle, C.handle, _ptr(d_z), num_inputs, log_m,
                                                     _ptr(d_a), _ptr(d_b), _ptr(d_c), ctypes.c_void_p(stream)),
           "zk_groth16_witness_map_r1cs_device")
    return d_a


def assemble_proof(pairing, key_points, accs, r, s):
    """key_points: alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2, a_query0, b_g1_query0, b_g2_query0 (affine Montgomery limbs);
    accs: a_acc, b_g1_acc, l_acc, h_acc, b_g2_acc (Jacobian MSM results); r, s: Fr Montgomery limbs -> (A, B, C) affine"""
    pairing = ark_serialize.pairing_id(pairing)
    g1, g2 = ark_serialize.PAIRING_CURVES[pairing]
    keep = {k: _np64(v).ravel() for k, v in list(key_points.items()) + list(accs.items()) + [("r", r), ("s", s)]}
    asm = Assembly(**{k: ctypes.cast(_ptr(keep[k]), ctypes.c_void_p) for k, _ in Assembly._fields_})
    a, b, c = (np.zeros(2 * base_limbs(g1), dtype=np.uint64), np.zeros(2 * base_limbs(g2), dtype=np.uint64),
               np.zeros(2 * base_limbs(g1), dtype=np.uint64))
    _check(_lib().zk_groth16_assemble_proof(pairing, ctypes.byref(asm), _ptr(a), _ptr(b), _ptr(c)), "zk_groth16_assemble_proof")
    return a, b, c


class Prover:
    """One circuit: its proving key (query vectors resident as Bases, single elements on the host) and its R1CS matrices.
    `prove` is ark-groth16 0.3 create_proof from the full assignment on."""

    def __init__(self, pairing, pk, A, B, C, num_inputs, to_device):
        """pk: ark_serialize.ProvingKey (e.g. deserialize_unchecked of the file the reference's `compile` wrote)"""
        self.pairing = ark_serialize.pairing_id(pairing)
        self.field = "Bls381Fr" if self.pairing == ark_serialize.BLS12_381 else "Bn254Fr"
        self.A, self.B, self.C, self.num_inputs, self.to_device = A, B, C, num_inputs, to_device
        self.h_query, self.l_query = pk.upload("h_query"), pk.upload("l_query")
        self.a_query, self.b_g1_query, self.b_g2_query = (pk.upload("a_query", 1), pk.upload("b_g1_query", 1), pk.upload("b_g2_query", 1))
        self.points = {"alpha_g1": pk.points("alpha_g1")[0], "beta_g1": pk.points("beta_g1")[0], "delta_g1": pk.points("delta_g1")[0],
                       "beta_g2": pk.points("beta_g2")[0], "delta_g2": pk.points("delta_g2")[0], "a_query0": pk.points("a_query")[0],
                       "b_g1_query0": pk.points("b_g1_query")[0], "b_g2_query0": pk.points("b_g2_query")[0]}
        m = 1
        while m < A.n_rows + num_inputs:
            m *= 2
        self.m = m
        assert self.h_query.n == m - 1

    def prove(self, z_mont, r_mont, s_mont, stream=0):
        """z_mont: full assignment [num_vars, 4] Montgomery limbs, z[0] = 1 -> (A, B, C) affine points and the proof bytes"""
        z_mont = _np64(z_mont)
        ni, m = self.num_inputs, self.m
        d_z = self.to_device(z_mont)
        d_abc = [self.to_device(np.zeros((m, 4), dtype=np.uint64)) for _ in range(3)]
        d_h = witness_map(self.field, self.A, self.B, self.C, d_z, ni, d_abc[0], d_abc[1], d_abc[2], stream=stream)
        d_zc = self.to_device(z_mont)
        vec_op(self.field, "into_repr", d_zc, stream=stream)             # the MSMs over z take canonical BigInts, like upstream
        # at most four MSMs in flight per device: submit four, collect one, submit the fifth.  The G2 MSM goes first: its host
        # tail (the Horner over the windows on Fq2 host limbs) is the longest of the five and then runs beside the G1 MSMs' device work
        t_b2 = msm_submit(self.b_g2_query, d_zc[1:], stream=stream, own_stream=True)
        t_h = msm_submit(self.h_query, d_h[:m - 1], montgomery=True, stream=stream, own_stream=True)
        t_l = msm_submit(self.l_query, d_zc[ni:], stream=stream, own_stream=True)
        t_a = msm_submit(self.a_query, d_zc[1:], stream=stream, own_stream=True)
        accs = {"b_g2_acc": t_b2.collect()}
        t_b1 = msm_submit(self.b_g1_query, d_zc[1:], stream=stream, own_stream=True)
        accs.update(h_acc=t_h.collect(), l_acc=t_l.collect(), a_acc=t_a.collect(), b_g1_acc=t_b1.collect())
        a, b, c = assemble_proof(self.pairing, self.points, accs, r_mont, s_mont)
        return (a, b, c), ark_serialize.proof_to_bytes(self.pairing, a, b, c)

    def free(self):
        for b in (self.h_query, self.l_query, self.a_query, self.b_g1_query, self.b_g2_query):
            b.free()
        for mtx in (self.A, self.B, self.C):
            mtx.free()
