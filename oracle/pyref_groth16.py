"""TEST INFRASTRUCTURE -- pure-Python Groth16 over a toy R1CS with a KNOWN trapdoor, in the exponent.

The reference proves with ark-groth16 0.3 (`Groth16::<Bls12_381>::prove`, lib/src/zk/encryption.rs:76 and siblings) and its
tests only assert `verify() == true`.  A pairing is not available here, so the same statement is checked on
discrete logarithms: every element of a synthetic proving key is [d] G for a d this module knows (tau, alpha, beta, gamma,
delta are drawn from a seed), hence the three proof elements an honest prover outputs have computable logarithms a, b, c
and `e(A, B) = e(alpha, beta) e(sum pub_j abc_j, gamma) e(C, delta)` is the scalar identity
    a b = alpha beta + sum_{j < num_inputs} z_j (beta u_j + alpha v_j + w_j) + c delta        (mod r).
The GPU path must produce exactly [a] G1, [b] G2, [c] G1.  PARITY UNPINNED like the rest of oracle/ (restates ark-groth16
0.3 generate_parameters / create_proof from the published algorithm; no reference vectors exist).

R1CS here = three lists of rows; a row is a list of (coefficient, variable index); variables: z[0] = 1, then the other
instance variables, then the witness.  QAP conventions are ark-groth16's: domain size m = next_pow2(num_constraints +
num_inputs), the rows num_constraints + j of A are the input-consistency rows (a single 1 at variable j).
"""
try:
    from . import pyref
except ImportError:
    import pyref


def random_r1cs(field, seed, num_inputs=3, num_constraints=40, long_rows=()):
    """a satisfied system: constraint i defines a fresh witness variable w = <LA, z> <LB, z>; `long_rows` lists constraints
    whose A row spans every variable defined so far (the reference's public-input packing has such rows)"""
    p = pyref.FIELDS[field][0]
    rng = pyref.Rng(seed)
    z = [1] + [rng.below(p) for _ in range(num_inputs - 1)] + [rng.below(p) for _ in range(4)]
    A, B, C = [], [], []

    def lin(nterms):
        idx = sorted({rng.below(len(z)) for _ in range(nterms)})
        out = []
        for j in idx:
            k = rng.below(4)
            out.append(((1, p - 1, 2, rng.below(p))[k], j))
        return out

    for i in range(num_constraints):
        la = [(1 if rng.below(2) else rng.below(p), j) for j in range(len(z))] if i in long_rows else lin(1 + rng.below(4))
        lb = lin(1 + rng.below(3))
        va = sum(c * z[j] for c, j in la) % p
        vb = sum(c * z[j] for c, j in lb) % p
        z.append(va * vb % p)
        A.append(la)
        B.append(lb)
        C.append([(1, len(z) - 1)])
    return {"field": field, "num_inputs": num_inputs, "A": A, "B": B, "C": C}, z


def domain(field, num_constraints, num_inputs):
    m = 1
    while m < num_constraints + num_inputs:
        m *= 2
    return m, pyref.root_of_unity(field, m.bit_length() - 1)


def evaluations(r1cs, z):
    """the vectors witness_map starts from: a, b, c on the domain"""
    p = pyref.FIELDS[r1cs["field"]][0]
    nc, ni = len(r1cs["A"]), r1cs["num_inputs"]
    m, _ = domain(r1cs["field"], nc, ni)
    ev = lambda rows: [sum(c * z[j] for c, j in row) % p for row in rows] + [0] * (m - nc)
    a, b, c = ev(r1cs["A"]), ev(r1cs["B"]), ev(r1cs["C"])
    for j in range(ni):
        a[nc + j] = z[j]
    return a, b, c


def h_coefficients(r1cs, z):
    """h(X) = (a(X) b(X) - c(X)) / (X^m - 1) by naive interpolation, product and exact division"""
    field = r1cs["field"]
    p = pyref.FIELDS[field][0]
    a, b, c = evaluations(r1cs, z)
    m, w = domain(field, len(r1cs["A"]), r1cs["num_inputs"])
    winv, minv = pow(w, -1, p), pow(m, -1, p)
    coef = lambda e: [x * minv % p for x in pyref.dft_naive(field, e, winv)]
    ca, cb, cc = coef(a), coef(b), coef(c)
    prod = [0] * (2 * m - 1)
    for i, x in enumerate(ca):
        if x:
            for j, y in enumerate(cb):
                prod[i + j] = (prod[i + j] + x * y) % p
    for i, x in enumerate(cc):
        prod[i] = (prod[i] - x) % p
    # divide by X^m - 1: q[k] = prod[k + m] + q[k + m]
    q = [0] * (m - 1)
    for k in range(m - 2, -1, -1):
        q[k] = (prod[k + m] + (q[k + m] if k + m < m - 1 else 0)) % p
    rem = [(prod[k] + q[k]) % p if k < m - 1 else prod[k] for k in range(m)]
    assert not any(rem), "the assignment does not satisfy the system"
    return q + [0]       # m coefficients, the last one zero (deg h <= m - 2)


def setup(r1cs, seed):
    """ark-groth16 0.3 generate_parameters with a known trapdoor -> the discrete logs of every key element"""
    field = r1cs["field"]
    p = pyref.FIELDS[field][0]
    rng = pyref.Rng(seed)
    tau, alpha, beta, gamma, delta = (1 + rng.below(p - 1) for _ in range(5))
    nc, ni = len(r1cs["A"]), r1cs["num_inputs"]
    nvars = 1 + max(j for rows in (r1cs["A"], r1cs["B"], r1cs["C"]) for row in rows for _, j in row)
    m, w = domain(field, nc, ni)
    zt = (pow(tau, m, p) - 1) % p
    L = [zt * pow(m, -1, p) % p * pow(w, i, p) % p * pow((tau - pow(w, i, p)) % p, -1, p) % p for i in range(m)]
    u, v, wv = [0] * nvars, [0] * nvars, [0] * nvars
    for j in range(ni):
        u[j] = L[nc + j]
    for vec, rows in ((u, r1cs["A"]), (v, r1cs["B"]), (wv, r1cs["C"])):
        for i, row in enumerate(rows):
            for c, j in row:
                vec[j] = (vec[j] + c * L[i]) % p
    dinv, ginv = pow(delta, -1, p), pow(gamma, -1, p)
    abc = [(beta * u[j] + alpha * v[j] + wv[j]) % p for j in range(nvars)]
    return {"tau": tau, "alpha": alpha, "beta": beta, "gamma": gamma, "delta": delta, "m": m, "num_inputs": ni,
            "a_query": u, "b_query": v,
            "h_query": [pow(tau, i, p) * zt % p * dinv % p for i in range(m - 1)],
            "l_query": [abc[j] * dinv % p for j in range(ni, nvars)],
            "gamma_abc": [abc[j] * ginv % p for j in range(ni)], "abc": abc}


def prove_logs(r1cs, key, z, r, s):
    """discrete logs (a, b, c) of the proof an honest ark-groth16 prover outputs for assignment z and blinding r, s"""
    p = pyref.FIELDS[r1cs["field"]][0]
    ni = key["num_inputs"]
    h = h_coefficients(r1cs, z)
    a = (key["alpha"] + sum(zj * uj for zj, uj in zip(z, key["a_query"])) + r * key["delta"]) % p
    b = (key["beta"] + sum(zj * vj for zj, vj in zip(z, key["b_query"])) + s * key["delta"]) % p
    c = (sum(zj * lj for zj, lj in zip(z[ni:], key["l_query"])) + sum(hi * qi for hi, qi in zip(h, key["h_query"]))
         + s * a + r * b - r * s * key["delta"]) % p
    return a, b, c


def verify_logs(r1cs, key, z_public, a, b, c):
    """the pairing equation of Groth16 verification, on logarithms"""
    p = pyref.FIELDS[r1cs["field"]][0]
    pub = sum(zj * gj for zj, gj in zip(z_public, key["gamma_abc"])) % p
    return a * b % p == (key["alpha"] * key["beta"] + pub * key["gamma"] + c * key["delta"]) % p
