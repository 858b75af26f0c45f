"""TEST INFRASTRUCTURE -- pure-Python restatement of the halo2_proofs 0.2 prover steps beyond commit / FFT,
on Python ints and the affine curve arithmetic of oracle/pyref.py.  PARITY UNPINNED (published algorithm; the reference's
halo2 crate only runs MockProver, circuits-halo2/src/encryption.rs:335, and holds no vectors).

  arithmetic.rs            BatchInvert (0 stays 0), compute_inner_product
  plonk/permutation/prover.rs  Argument::commit: the grand product Z of one chunk of columns
        Z(0) = last Z of the previous chunk (1 for the first);  Z(i+1) = Z(i) * prod_col (v + beta delta^col omega^i + gamma)
                                                                          / prod_col (v + beta sigma_col(omega^i) + gamma)
  plonk/lookup/prover.rs   commit_product: Z(i+1) = Z(i) (A + beta)(S + gamma) / ((A' + beta)(S' + gamma))
  poly/commitment/prover.rs create_proof, the argument's k rounds:
        L_j = <p'_hi, G'_lo>, R_j = <p'_lo, G'_hi>, value_l = <p'_hi, b_lo>, value_r = <p'_lo, b_hi>   (before the U / W blinding terms)
        fold: p'_i += u^-1 p'_(i+half) ; b_i += u b_(i+half) ; G'_i += [u] G'_(i+half)
Blinding rows / scalars and the transcript belong to the caller (RNG and hashing stay on the CPU).
"""
try:
    from . import pyref
except ImportError:
    import pyref


def batch_invert(field, a):
    p = pyref.FIELDS[field][0]
    return [pow(x, -1, p) if x else 0 for x in a]


def prefix_product(field, f, first=1):
    """z[0] = first, z[i+1] = z[i] f[i]  -> (z of len(f) entries, product of everything)"""
    p = pyref.FIELDS[field][0]
    z, cur = [], first % p
    for x in f:
        z.append(cur)
        cur = cur * x % p
    return z, cur


def permutation_factors(field, columns, sigmas, beta, gamma, delta, omega, first_col):
    """per row: numerator / denominator of one chunk (columns first_col .. first_col + len(columns))"""
    p = pyref.FIELDS[field][0]
    n = len(columns[0])
    num, den = [1] * n, [1] * n
    for c, (v, s) in enumerate(zip(columns, sigmas)):
        dc = pow(delta, first_col + c, p) * beta % p
        w = 1
        for i in range(n):
            num[i] = num[i] * ((v[i] + dc * w + gamma) % p) % p
            den[i] = den[i] * ((v[i] + beta * s[i] + gamma) % p) % p
            w = w * omega % p
    inv = batch_invert(field, den)
    return [a * b % p for a, b in zip(num, inv)]


def lookup_factors(field, A, S, Ap, Sp, beta, gamma):
    p = pyref.FIELDS[field][0]
    den = batch_invert(field, [((a + beta) * (s + gamma)) % p for a, s in zip(Ap, Sp)])
    return [(a + beta) * (s + gamma) % p * d % p for a, s, d in zip(A, S, den)]


def inner_product(field, a, b):
    p = pyref.FIELDS[field][0]
    return sum(x * y for x, y in zip(a, b)) % p


def ipa_round(curve, p_prime, b, g):
    """-> (L, R, value_l, value_r) of one round on vectors of length 2 * half; points as pyref affine tuples / None"""
    sf = pyref.CURVES[curve][1]
    half = len(p_prime) // 2
    L = pyref.msm_naive(curve, p_prime[half:], g[:half])
    R = pyref.msm_naive(curve, p_prime[:half], g[half:])
    return L, R, inner_product(sf, p_prime[half:], b[:half]), inner_product(sf, p_prime[:half], b[half:])


def ipa_fold(curve, p_prime, b, g, u):
    sf = pyref.CURVES[curve][1]
    r = pyref.FIELDS[sf][0]
    half = len(p_prime) // 2
    ui = pow(u, -1, r)
    p2 = [(p_prime[i] + p_prime[i + half] * ui) % r for i in range(half)]
    b2 = [(b[i] + b[i + half] * u) % r for i in range(half)]
    g2 = [pyref.ec_add(curve, g[i], pyref.ec_mul(curve, u, g[i + half])) for i in range(half)]
    return p2, b2, g2


def ipa_argument(curve, p_prime, b, g, challenges):
    """all k rounds with the given challenges -> ([(L, R, value_l, value_r)], final c = p'[0], final b, final G)"""
    rounds = []
    for u in challenges:
        rounds.append(ipa_round(curve, p_prime, b, g))
        p_prime, b, g = ipa_fold(curve, p_prime, b, g, u)
    assert len(p_prime) == 1
    return rounds, p_prime[0], b[0], g[0]


def kate_division(field, a, b):
    """arithmetic.rs kate_division(a, b): the loop as published -- b = -b; q = [0] * (len(a) - 1); tmp = 0; walking q and a from
    the top: lead = a_i - tmp; q_i = lead; tmp = lead * b -- i.e. the quotient of a(X) / (X - b), remainder dropped"""
    p = pyref.FIELDS[field][0]
    nb = (-b) % p
    q = [0] * (len(a) - 1)
    tmp = 0
    for j in range(len(q) - 1, -1, -1):
        lead = (a[j + 1] - tmp) % p
        q[j] = lead
        tmp = lead * nb % p
    return q


def multiopen_quotient(field, q_polys, point_sets, x_2, n):
    """poly/multiopen/prover.rs create_proof, from the per-set folded polynomials to q'(X): every set's polynomial divided by
    (X - point) for each point of the set (kate_division folded over the points), resized to n, and the sets combined as
    q' = q' x_2 + poly"""
    p = pyref.FIELDS[field][0]
    acc = None
    for poly, points in zip(q_polys, point_sets):
        cur = list(poly)
        for pt in points:
            cur = kate_division(field, cur, pt)
        cur = cur + [0] * (n - len(cur))
        acc = cur if acc is None else [(u * x_2 + v) % p for u, v in zip(acc, cur)]
    return acc


def permute_expression_pair(field, inputs, table, usable_rows):
    """plonk/lookup/prover.rs permute_expression_pair, the deterministic part (the blinding rows appended afterwards are the
    caller's RNG): A' = the first usable_rows inputs sorted (the field's Ord = canonical integers); S': at the first
    occurrence of every input value that value (one instance leaves the table's multiset; an input value that is not in the
    table is an error), and the leftover table values, ascending, into the repeated-input rows taken from the LAST one down
    (upstream pops them off a Vec)"""
    a = sorted(inputs[:usable_rows])
    left = {}
    for v in table[:usable_rows]:
        left[v] = left.get(v, 0) + 1
    s_perm = [0] * usable_rows
    repeated = []
    for row, v in enumerate(a):
        if row == 0 or v != a[row - 1]:
            s_perm[row] = v
            if left.get(v, 0) <= 0:
                raise ValueError("ConstraintSystemFailure: input value not in the table")
            left[v] -= 1
        else:
            repeated.append(row)
    for v in sorted(left):
        for _ in range(left[v]):
            s_perm[repeated.pop()] = v
    assert not repeated
    return a, s_perm


def eval_program(field, program, columns, consts, n_ext, rot_scale, i):
    """the quotient evaluator's stack program at row i of the extended domain (see include/zkcp_amd_prover.h, zk_expr_*):
    ops: ("col", column, rotation) ("const", index) ("add",) ("sub",) ("mul",) ("neg",) ("scale", index)"""
    p = pyref.FIELDS[field][0]
    st = []
    for op in program:
        if op[0] == "col":
            st.append(columns[op[1]][(i + op[2] * rot_scale) % n_ext])
        elif op[0] == "const":
            st.append(consts[op[1]] % p)
        elif op[0] == "neg":
            st.append((-st.pop()) % p)
        elif op[0] == "scale":
            st.append(st.pop() * consts[op[1]] % p)
        else:
            y, x = st.pop(), st.pop()
            st.append((x + y) % p if op[0] == "add" else (x - y) % p if op[0] == "sub" else x * y % p)
    assert len(st) == 1
    return st[0]
