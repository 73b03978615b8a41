"""Shared helpers of the generic-FRI tests (tests/test_fri_generic.py on the CPU, tests/test_gpu_fri_generic.py on the GPU):
seeded FRI instances (oracles x opening batches), the two backends behind one small interface — the CPU oracle
(oracle_lib.Batch / fri_prove) and the HIP library (cityprover.PolyBatch / fri_prove) — and a toy AIR proved the way a
STARK prover built on plonky2's FRI proves (starkyx `ByteStark::prove`,
city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:518-524, is that shape of client): commit the trace with
`PolynomialBatch::from_values`, draw alphas, commit the quotient, draw zeta, open, `prove_openings`.
Test infrastructure only."""
import numpy as np

import oracle_lib as O

P = O.P
W = 7  # F_p^2 = F_p[X]/(X^2 - 7)


# ---- F_p^2 in Python integers -----------------------------------------------------------------------------------------
def e_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def e_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def e_mul(a, b):
    return ((a[0] * b[0] + W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def e_scale(a, s):
    return (a[0] * s % P, a[1] * s % P)


def e_inv(a):
    nrm = (a[0] * a[0] - W * a[1] * a[1]) % P
    ni = pow(nrm, P - 2, P)
    return (a[0] * ni % P, (-a[1]) * ni % P)


def e_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = e_mul(r, a)
        a = e_mul(a, a)
        e >>= 1
    return r


def root_of_unity(bits):
    return pow(7, (P - 1) >> bits, P)


# ---- the two backends -------------------------------------------------------------------------------------------------
class OracleBackend:
    name = "oracle"

    def commit(self, polys, rate_bits, cap_height, from_coeffs=False, salts=None):
        return O.Batch(polys, rate_bits, cap_height, from_coeffs, salts)

    def challenger(self):
        return O.challenger_new()

    def observe(self, c, elems):
        O.challenger_observe(c, elems)

    def challenges(self, c, count):
        return O.challenger_challenges(c, count)

    def state(self, c):
        return O.challenger_tuple(c)

    def fri_params(self, *a):
        return O.fri_params(*a)

    def fri_prove(self, oracles, batches, params, c, pow_override=None):
        return O.fri_prove(oracles, batches, params, c, pow_override)[0]


class GpuBackend:
    name = "gpu"

    def __init__(self, prover):
        import cityprover
        self.cp, self.p = cityprover, prover

    def commit(self, polys, rate_bits, cap_height, from_coeffs=False, salts=None):
        return self.cp.PolyBatch(self.p, polys, rate_bits, cap_height, from_coeffs, salts)

    def challenger(self):
        return self.cp.ChallengerState()

    def observe(self, c, elems):
        c.observe(elems)

    def challenges(self, c, count):
        return c.challenges(count)

    def state(self, c):
        return c.as_tuple()

    def fri_params(self, *a):
        return self.cp.fri_params(*a)

    def fri_prove(self, oracles, batches, params, c, pow_override=None):
        return self.cp.fri_prove(self.p, oracles, batches, params, c, pow_override)


# ---- seeded instances -------------------------------------------------------------------------------------------------
def random_instance(seed, degree_bits=None, max_k=12):
    """(spec dict) a FRI instance with random oracles, opening batches (as runs, possibly many and overlapping) and FriParams."""
    rng = np.random.default_rng(seed)
    db = int(rng.integers(3, 9)) if degree_bits is None else degree_bits
    rb = int(rng.integers(1, 4))
    lb = db + rb
    ch = int(rng.integers(0, min(lb, 5) + 1))
    arity = []
    bits = lb
    while len(arity) < 4 and rng.random() < 0.7:
        a = int(rng.integers(1, 5))
        if bits - a < max(ch, rb):
            break
        arity.append(a)
        bits -= a
    n_or = int(rng.integers(1, 6))
    ks = [int(rng.integers(1, max_k + 1)) for _ in range(n_or)]
    blinding = [bool(rng.random() < 0.25) for _ in range(n_or)]
    n_b = int(rng.integers(1, 5))
    batches = []
    for _ in range(n_b):
        n_r = int(rng.integers(1, 7)) if rng.random() < 0.8 else int(rng.integers(17, 40))  # > 16 runs: several combine launches
        rr = []
        for _ in range(n_r):
            o = int(rng.integers(0, n_or))
            f = int(rng.integers(0, ks[o]))
            c = int(rng.integers(1, ks[o] - f + 1))
            rr.append((o, f, c))
        batches.append(rr)
    return dict(seed=seed, degree_bits=db, rate_bits=rb, cap_height=ch, arity_bits=tuple(arity), pow_bits=int(rng.integers(0, 9)),
                num_query_rounds=int(rng.integers(1, 9)), ks=ks, blinding=blinding, batches=batches,
                from_coeffs=[bool(rng.random() < 0.3) for _ in range(n_or)])


def instance_inputs(spec):
    """the polynomials (and salts) of an instance, from its seed (numpy's PCG64: 10^8 elements in a second)"""
    n = 1 << spec["degree_bits"]
    N = n << spec["rate_bits"]
    rng = np.random.default_rng(spec["seed"] + 0x5EED)
    polys = [rng.integers(0, P, size=(k, n), dtype=np.uint64) for k in spec["ks"]]
    salts = [rng.integers(0, P, size=(4, N), dtype=np.uint64) if b else None for b in spec["blinding"]]
    return polys, salts


def run_instance(be, spec, polys=None, salts=None, pow_override=None):
    """commit -> transcript -> openings -> prove_openings on backend `be`. Returns a dict with everything a verifier needs."""
    if polys is None:
        polys, salts = instance_inputs(spec)
    db, rb, ch = spec["degree_bits"], spec["rate_bits"], spec["cap_height"]
    fc = spec.get("from_coeffs", [False] * len(polys))
    B = [be.commit(p, rb, ch, fc[i], salts[i] if salts else None) for i, p in enumerate(polys)]
    try:
        c = be.challenger()
        caps = [b.cap() for b in B]
        for cap in caps:
            be.observe(c, cap)
        # opening points: zeta from the transcript, then g^j * zeta (local / next-row / further rows)
        zeta = tuple(int(v) for v in be.challenges(c, 2))
        g = root_of_unity(db)
        batches, opened = [], []
        for j, rr in enumerate(spec["batches"]):
            pt = e_scale(zeta, pow(g, j, P))
            batches.append((pt, rr))
            vals = [B[o].eval_ext(np.array(pt, dtype=np.uint64), f, cnt) for o, f, cnt in rr]
            opened.append(np.concatenate(vals))
        for o in opened:
            be.observe(c, o)
        state_before = be.state(c)
        params = be.fri_params(db, rb, ch, spec["pow_bits"], spec["num_query_rounds"], spec["arity_bits"])
        proof = be.fri_prove(B, batches, params, c, pow_override)
        return dict(caps=caps, batches=batches, opened=opened, proof=proof, state_before=state_before, state_after=be.state(c),
                    infos=[(k, bl) for k, bl in zip(spec["ks"], spec["blinding"])])
    finally:
        for b in B:
            b.close()


def replay_challenger(kind, res):
    """a fresh transcript brought to the point where prove_openings / verify_fri_proof start (caps, zeta, openings observed)"""
    if kind == "oracle":
        c = O.challenger_new()
        for cap in res["caps"]:
            O.challenger_observe(c, cap)
        O.challenger_challenges(c, 2)
        for o in res["opened"]:
            O.challenger_observe(c, o)
        return c
    import cityprover
    c = cityprover.ChallengerState()
    for cap in res["caps"]:
        c.observe(cap)
    c.challenges(2)
    for o in res["opened"]:
        c.observe(o)
    return c


def verify_both(spec, res):
    """the oracle's verifier and the product's host verifier (cp_fri_verify) on one result; both must accept and leave the
    transcript where the prover left it"""
    import cityprover
    db, rb, ch = spec["degree_bits"], spec["rate_bits"], spec["cap_height"]
    po = O.fri_params(db, rb, ch, spec["pow_bits"], spec["num_query_rounds"], spec["arity_bits"])
    c = replay_challenger("oracle", res)
    rc, _ = O.fri_verify(po, res["infos"], res["caps"], res["batches"], res["opened"], c, res["proof"])
    assert rc == 0, f"oracle verifier rejects: {rc}"
    assert O.challenger_tuple(c) == res["state_after"]
    pc = cityprover.fri_params(db, rb, ch, spec["pow_bits"], spec["num_query_rounds"], spec["arity_bits"])
    s = replay_challenger("product", res)
    cityprover.fri_verify(pc, res["infos"], res["caps"], res["batches"], res["opened"], s, res["proof"])
    assert s.as_tuple() == res["state_after"]


# ---- a toy AIR: Fibonacci with a running sum --------------------------------------------------------------------------
# Columns a, b, s over n rows:  a' = b,  b' = a + b,  s' = s + a*b*b  (degree 3: both quotient chunks are non-zero);  first row: a = 0, b = 1, s = 0.
# Quotient per challenge alpha_i: t_i = (sum_j alpha_i^j C_j) / Z_H with the transition constraints multiplied by
# (x - g^(n-1)) and the first-row constraints by L_0(x) = Z_H(x) / (n (x - 1)) — starky's `constraint_transition` /
# `constraint_first_row`. Degree < 2n, committed as two degree-n chunks per challenge.
def toy_trace(n):
    a, b, s = [0] * n, [0] * n, [0] * n
    b[0] = 1
    for i in range(1, n):
        a[i] = b[i - 1]
        b[i] = (a[i - 1] + b[i - 1]) % P
        s[i] = (s[i - 1] + a[i - 1] * b[i - 1] * b[i - 1]) % P
    return np.array([a, b, s], dtype=np.uint64)


def toy_constraints(loc, nxt, x, n, g_last, add, sub, mul, scale, inv, one):
    """the five constraints at a point, in a field given by its operations (base field on the coset, F_p^2 at zeta);
    x: the point; returns the list C_j(x) already multiplied by its selector polynomial, NOT divided by Z_H"""
    a, b, s = loc
    a2, b2, s2 = nxt
    zh = sub(_pw(x, n, mul, one), one)
    l0 = mul(zh, inv(scale(sub(x, one), n % P)))
    tr = sub(x, scale(one, g_last))
    return [mul(sub(a2, b), tr), mul(sub(b2, add(a, b)), tr), mul(sub(s2, add(s, mul(mul(a, b), b))), tr),
            mul(a, l0), mul(sub(b, one), l0), mul(s, l0)], zh


def _pw(x, e, mul, one):
    r = one
    while e:
        if e & 1:
            r = mul(r, x)
        x = mul(x, x)
        e >>= 1
    return r


def toy_stark_prove(be, degree_bits=6, rate_bits=1, cap_height=2, pow_bits=6, num_query_rounds=9, arity_bits=(2,), num_challenges=2):
    n = 1 << degree_bits
    N = n << rate_bits
    g = root_of_unity(degree_bits)
    trace = toy_trace(n)
    T = be.commit(trace, rate_bits, cap_height)
    c = be.challenger()
    be.observe(c, T.cap())
    alphas = [int(v) for v in be.challenges(c, num_challenges)]
    # constraint evaluation on the LDE coset from `get_lde_values` (natural order; the next row is 2^rate_bits positions on)
    rows = T.lde_rows(0, N, 1)
    wN = root_of_unity(degree_bits + rate_bits)
    fa = lambda u, v: (u + v) % P
    fs = lambda u, v: (u - v) % P
    fm = lambda u, v: u * v % P
    fi = lambda u: pow(u, P - 2, P)
    q_vals = np.zeros((num_challenges, N), dtype=np.uint64)
    x = 7
    g_last = pow(g, n - 1, P)
    for i in range(N):
        loc = [int(v) for v in rows[i]]
        nxt = [int(v) for v in rows[(i + (1 << rate_bits)) % N]]
        cons, zh = toy_constraints(loc, nxt, x, n, g_last, fa, fs, fm, fm, fi, 1)
        zhi = fi(zh)
        for ci, al in enumerate(alphas):
            acc = 0
            for cj in reversed(cons):
                acc = (acc * al + cj) % P
            q_vals[ci, i] = acc * zhi % P
        x = x * wN % P
    # coset iNTT (size N) -> coefficients, split into degree-n chunks
    shift_inv = pow(7, P - 2, P)
    chunks = []
    for ci in range(num_challenges):
        co = O.intt(q_vals[ci])
        sc, acc = [], 1
        for j in range(N):
            sc.append(int(co[j]) * acc % P)
            acc = acc * shift_inv % P
        for k in range(1 << rate_bits):
            chunks.append(sc[k * n:(k + 1) * n])
    Q = be.commit(np.array(chunks, dtype=np.uint64), rate_bits, cap_height, from_coeffs=True)
    be.observe(c, Q.cap())
    zeta = tuple(int(v) for v in be.challenges(c, 2))
    zeta_next = e_scale(zeta, g)
    kq = len(chunks)
    batches = [(zeta, [(0, 0, 3), (1, 0, kq)]), (zeta_next, [(0, 0, 3)])]
    opened = [np.concatenate([T.eval_ext(np.array(zeta, dtype=np.uint64)), Q.eval_ext(np.array(zeta, dtype=np.uint64))]),
              T.eval_ext(np.array(zeta_next, dtype=np.uint64))]
    for o in opened:
        be.observe(c, o)
    params = be.fri_params(degree_bits, rate_bits, cap_height, pow_bits, num_query_rounds, arity_bits)
    proof = be.fri_prove([T, Q], batches, params, c)
    out = dict(trace_cap=T.cap(), quotient_cap=Q.cap(), opened=opened, proof=proof, state_after=be.state(c), kq=kq,
               cfg=(degree_bits, rate_bits, cap_height, pow_bits, num_query_rounds, arity_bits, num_challenges))
    T.close()
    Q.close()
    return out


def toy_stark_verify(pr, use_product=True):
    """constraint check at zeta from the openings + the FRI verifier (the product's host verifier or the oracle's)"""
    import cityprover
    degree_bits, rate_bits, cap_height, pow_bits, nq, arity_bits, num_challenges = pr["cfg"]
    n = 1 << degree_bits
    g = root_of_unity(degree_bits)
    if use_product:
        c = cityprover.ChallengerState()
        obs, chal = (lambda e: c.observe(e)), (lambda k: c.challenges(k))
    else:
        c = O.challenger_new()
        obs, chal = (lambda e: O.challenger_observe(c, e)), (lambda k: O.challenger_challenges(c, k))
    obs(pr["trace_cap"])
    alphas = [int(v) for v in chal(num_challenges)]
    obs(pr["quotient_cap"])
    zeta = tuple(int(v) for v in chal(2))
    zeta_next = e_scale(zeta, g)
    o0, o1 = pr["opened"]
    ext = lambda row: (int(row[0]), int(row[1]))
    loc = [ext(o0[j]) for j in range(3)]
    nxt = [ext(o1[j]) for j in range(3)]
    one = (1, 0)
    cons, zh = toy_constraints(loc, nxt, zeta, n, pow(g, n - 1, P), e_add, e_sub, e_mul, e_scale, e_inv, one)
    zn = e_pow(zeta, n)
    for ci, al in enumerate(alphas):
        acc = (0, 0)
        for cj in reversed(cons):
            acc = e_add(e_scale(acc, al), cj)
        t = (0, 0)
        for k in reversed(range(1 << rate_bits)):
            t = e_add(e_mul(t, zn), ext(o0[3 + ci * (1 << rate_bits) + k]))
        if e_mul(t, zh) != acc:
            return "constraints fail at zeta"
    for o in pr["opened"]:
        obs(o)
    batches = [(zeta, [(0, 0, 3), (1, 0, pr["kq"])]), (zeta_next, [(0, 0, 3)])]
    infos = [(3, 0), (pr["kq"], 0)]
    caps = [pr["trace_cap"], pr["quotient_cap"]]
    if use_product:
        try:
            cityprover.fri_verify(cityprover.fri_params(degree_bits, rate_bits, cap_height, pow_bits, nq, arity_bits), infos, caps, batches,
                                  pr["opened"], c, pr["proof"])
        except cityprover.CityProverError as e:
            return str(e)
        return None if c.as_tuple() == pr["state_after"] else "transcript differs"
    rc, _ = O.fri_verify(O.fri_params(degree_bits, rate_bits, cap_height, pow_bits, nq, arity_bits), infos, caps, batches, pr["opened"], c,
                         pr["proof"])
    return None if rc == 0 else f"oracle verifier: {rc}"
