"""Synthetic SATISFIABLE circuits over an arbitrary gate set, laid out the way plonky2's `CircuitBuilder::build`
lays out selectors: gates sorted by (degree, id), greedy selector groups with `size + degree < max_degree`
(plonky2 0.2.2 `selector_polynomials`, max_degree = quotient_degree_factor + 1), a row's own group selector holds
its gate index, every other selector holds UNUSED = 2^32-1.

Witness rows follow each gate's generator (in-tree generators cited per function; upstream ones restated from
plonky2 0.2.2). Wires a row's gate does not constrain are random; a few of those free routed cells are tied by
copy constraints through sigma. This is test/bench DATA generation: it mirrors what `CircuitBuilder::build` +
witness generation hand to `prove`, not any particular city-rollup circuit."""
import numpy as np

from synth_circuit import (P, UNUSED, OracleBackend, comparison_row, u32_arithmetic_row, u32_range_check_row, CIRC)

(NOOP, CONSTANT, PUBLIC_INPUT, ARITHMETIC, POSEIDON, COMPARISON, U32_ARITHMETIC, U32_RANGE_CHECK, U32_ADD_MANY,
 U32_SUBTRACTION, U32_INTERLEAVE, UNINTERLEAVE_TO_U32, UNINTERLEAVE_TO_B32, ARITHMETIC_EXT, MUL_EXT, BASE_SUM,
 RANDOM_ACCESS, REDUCING, REDUCING_EXT, POSEIDON_MDS, COSET_INTERPOLATION, EXPONENTIATION) = range(22)

# (type, param, param2, param3) with the parameters `new_from_config(standard_recursion_config)` gives (135 wires, 80 routed)
CITY_COMMON = [  # city_common_circuit/src/builder/pad_circuit.rs:31-55 (+ Noop, PublicInput, which every circuit has)
    (NOOP, 0, 0, 0), (CONSTANT, 2, 0, 0), (PUBLIC_INPUT, 0, 0, 0), (COMPARISON, 32, 16, 0), (RANDOM_ACCESS, 4, 4, 2),
    (POSEIDON, 0, 0, 0), (POSEIDON_MDS, 0, 0, 0), (REDUCING, 43, 0, 0), (REDUCING_EXT, 32, 0, 0), (ARITHMETIC, 20, 0, 0),
    (ARITHMETIC_EXT, 10, 0, 0), (MUL_EXT, 13, 0, 0), (BASE_SUM, 63, 2, 0), (COSET_INTERPOLATION, 4, 6, 0)]
U32_GATES = [  # city_common_circuit/src/u32/gates/*.rs, num_ops from each gate's new_from_config at 135/80 wires
    (U32_ARITHMETIC, 3, 0, 0), (U32_RANGE_CHECK, 7, 0, 0), (U32_ADD_MANY, 5, 3, 0), (U32_SUBTRACTION, 6, 0, 0),
    (U32_INTERLEAVE, 3, 0, 0), (UNINTERLEAVE_TO_U32, 2, 0, 0), (UNINTERLEAVE_TO_B32, 2, 0, 0)]
OTHER_UPSTREAM = [(EXPONENTIATION, 66, 0, 0)]  # ExponentiationGate::new_from_config: min(80 - 2, (135 - 2) / 2) power bits
ALL_GATES = CITY_COMMON + U32_GATES + OTHER_UPSTREAM

_ID = {NOOP: "NoopGate", CONSTANT: "ConstantGate", PUBLIC_INPUT: "PublicInputGate", ARITHMETIC: "ArithmeticGate",
       POSEIDON: "PoseidonGate", COMPARISON: "ComparisonGate", U32_ARITHMETIC: "U32ArithmeticGate",
       U32_RANGE_CHECK: "U32RangeCheckGate", U32_ADD_MANY: "U32AddManyGate", U32_SUBTRACTION: "U32SubtractionGate",
       U32_INTERLEAVE: "U32InterleaveGate", UNINTERLEAVE_TO_U32: "UninterleaveToU32Gate",
       UNINTERLEAVE_TO_B32: "UninterleaveToB32Gate", ARITHMETIC_EXT: "ArithmeticExtensionGate", MUL_EXT: "MulExtensionGate",
       BASE_SUM: "BaseSumGate", RANDOM_ACCESS: "RandomAccessGate", REDUCING: "ReducingGate",
       REDUCING_EXT: "ReducingExtensionGate", POSEIDON_MDS: "PoseidonMdsGate", COSET_INTERPOLATION: "CosetInterpolationGate",
       EXPONENTIATION: "ExponentiationGate"}


def gate_degree(g):
    t, a, b, _ = g
    return {NOOP: 0, CONSTANT: 1, PUBLIC_INPUT: 1, POSEIDON_MDS: 1, BASE_SUM: b, REDUCING: 2, REDUCING_EXT: 2,
            U32_INTERLEAVE: 2, UNINTERLEAVE_TO_U32: 2, UNINTERLEAVE_TO_B32: 2, ARITHMETIC: 3, ARITHMETIC_EXT: 3, MUL_EXT: 3,
            COMPARISON: 1 << (-(-a // b) if b else 0), U32_ARITHMETIC: 4, U32_RANGE_CHECK: 4, U32_ADD_MANY: 4,
            U32_SUBTRACTION: 4, RANDOM_ACCESS: a + 1, COSET_INTERPOLATION: b, POSEIDON: 7, EXPONENTIATION: 4}[t]


def gate_num_wires(g):
    t, a, b, c = g
    if t == COSET_INTERPOLATION:
        n = 1 << a
        return 1 + 2 * n + 4 + 4 * ((n - 2) // (b - 1)) + 2
    return {NOOP: 0, CONSTANT: a, PUBLIC_INPUT: 4, ARITHMETIC: 4 * a, POSEIDON: 135,
            COMPARISON: 4 + 5 * b + (-(-a // b) if b else 0) + 1, U32_ARITHMETIC: 38 * a, U32_RANGE_CHECK: 17 * a,
            U32_ADD_MANY: a * (b + 21), U32_SUBTRACTION: 21 * a, U32_INTERLEAVE: 34 * a, UNINTERLEAVE_TO_U32: 67 * a,
            UNINTERLEAVE_TO_B32: 67 * a, ARITHMETIC_EXT: 8 * a, MUL_EXT: 6 * a, BASE_SUM: 1 + a,
            RANDOM_ACCESS: b * (2 + (1 << a)) + c + a * b, REDUCING: 6 + a + 2 * (a - 1), REDUCING_EXT: 6 + 2 * a + 2 * (a - 1),
            POSEIDON_MDS: 48, EXPONENTIATION: 2 + 2 * a}[t]


def selector_groups(gates, max_degree):
    """plonky2 `selector_polynomials`: gates must already be sorted by (degree, id)."""
    degs = [gate_degree(g) for g in gates]
    if degs[-1] + len(gates) - 1 <= max_degree:
        return [(0, len(gates))]
    assert degs[-1] < max_degree, "a gate's degree is too high for this quotient degree factor"
    groups, start = [], 0
    while start < len(gates):
        size = 0
        while start + size < len(gates) and size + degs[start + size] < max_degree:
            size += 1
        groups.append((start, start + size))
        start += size
    return groups


# ---- F_p^2 helpers (extension-valued wires) ----
def e_mul(x, y):
    return ((x[0] * y[0] + 7 * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def e_add(x, y):
    return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)


def e_sub(x, y):
    return ((x[0] - y[0]) % P, (x[1] - y[1]) % P)


def e_scale(x, s):
    return (x[0] * s % P, x[1] * s % P)


def _felt(rng):
    return int(rng.integers(0, P, dtype=np.uint64))


def _ext(rng):
    return (_felt(rng), _felt(rng))


def _flat(pairs):
    return [v for p in pairs for v in p]


# ---- witness rows ----
def arithmetic_row(rng, num_ops, c0, c1):
    w = []
    for _ in range(num_ops):
        m0, m1, ad = _felt(rng), _felt(rng), _felt(rng)
        w += [m0, m1, ad, (m0 * m1 % P * c0 + ad * c1) % P]
    return w


def arithmetic_ext_row(rng, num_ops, c0, c1):
    w = []
    for _ in range(num_ops):
        m0, m1, ad = _ext(rng), _ext(rng), _ext(rng)
        w += _flat([m0, m1, ad, e_add(e_scale(e_mul(m0, m1), c0), e_scale(ad, c1))])
    return w


def mul_ext_row(rng, num_ops, c0):
    w = []
    for _ in range(num_ops):
        m0, m1 = _ext(rng), _ext(rng)
        w += _flat([m0, m1, e_scale(e_mul(m0, m1), c0)])
    return w


def base_sum_row(rng, num_limbs, base):
    limbs = [int(v) for v in rng.integers(0, base, num_limbs)]
    return [sum(l * base ** i for i, l in enumerate(limbs)) % P] + limbs


def exponentiation_row(rng, num_power_bits):
    """ExponentiationGenerator: intermediate_values[i] = prev^2 * (base if bit else 1), bits taken from the top"""
    base = _felt(rng)
    bits = [int(v) for v in rng.integers(0, 2, num_power_bits)]
    inter, cur = [], 1
    for i in range(num_power_bits):
        cur = cur * cur % P
        if bits[num_power_bits - 1 - i]:
            cur = cur * base % P
        inter.append(cur)
    return [base] + bits + [inter[-1]] + inter


def random_access_row(rng, bits, copies, extra, consts):
    vec = 1 << bits
    routed, bitw = [], []
    for _ in range(copies):
        idx = int(rng.integers(0, vec))
        items = [_felt(rng) for _ in range(vec)]
        routed += [idx, items[idx]] + items
        bitw += [(idx >> i) & 1 for i in range(bits)]
    return routed + list(consts[:extra]) + bitw


def reducing_row(rng, n, ext_coeffs):
    alpha, old_acc = _ext(rng), _ext(rng)
    coeffs = [_ext(rng) if ext_coeffs else (_felt(rng), 0) for _ in range(n)]
    accs, acc = [], old_acc
    for c in coeffs:
        acc = e_add(e_mul(acc, alpha), c)
        accs.append(acc)
    cw = _flat(coeffs) if ext_coeffs else [c[0] for c in coeffs]
    return _flat([accs[-1], alpha, old_acc]) + cw + _flat(accs[:-1])


def poseidon_mds_row(rng):
    ins = [_ext(rng) for _ in range(12)]
    outs = []
    for r in range(12):
        o = tuple((sum(CIRC[i] * ins[(i + r) % 12][h] for i in range(12)) + (8 * ins[0][h] if r == 0 else 0)) % P
                  for h in range(2))
        outs.append(o)
    return _flat(ins) + _flat(outs)


def coset_interpolation_row(rng, bits, degree):
    n = 1 << bits
    n_inter = (n - 2) // (degree - 1)
    g = pow(7, (P - 1) >> bits, P)
    domain = [pow(g, i, P) for i in range(n)]
    weights = []
    for i in range(n):
        d = 1
        for j in range(n):
            if j != i:
                d = d * (domain[i] - domain[j]) % P
        weights.append(pow(d, P - 2, P))
    shift = _felt(rng) or 1
    values = [_ext(rng) for _ in range(n)]
    point = _ext(rng)
    sp = e_scale(point, pow(shift, P - 2, P))

    def partial(lo, hi, ev, pr):
        for i in range(lo, hi):
            term = e_sub(sp, (domain[i], 0))
            ev, pr = e_add(e_mul(ev, term), e_mul(values[i], e_scale(pr, weights[i]))), e_mul(pr, term)
        return ev, pr
    ev, pr = partial(0, degree, (0, 0), (1, 0))
    ievals, iprods = [], []
    for i in range(n_inter):
        ievals.append(ev)
        iprods.append(pr)
        lo = 1 + (degree - 1) * (i + 1)
        ev, pr = partial(lo, min(lo + degree - 1, n), ev, pr)
    return [shift] + _flat(values) + _flat([point, ev]) + _flat(ievals) + _flat(iprods) + _flat([sp])


def add_many_row(rng, num_ops, num_addends):
    """add_many_u32.rs:309-360"""
    routed, limbs = [], []
    for _ in range(num_ops):
        addends = [int(v) for v in rng.integers(0, 2**32, num_addends)]
        carry = int(rng.integers(0, 2**32))
        out = sum(addends) + carry
        res, oc = out & 0xFFFFFFFF, out >> 32
        routed += addends + [carry, res, oc]
        limbs += [(res >> (2 * j)) & 3 for j in range(16)] + [(oc >> (2 * j)) & 3 for j in range(2)]
    return routed + limbs


def subtraction_row(rng, num_ops):
    """subtraction_u32.rs:299-330"""
    routed, limbs = [], []
    for _ in range(num_ops):
        x, y, b = int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32)), int(rng.integers(0, 2))
        r = (x - y - b) % P
        ob = int(r > (1 << 32))
        r = (r + (ob << 32)) % P
        routed += [x, y, b, r, ob]
        limbs += [(r >> (2 * j)) & 3 for j in range(16)]
    return routed + limbs


def interleave_row(rng, num_ops):
    """interleave_u32.rs:282-305 — bit wires big-endian"""
    routed, bits = [], []
    for _ in range(num_ops):
        x = int(rng.integers(0, 2**32))
        b = [(x >> (31 - i)) & 1 for i in range(32)]
        routed += [x, sum(bit << (2 * (31 - i)) for i, bit in enumerate(b))]
        bits += b
    return routed + bits


def uninterleave_row(rng, num_ops, to_b32):
    """uninterleave_to_u32.rs / uninterleave_to_b32.rs generators — 64 big-endian bit wires of a field element"""
    routed, bits = [], []
    for _ in range(num_ops):
        x = _felt(rng)
        b = [(x >> (63 - i)) & 1 for i in range(64)]
        sh = (lambda j: 2 * (31 - j)) if to_b32 else (lambda j: 31 - j)
        routed += [x, sum(b[2 * j] << sh(j) for j in range(32)), sum(b[2 * j + 1] << sh(j) for j in range(32))]
        bits += b
    return routed + bits


def build_gate_set(gate_set=CITY_COMMON, db=7, num_routed=80, num_wires=135, chunk=8, nc=2, seed=0, rate_bits=3,
                   cap_height=2, pow_bits=5, num_query_rounds=4, arity_bits=(2,), n_copies=8, weights=None,
                   noop_fraction=0.1, backend=None, witness_seed=None):
    """weights: {gate type: relative row frequency} (default: uniform over the set's non-trivial gates).
    backend: see synth_circuit.build (None = the oracle; tests only).
    witness_seed: None = one stream of randomness for circuit and witness (the historical behaviour); an integer = the CIRCUIT
    (row types, constants, copy constraints) depends on `seed` alone and the WITNESS (public inputs, every free wire value) on
    (seed, witness_seed): several satisfying witnesses of one circuit ("witness from seed = job index")."""
    use_oracle = backend is None
    if use_oracle:
        backend = OracleBackend()
    rng = np.random.default_rng(seed)
    wrng = rng if witness_seed is None else np.random.default_rng([int(seed), int(witness_seed), 0x5EED])
    n = 1 << db
    assert chunk == 1 << rate_bits
    gates = sorted(gate_set, key=lambda g: (gate_degree(g), _ID[g[0]]))
    for g in gates:
        assert gate_num_wires(g) <= num_wires, g
    groups = selector_groups(gates, chunk + 1)
    nsel = len(groups)
    ncst = nsel + 2
    gate_list = []
    for gi, g in enumerate(gates):
        si = next(s for s, (lo, hi) in enumerate(groups) if lo <= gi < hi)
        gate_list.append((g[0], si, groups[si][0], groups[si][1], g[1], g[2], g[3]))
    index_of = {g[0]: i for i, g in enumerate(gates)}
    npp = (num_routed + chunk - 1) // chunk - 1
    k_is = [pow(7, j, P) for j in range(num_routed)]
    shape = ogates = None
    if use_oracle:
        O = backend.O
        shape = O.standard_shape(degree_bits=db, num_wires=num_wires, num_routed=num_routed, num_constants=ncst,
                                 num_challenges=nc, num_partial_products=npp, quotient_degree_factor=chunk,
                                 rate_bits=rate_bits, cap_height=cap_height, pow_bits=pow_bits,
                                 num_query_rounds=num_query_rounds, arity_bits=arity_bits)
        ogates = O.make_gates(gate_list, nsel, k_is)
    public_inputs = [int(x) for x in wrng.integers(0, P, 5, dtype=np.uint64)]
    pi_hash = backend.hash_no_pad(public_inputs)

    # rows: PublicInput, Constant, one of every other gate, then a weighted random mix with Noop padding
    active = [g[0] for g in gates if g[0] not in (NOOP, PUBLIC_INPUT, CONSTANT)]
    if weights is None:
        weights = {t: 1.0 for t in active}
    wl = np.array([weights.get(t, 0.0) for t in active], dtype=np.float64)
    if active:
        wl = wl / wl.sum()
    else:
        noop_fraction = 1.0
    types = [PUBLIC_INPUT, CONSTANT] + active
    assert len(types) <= n
    while len(types) < n:
        types.append(NOOP if rng.random() < noop_fraction else int(rng.choice(active, p=wl)))
    gate_of_row = [index_of[t] for t in types]
    sels = np.full((nsel, n), UNUSED, dtype=np.uint64)
    for i, g in enumerate(gate_of_row):
        sels[gate_list[g][1], i] = g
    c0 = rng.integers(0, P, n, dtype=np.uint64)
    c1 = rng.integers(0, P, n, dtype=np.uint64)
    wires = wrng.integers(0, P, (num_wires, n), dtype=np.uint64)
    params = {g[0]: g for g in gates}
    for i, t in enumerate(types):
        _, a, b, c = params[t]
        k0, k1 = int(c0[i]), int(c1[i])
        row = None
        if t == PUBLIC_INPUT:
            row = pi_hash
        elif t == CONSTANT:
            row = [k0, k1][:a]
        elif t == ARITHMETIC:
            row = arithmetic_row(wrng, a, k0, k1)
        elif t == ARITHMETIC_EXT:
            row = arithmetic_ext_row(wrng, a, k0, k1)
        elif t == MUL_EXT:
            row = mul_ext_row(wrng, a, k0)
        elif t == BASE_SUM:
            row = base_sum_row(wrng, a, b)
        elif t == RANDOM_ACCESS:
            row = random_access_row(wrng, a, b, c, [k0, k1])
        elif t == REDUCING:
            row = reducing_row(wrng, a, False)
        elif t == REDUCING_EXT:
            row = reducing_row(wrng, a, True)
        elif t == POSEIDON_MDS:
            row = poseidon_mds_row(wrng)
        elif t == COSET_INTERPOLATION:
            row = coset_interpolation_row(wrng, a, b)
        elif t == EXPONENTIATION:
            row = exponentiation_row(wrng, a)
        elif t == COMPARISON:
            x, y = int(wrng.integers(0, 2**a)), int(wrng.integers(0, 2**a))
            row = comparison_row(x, x if wrng.random() < 0.2 else y, a, b)
        elif t == U32_ARITHMETIC:
            row = u32_arithmetic_row([tuple(int(v) for v in wrng.integers(0, 2**32, 3)) for _ in range(a)])
        elif t == U32_RANGE_CHECK:
            row = u32_range_check_row([int(v) for v in wrng.integers(0, 2**32, a)])
        elif t == U32_ADD_MANY:
            row = add_many_row(wrng, a, b)
        elif t == U32_SUBTRACTION:
            row = subtraction_row(wrng, a)
        elif t == U32_INTERLEAVE:
            row = interleave_row(wrng, a)
        elif t == UNINTERLEAVE_TO_U32:
            row = uninterleave_row(wrng, a, False)
        elif t == UNINTERLEAVE_TO_B32:
            row = uninterleave_row(wrng, a, True)
        if row is not None:
            assert len(row) == gate_num_wires(params[t]), (t, len(row))
            wires[:len(row), i] = np.array(row, dtype=np.uint64)
    prow = [i for i, t in enumerate(types) if t == POSEIDON]
    if prow:
        rows = backend.poseidon_rows(np.ascontiguousarray(wires[:12, prow].T), wrng.integers(0, 2, len(prow), dtype=np.uint64))
        wires[:135, prow] = rows.T

    # copy constraints between free routed cells (columns the row's gate does not touch)
    omega = pow(7, (P - 1) >> db, P)
    xs = [1] * n
    for i in range(1, n):
        xs[i] = xs[i - 1] * omega % P
    ident = np.array([[k_is[j] * xs[i] % P for i in range(n)] for j in range(num_routed)], dtype=np.uint64)
    sigma = ident.copy()
    free_rows = [i for i, t in enumerate(types) if gate_num_wires(params[t]) < num_routed]
    used, copies = set(), []
    for _ in range(n_copies * 4):
        if len(copies) >= n_copies or len(free_rows) < 2:
            break
        ra, rb = (int(v) for v in rng.choice(free_rows, 2, replace=False))
        ja = int(rng.integers(gate_num_wires(params[types[ra]]), num_routed))
        jb = int(rng.integers(gate_num_wires(params[types[rb]]), num_routed))
        if (ja, ra) in used or (jb, rb) in used:
            continue
        used |= {(ja, ra), (jb, rb)}
        sigma[ja, ra], sigma[jb, rb] = ident[jb, rb], ident[ja, ra]
        wires[jb, rb] = wires[ja, ra]
        copies.append(((ja, ra), (jb, rb)))
    cs_values = np.vstack([sels, c0[None, :], c1[None, :], sigma]).astype(np.uint64)
    return dict(shape=shape, gates=ogates, k_is=k_is, public_inputs=public_inputs, cs_values=np.ascontiguousarray(cs_values),
                wires=np.ascontiguousarray(wires), gate_of_row=gate_of_row, row_types=types,
                gate_list=[tuple(int(v) for v in g) for g in gate_list], num_selectors=nsel, num_constants=ncst,
                num_partial_products=npp, groups=groups, copies=copies, sorted_gates=gates)
