"""Small synthetic, SATISFIABLE plonky2-style circuits for tests and benches.

  gates (index = position):  0 Noop, 1 Constant(2), 2 PublicInput, 3 Arithmetic(num_ops)   [+ 4 Poseidon]
  without Poseidon: one selector polynomial (group = gates 0..3), constants columns [sel, c0, c1]
  with Poseidon:    two selector polynomials, group 0 = gates 0..3, group 1 = gate 4 (a degree-7 gate
                    cannot share a group at quotient_degree_factor 8), constants [sel0, sel1, c0, c1];
                    a row's own group selector holds its gate index, the other holds UNUSED = 2^32-1.

Copy constraints: random pairs of arithmetic INPUT cells are tied through sigma (and given equal values);
everything else is the identity permutation. The witness satisfies every gate. This mirrors the data
`CircuitBuilder::build` + witness generation hand to `prove` (circuits arrive as data), not
any particular city-rollup circuit."""
import numpy as np

P = 0xFFFFFFFF00000001
UNUSED = 0xFFFFFFFF
GATE_NOOP, GATE_CONSTANT, GATE_PUBLIC_INPUT, GATE_ARITHMETIC, GATE_POSEIDON = 0, 1, 2, 3, 4
_RC = None
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]


class OracleBackend:
    """Default (tests): hashing / round constants from the CPU oracle, Poseidon rows in pure Python."""

    def __init__(self):
        import oracle_lib as O
        self.O = O

    def hash_no_pad(self, xs):
        return [int(v) for v in self.O.hash_no_pad(self.O.arr(xs))]

    def poseidon_rows(self, inputs, swaps):
        return np.array([poseidon_gate_row([int(v) for v in inputs[i]], int(swaps[i])) for i in range(len(swaps))],
                        dtype=np.uint64).reshape(len(swaps), 135)


def _rc():
    global _RC
    if _RC is None:
        import oracle_lib as O
        rc = np.zeros(360, np.uint64)
        O.lib().or_poseidon_round_constants(O.ptr(rc))
        _RC = [int(x) for x in rc]
    return _RC


def _mds(s):
    return [(sum(CIRC[i] * s[(i + r) % 12] for i in range(12)) + (8 * s[0] if r == 0 else 0)) % P for r in range(12)]


def poseidon_gate_row(inputs, swap):
    """All 135 wires of a PoseidonGate row (plonky2 wire layout) for 12 inputs and a swap bit."""
    rc = _rc()
    w = [0] * 135
    w[0:12] = inputs
    w[24] = swap
    st = list(inputs)
    for i in range(4):
        d = swap * ((inputs[i + 4] - inputs[i]) % P) % P
        w[25 + i] = d
        st[i] = (inputs[i] + d) % P
        st[i + 4] = (inputs[i + 4] - d) % P
    rnd = 0
    for r in range(4):
        st = [(st[i] + rc[rnd * 12 + i]) % P for i in range(12)]
        if r:
            w[29 + 12 * (r - 1):29 + 12 * r] = st
        st = _mds([pow(x, 7, P) for x in st])
        rnd += 1
    for r in range(22):
        st = [(st[i] + rc[rnd * 12 + i]) % P for i in range(12)]
        w[65 + r] = st[0]
        st[0] = pow(st[0], 7, P)
        st = _mds(st)
        rnd += 1
    for r in range(4):
        st = [(st[i] + rc[rnd * 12 + i]) % P for i in range(12)]
        w[87 + 12 * r:87 + 12 * (r + 1)] = st
        st = _mds([pow(x, 7, P) for x in st])
        rnd += 1
    w[12:24] = st
    return w


def build(db=5, num_routed=8, num_wires=12, chunk=4, nc=2, seed=0, rate_bits=3, cap_height=2, pow_bits=5,
          num_query_rounds=4, arity_bits=(2,), n_copies=6, poseidon_fraction=0.0, backend=None):
    """backend: object with hash_no_pad(list) and poseidon_rows(inputs, swaps); None = the oracle (tests).
    With a non-oracle backend the returned dict has no oracle `shape` / `gates` objects."""
    use_oracle = backend is None
    if use_oracle:
        backend = OracleBackend()
    rng = np.random.default_rng(seed)
    n = 1 << db
    num_ops = num_routed // 4
    assert chunk == 1 << rate_bits, "quotient_degree_factor must equal the blow-up (step = 1)"
    with_poseidon = poseidon_fraction > 0
    if with_poseidon:
        assert num_wires >= 135
    nsel = 2 if with_poseidon else 1
    ncst = nsel + 2
    npp = (num_routed + chunk - 1) // chunk - 1
    k_is = [pow(7, j, P) for j in range(num_routed)]
    gate_list = [(GATE_NOOP, 0, 0, 4, 0), (GATE_CONSTANT, 0, 0, 4, 2), (GATE_PUBLIC_INPUT, 0, 0, 4, 0),
                 (GATE_ARITHMETIC, 0, 0, 4, num_ops)]
    if with_poseidon:
        gate_list.append((GATE_POSEIDON, 1, 4, 5, 0))
    shape = gates = None
    if use_oracle:
        O = backend.O
        shape = O.standard_shape(degree_bits=db, num_wires=num_wires, num_routed=num_routed, num_constants=ncst,
                                 num_challenges=nc, num_partial_products=npp, quotient_degree_factor=chunk,
                                 rate_bits=rate_bits, cap_height=cap_height, pow_bits=pow_bits,
                                 num_query_rounds=num_query_rounds, arity_bits=arity_bits)
        gates = O.make_gates(gate_list, nsel, k_is)
    public_inputs = [int(x) for x in rng.integers(0, P, 5, dtype=np.uint64)]
    pi_hash = backend.hash_no_pad(public_inputs)

    def pick():
        u = rng.random()
        if with_poseidon and u < poseidon_fraction:
            return 4
        return 3 if rng.random() < 0.8 else 0
    gate_of_row = [2, 1] + [pick() for _ in range(n - 2)]
    sels = np.full((nsel, n), UNUSED, dtype=np.uint64)
    for i, g in enumerate(gate_of_row):
        sels[1 if g == 4 else 0, i] = g
    c0 = rng.integers(0, P, n, dtype=np.uint64)
    c1 = rng.integers(0, P, n, dtype=np.uint64)
    wires = rng.integers(0, P, (num_wires, n), dtype=np.uint64)

    omega = pow(7, (P - 1) >> db, P)
    xs = [1] * n
    for i in range(1, n):
        xs[i] = xs[i - 1] * omega % P
    ident = np.array([[k_is[j] * xs[i] % P for i in range(n)] for j in range(num_routed)], dtype=np.uint64)
    sigma = ident.copy()
    arith_rows = [i for i in range(n) if gate_of_row[i] == 3]
    used = set()
    for _ in range(n_copies):
        if len(arith_rows) < 2 or num_ops == 0:
            break
        (ra, rb) = rng.choice(arith_rows, 2, replace=False)
        ja = 4 * int(rng.integers(0, num_ops)) + int(rng.integers(0, 3))
        jb = 4 * int(rng.integers(0, num_ops)) + int(rng.integers(0, 3))
        if (ja, ra) in used or (jb, rb) in used or (ja, ra) == (jb, rb):
            continue
        used |= {(ja, ra), (jb, rb)}
        sigma[ja, ra], sigma[jb, rb] = ident[jb, rb], ident[ja, ra]
        wires[jb, rb] = wires[ja, ra]
    for i in range(n):
        g = gate_of_row[i]
        if g == 2:
            for j in range(4):
                wires[j, i] = pi_hash[j]
        elif g == 1:
            wires[0, i], wires[1, i] = c0[i], c1[i]
        elif g == 3:
            for op in range(num_ops):
                m0, m1, ad = (int(wires[4 * op + t, i]) for t in range(3))
                wires[4 * op + 3, i] = (m0 * m1 % P * int(c0[i]) + ad * int(c1[i])) % P
    prow = [i for i in range(n) if gate_of_row[i] == 4]
    if prow:
        rows = backend.poseidon_rows(np.ascontiguousarray(wires[:12, prow].T), rng.integers(0, 2, len(prow), dtype=np.uint64))
        wires[:135, prow] = rows.T
    cs_values = np.vstack([sels, c0[None, :], c1[None, :], sigma]).astype(np.uint64)
    return dict(shape=shape, gates=gates, k_is=k_is, public_inputs=public_inputs,
                cs_values=np.ascontiguousarray(cs_values), wires=np.ascontiguousarray(wires), gate_of_row=gate_of_row,
                num_ops=num_ops, gate_list=[tuple(int(v) for v in g) for g in gate_list], num_selectors=nsel,
                num_constants=ncst, num_partial_products=npp)
