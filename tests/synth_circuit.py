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
GATE_COMPARISON, GATE_U32_ARITHMETIC, GATE_U32_RANGE_CHECK = 5, 6, 7
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


def comparison_row(a, b, num_bits=32, num_chunks=16):
    """Wires of a ComparisonGate row — witness logic of city_common_circuit/src/u32/gates/comparison.rs:440-520."""
    cb = -(-num_bits // num_chunks)
    cs = 1 << cb
    fch = [(a >> (cb * i)) % cs for i in range(num_chunks)]
    sch = [(b >> (cb * i)) % cs for i in range(num_chunks)]
    eq = [int(f == s) for f, s in zip(fch, sch)]
    dummy = [1 if f == s else pow((s - f) % P, P - 2, P) for f, s in zip(fch, sch)]
    msd, inter = 0, []
    for i in range(num_chunks):
        if fch[i] != sch[i]:
            msd = (sch[i] - fch[i]) % P
            inter.append(0)
        else:
            inter.append(msd)
    two_n_plus = (cs + msd) % P
    bits = [(two_n_plus >> i) & 1 for i in range(cb + 1)]
    return [a, b, int(a <= b), msd] + fch + sch + dummy + eq + inter + bits


def u32_arithmetic_row(ops):
    """ops: list of (m0, m1, addend), all < 2^32 — arithmetic_u32.rs:375-425."""
    n = len(ops)
    routed, limbs = [], []
    for m0, m1, ad in ops:
        out = m0 * m1 + ad
        hi, lo = out >> 32, out & 0xFFFFFFFF
        diff = 0xFFFFFFFF - hi
        inv = 0 if diff == 0 else pow(diff, P - 2, P)
        routed += [m0, m1, ad, lo, hi, inv]
        limbs += [(out >> (2 * j)) & 3 for j in range(32)]
    return routed + limbs


def u32_range_check_row(vals):
    """vals: list of u32 — range_check_u32.rs (aux limbs = base-4 digits, little endian)."""
    return list(vals) + [(v >> (2 * j)) & 3 for v in vals for j in range(16)]


def build(db=5, num_routed=8, num_wires=12, chunk=4, nc=2, seed=0, rate_bits=3, cap_height=2, pow_bits=5,
          num_query_rounds=4, arity_bits=(2,), n_copies=6, poseidon_fraction=0.0, backend=None, u32_gates=False):
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
    if u32_gates:
        assert num_wires >= 119
    nsel = 1 + int(u32_gates) + int(with_poseidon)
    ncst = nsel + 2
    npp = (num_routed + chunk - 1) // chunk - 1
    k_is = [pow(7, j, P) for j in range(num_routed)]
    gate_list = [(GATE_NOOP, 0, 0, 4, 0, 0), (GATE_CONSTANT, 0, 0, 4, 2, 0), (GATE_PUBLIC_INPUT, 0, 0, 4, 0, 0),
                 (GATE_ARITHMETIC, 0, 0, 4, num_ops, 0)]
    u32_ids = {}
    if u32_gates:   # second selector group: the in-tree u32 gates (degree 4 each)
        base = len(gate_list)
        gate_list += [(GATE_COMPARISON, 1, base, base + 3, 32, 16), (GATE_U32_ARITHMETIC, 1, base, base + 3, 3, 0),
                      (GATE_U32_RANGE_CHECK, 1, base, base + 3, 7, 0)]
        u32_ids = {"cmp": base, "arith": base + 1, "range": base + 2}
    pos_id = None
    if with_poseidon:
        pos_id = len(gate_list)
        gate_list.append((GATE_POSEIDON, nsel - 1, pos_id, pos_id + 1, 0, 0))
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
            return pos_id
        if u32_gates and rng.random() < 0.4:
            return int(rng.choice(list(u32_ids.values())))
        return 3 if rng.random() < 0.8 else 0
    gate_of_row = [2, 1] + [pick() for _ in range(n - 2)]
    sels = np.full((nsel, n), UNUSED, dtype=np.uint64)
    for i, g in enumerate(gate_of_row):
        sels[gate_list[g][1], i] = g
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
    for i in range(n):
        g = gate_of_row[i]
        if u32_gates and g == u32_ids["cmp"]:
            a, b = int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32))
            if rng.random() < 0.2:
                b = a
            row = comparison_row(a, b)
        elif u32_gates and g == u32_ids["arith"]:
            row = u32_arithmetic_row([tuple(int(v) for v in rng.integers(0, 2**32, 3)) for _ in range(3)])
        elif u32_gates and g == u32_ids["range"]:
            row = u32_range_check_row([int(v) for v in rng.integers(0, 2**32, 7)])
        else:
            continue
        wires[:len(row), i] = np.array(row, dtype=np.uint64)
    prow = [i for i in range(n) if pos_id is not None and gate_of_row[i] == pos_id]
    if prow:
        rows = backend.poseidon_rows(np.ascontiguousarray(wires[:12, prow].T), rng.integers(0, 2, len(prow), dtype=np.uint64))
        wires[:135, prow] = rows.T
    cs_values = np.vstack([sels, c0[None, :], c1[None, :], sigma]).astype(np.uint64)
    return dict(shape=shape, gates=gates, k_is=k_is, public_inputs=public_inputs,
                cs_values=np.ascontiguousarray(cs_values), wires=np.ascontiguousarray(wires), gate_of_row=gate_of_row,
                num_ops=num_ops, gate_list=[tuple(int(v) for v in g) for g in gate_list], num_selectors=nsel,
                num_constants=ncst, num_partial_products=npp, poseidon_gate_index=pos_id, u32_gate_ids=u32_ids)
