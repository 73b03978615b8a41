"""A small synthetic, SATISFIABLE plonky2-style circuit for tests: rows of Arithmetic / Constant /
PublicInput / Noop gates with two selector-free groups... kept deliberately simple:

  gate 0 = NoopGate, gate 1 = ConstantGate(2), gate 2 = PublicInputGate, gate 3 = ArithmeticGate(num_ops)
  one selector polynomial (group = all four gates, so num_selectors = 1), 2 gate constants
  => constants columns = [selector, c0, c1].

Copy constraints: a random set of wire cells is tied pairwise through sigma (cells tied together are
given equal values), everything else is the identity permutation. The witness is built so that every
gate constraint holds. This mirrors the data `CircuitBuilder::build` + witness generation hand to
`prove` (circuits arrive as data), not any particular city-rollup circuit."""
import numpy as np

import oracle_lib as O

P = O.P


def build(db=5, num_routed=8, num_wires=12, chunk=4, nc=2, seed=0, rate_bits=3, cap_height=2, pow_bits=5,
          num_query_rounds=4, arity_bits=(2,), n_copies=6):
    rng = np.random.default_rng(seed)
    n = 1 << db
    num_ops = num_routed // 4
    assert chunk == 1 << rate_bits, "quotient_degree_factor must equal the blow-up (step = 1)"
    npp = (num_routed + chunk - 1) // chunk - 1
    shape = O.standard_shape(degree_bits=db, num_wires=num_wires, num_routed=num_routed, num_constants=3,
                             num_challenges=nc, num_partial_products=npp, quotient_degree_factor=chunk,
                             rate_bits=rate_bits, cap_height=cap_height, pow_bits=pow_bits,
                             num_query_rounds=num_query_rounds, arity_bits=arity_bits)
    k_is = [pow(7, j, P) for j in range(num_routed)]
    gates = O.make_gates([(O.GATE_NOOP, 0, 0, 4, 0), (O.GATE_CONSTANT, 0, 0, 4, 2), (O.GATE_PUBLIC_INPUT, 0, 0, 4, 0),
                          (O.GATE_ARITHMETIC, 0, 0, 4, num_ops)], 1, k_is)
    public_inputs = [int(x) for x in rng.integers(0, P, 5, dtype=np.uint64)]
    pi_hash = [int(x) for x in O.hash_no_pad(O.arr(public_inputs))]
    # gate per row: row 0 public input, row 1 constant, the rest mostly arithmetic, some noop
    gate_of_row = [2, 1] + [3 if rng.random() < 0.8 else 0 for _ in range(n - 2)]
    sel = np.array(gate_of_row, dtype=np.uint64)                      # selector value = gate index
    c0 = rng.integers(0, P, n, dtype=np.uint64)
    c1 = rng.integers(0, P, n, dtype=np.uint64)
    wires = rng.integers(0, P, (num_wires, n), dtype=np.uint64)

    def fix_row(i):
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

    # copy constraints between INPUT cells of arithmetic rows (multiplicands / addends), set before the outputs
    omega = pow(7, (P - 1) >> db, P)
    ident = np.array([[k_is[j] * pow(omega, i, P) % P for i in range(n)] for j in range(num_routed)], dtype=np.uint64)
    sigma = ident.copy()
    arith_rows = [i for i in range(n) if gate_of_row[i] == 3]
    used = set()
    for _ in range(n_copies):
        if len(arith_rows) < 2 or num_ops == 0:
            break
        (ra, rb) = rng.choice(arith_rows, 2, replace=False)
        ja, jb = 4 * int(rng.integers(0, num_ops)) + int(rng.integers(0, 3)), 4 * int(rng.integers(0, num_ops)) + int(rng.integers(0, 3))
        if (ja, ra) in used or (jb, rb) in used or (ja, ra) == (jb, rb):
            continue
        used |= {(ja, ra), (jb, rb)}
        sigma[ja, ra], sigma[jb, rb] = ident[jb, rb], ident[ja, ra]
        wires[jb, rb] = wires[ja, ra]
    for i in range(n):
        fix_row(i)
    cs_values = np.vstack([sel[None, :], c0[None, :], c1[None, :], sigma]).astype(np.uint64)
    return dict(shape=shape, gates=gates, k_is=k_is, public_inputs=public_inputs, cs_values=np.ascontiguousarray(cs_values),
                wires=np.ascontiguousarray(wires), gate_of_row=gate_of_row, num_ops=num_ops,
                gate_list=[(0, 0, 0, 4, 0), (1, 0, 0, 4, 2), (2, 0, 0, 4, 0), (3, 0, 0, 4, num_ops)])
