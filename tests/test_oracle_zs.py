"""A7 oracle: Z / partial products satisfy the permutation-argument identities they exist for."""
import numpy as np

import oracle_lib as O

P = O.P


def setup(db=5, R=8, W=12, chunk=4, nc=2, seed=1):
    npp = (R + chunk - 1) // chunk - 1
    sh = O.standard_shape(degree_bits=db, num_wires=W, num_routed=R, num_constants=2, num_challenges=nc,
                          num_partial_products=npp, quotient_degree_factor=chunk, rate_bits=3, cap_height=1,
                          pow_bits=2, num_query_rounds=2, arity_bits=(1,))
    n = 1 << db
    omega = pow(7, (P - 1) >> db, P)
    k_is = [pow(7, j, P) for j in range(R)]
    ident = np.array([[k_is[j] * pow(omega, i, P) % P for i in range(n)] for j in range(R)], dtype=np.uint64)
    wires = O.splitmix64_felts(seed, W * n).reshape(W, n)
    return sh, n, omega, k_is, ident, wires


def test_identity_permutation_gives_all_ones():
    sh, n, omega, k_is, ident, wires = setup()
    out = O.zs_partial_products(sh, wires, ident, k_is, [3, 5], [7, 11])
    assert (out == 1).all()


def test_valid_copy_constraints_wrap_to_one():
    """sigma swaps (wire 1,row 3) <-> (wire 6,row 20) and the two cells hold the same value: the grand
    product over all rows is 1; with different values it is not."""
    sh, n, omega, k_is, ident, wires = setup(seed=2)
    sig = ident.copy()
    sig[1, 3], sig[6, 20] = ident[6, 20], ident[1, 3]
    wires[6, 20] = wires[1, 3]
    betas, gammas = [123456789, 987654321], [1111, 2222]
    out = O.zs_partial_products(sh, wires, sig, k_is, betas, gammas)
    nc, npp, R, chunk = 2, sh.num_partial_products, sh.num_routed_wires, sh.quotient_degree_factor
    for c in range(nc):
        Z = [int(v) for v in out[c]]
        PP = [[int(v) for v in out[nc + c * npp + t]] for t in range(npp)]
        assert Z[0] == 1
        for i in range(n):
            x = pow(omega, i, P)
            q = []
            for j in range(R):
                w = int(wires[j, i])
                num = (w + betas[c] * k_is[j] % P * x + gammas[c]) % P
                den = (w + betas[c] * int(sig[j, i]) + gammas[c]) % P
                q.append(num * pow(den, P - 2, P) % P)
            acc = Z[i]
            for t in range(npp + 1):
                for j in range(t * chunk, min(R, (t + 1) * chunk)):
                    acc = acc * q[j] % P
                if t < npp:
                    assert PP[t][i] == acc           # partial product t
            assert acc == Z[(i + 1) % n]              # Z(g x); wraps to Z(1) = 1 at the last row
    wires[6, 20] = (int(wires[6, 20]) + 1) % P        # break the copy constraint
    out2 = O.zs_partial_products(sh, wires, sig, k_is, betas, gammas)
    assert not (out2 == out).all()
