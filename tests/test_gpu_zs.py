"""GPU parity for A7 (Z / partial products of the permutation argument), batched over proofs."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
P = O.P


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    yield p
    p.close()


@pytest.mark.parametrize("db,R,W,chunk,nc,B", [(5, 8, 12, 4, 2, 3), (12, 80, 135, 8, 2, 2), (7, 10, 10, 3, 3, 1),
                                               (9, 80, 135, 8, 1, 4)])
def test_zs_partial_products_match_oracle(prover, db, R, W, chunk, nc, B):
    import cityprover as cp
    npp = (R + chunk - 1) // chunk - 1
    kw = dict(degree_bits=db, num_constants=3, num_routed_wires=R, num_wires=W, num_challenges=nc,
              num_partial_products=npp, quotient_degree_factor=chunk, rate_bits=3, cap_height=2, pow_bits=4,
              num_query_rounds=3, arity_bits=(2,))
    sh = cp.standard_recursion_shape(**kw)
    osh = O.standard_shape(degree_bits=db, num_wires=W, num_routed=R, num_constants=3, num_challenges=nc,
                           num_partial_products=npp, quotient_degree_factor=chunk, rate_bits=3, cap_height=2,
                           pow_bits=4, num_query_rounds=3, arity_bits=(2,))
    n = 1 << db
    k_is = [pow(7, j, P) for j in range(R)]
    circs, css = [], []
    for c in range(2):
        cs = O.splitmix64_felts(50 + c, (3 + R) * n).reshape(3 + R, n)
        css.append(cs)
        circs.append(cp.Circuit(prover, sh, [c, 0, 0, 0], cs))
    pick = [circs[i % 2] for i in range(B)]
    wires = O.splitmix64_felts(99, B * W * n).reshape(B, W, n)
    betas = O.splitmix64_felts(7, B * nc).reshape(B, nc)
    gammas = O.splitmix64_felts(8, B * nc).reshape(B, nc)
    dw, dout = prover.to_device(wires), prover.alloc(B * nc * (1 + npp) * n)
    cp.zs_partial_products_dev(prover, pick, dw.ptr, betas, gammas, dout.ptr)
    got = dout.download().reshape(B, nc * (1 + npp), n)
    for b in range(B):
        want = O.zs_partial_products(osh, wires[b], css[b % 2][3:], k_is, betas[b], gammas[b])
        assert (got[b] == want).all(), f"proof {b}"
    dw.free(); dout.free()
    for c in circs:
        c.close()


def test_custom_k_is_and_bad_shape(prover):
    import cityprover as cp
    sh = cp.standard_recursion_shape(degree_bits=5, num_constants=2, num_routed_wires=8, num_wires=10, num_challenges=1,
                                     num_partial_products=1, quotient_degree_factor=4, rate_bits=3, cap_height=1,
                                     pow_bits=2, num_query_rounds=2, arity_bits=(1,))
    n = 32
    cs = O.splitmix64_felts(1, 10 * n).reshape(10, n)
    k_is = [pow(3, j + 1, P) for j in range(8)]
    circ = cp.Circuit(prover, sh, [0] * 4, cs, k_is=k_is)
    wires = O.splitmix64_felts(2, 10 * n).reshape(1, 10, n)
    dw, dout = prover.to_device(wires), prover.alloc(2 * n)
    cp.zs_partial_products_dev(prover, [circ], dw.ptr, [[5]], [[6]], dout.ptr)
    osh = O.standard_shape(degree_bits=5, num_wires=10, num_routed=8, num_constants=2, num_challenges=1,
                           num_partial_products=1, quotient_degree_factor=4, rate_bits=3, cap_height=1, pow_bits=2,
                           num_query_rounds=2, arity_bits=(1,))
    assert (dout.download().reshape(2, n) == O.zs_partial_products(osh, wires[0], cs[2:], k_is, [5], [6])).all()
    with pytest.raises(cp.CityProverError):
        cp.zs_partial_products_dev(prover, [circ], dw.ptr, [[P]], [[6]], dout.ptr)  # non-canonical challenge
    dw.free(); dout.free(); circ.close()
