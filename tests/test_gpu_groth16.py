"""cp_groth16_prove_bls12381 against the trapdoor of a setup generated here: a random satisfied
R1CS, its QAP evaluated at a known tau, the proving-key points as [log] G, and the three proof elements compared with
[expected log] G where the expected logs come from the Groth16 formulas in Python integers (which are also checked to
satisfy the verification equation in the exponent). No pairing needed; parity with gnark is unpinned by the reference."""
import ctypes

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    yield p
    p.close()


def limbs(v, n):
    return [(int(v) >> (64 * i)) & (2**64 - 1) for i in range(n)]


def g1_rows(points):
    return np.array([limbs(P[0], 6) + limbs(P[1], 6) for P in points], dtype=np.uint64)


def g2_rows(points):
    return np.array([limbs(P[0][0], 6) + limbs(P[0][1], 6) + limbs(P[1][0], 6) + limbs(P[1][1], 6) for P in points], dtype=np.uint64)


@pytest.mark.parametrize("log_n,n_pub,n_in", [(3, 2, 3), (5, 3, 6)])
def test_groth16_proof_matches_the_trapdoor(prover, log_n, n_pub, n_in):
    import cityprover as cp
    _, r, G1 = O.bls_constants()
    G2 = O.bls_g2_generator()
    rng = np.random.default_rng(50 + log_n)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (r - 1) + 1
    n = 1 << log_n
    # ---- a satisfied R1CS: wire 0 = 1, then public inputs, private inputs, and one product wire per constraint ----
    w = [1] + [rnd() for _ in range(n_pub - 1 + n_in)]
    A, B, C = [], [], []
    for j in range(n):
        avail = len(w)
        ra = {int(i): rnd() for i in rng.choice(avail, size=min(3, avail), replace=False)}
        rb = {int(i): rnd() for i in rng.choice(avail, size=min(2, avail), replace=False)}
        out = sum(c * w[i] for i, c in ra.items()) % r * (sum(c * w[i] for i, c in rb.items()) % r) % r
        w.append(out)
        A.append(ra); B.append(rb); C.append({avail: 1})
    m = len(w)
    dot = lambda row: sum(c * w[i] for i, c in row.items()) % r
    a_ev, b_ev, c_ev = [dot(x) for x in A], [dot(x) for x in B], [dot(x) for x in C]
    assert all(x * y % r == z for x, y, z in zip(a_ev, b_ev, c_ev))
    # ---- trapdoor and QAP at tau ----
    tau, alpha, beta, delta = rnd(), rnd(), rnd(), rnd()
    omega = pow(7, (r - 1) >> log_n, r)
    zt = (pow(tau, n, r) - 1) % r
    L = [zt * pow(n, -1, r) % r * pow(omega, j, r) % r * pow((tau - pow(omega, j, r)) % r, -1, r) % r for j in range(n)]
    col = lambda M, i: sum(M[j].get(i, 0) * L[j] for j in range(n)) % r
    u, v, ww = [col(A, i) for i in range(m)], [col(B, i) for i in range(m)], [col(C, i) for i in range(m)]
    dinv = pow(delta, -1, r)
    k_log = [(beta * u[i] + alpha * v[i] + ww[i]) * dinv % r for i in range(n_pub, m)]
    z_log = [pow(tau, j, r) * zt % r * dinv % r for j in range(n - 1)]
    # ---- h(tau) from the polynomial identity (independent of the library's NTT route) ----
    interp = lambda ev: sum(e * l for e, l in zip(ev, L)) % r
    h_tau = (interp(a_ev) * interp(b_ev) - interp(c_ev)) * pow(zt, -1, r) % r
    rr, ss = rnd(), rnd()
    a_log = (alpha + sum(x * y for x, y in zip(w, u)) + rr * delta) % r
    b_log = (beta + sum(x * y for x, y in zip(w, v)) + ss * delta) % r
    c_log = (sum(x * y for x, y in zip(w[n_pub:], k_log)) + h_tau * zt * dinv + ss * a_log + rr * b_log - rr * ss * delta) % r
    # the verification equation in the exponent: a b = alpha beta + sum_pub w_i (beta u_i + alpha v_i + w_i) + c delta
    pub = sum(w[i] * (beta * u[i] + alpha * v[i] + ww[i]) for i in range(n_pub)) % r
    assert a_log * b_log % r == (alpha * beta + pub + c_log * delta) % r
    # ---- proving key as points ----
    def pts1(logs):
        return [O.bls_g1_mul(G1, x) if x else G1 for x in logs], np.array([0 if x else 1 for x in logs], np.uint8)
    pa, a_inf = pts1(u)
    pb1, b_inf = pts1(v)
    pb2 = [O.bls_g2_mul(G2, x) if x else G2 for x in v]
    pk_, _ = pts1(k_log)
    pz, _ = pts1(z_log)
    sets = [cp.G1Points(prover, g1_rows(pa)), cp.G1Points(prover, g1_rows(pb1)), cp.G2Points(prover, g2_rows(pb2)),
            cp.G1Points(prover, g1_rows(pk_)), cp.G1Points(prover, g1_rows(pz))]
    flags = lambda f: prover.to_device(np.frombuffer(np.concatenate([f, np.zeros(-len(f) % 8, np.uint8)]).tobytes(), np.uint64))
    d_ainf, d_binf = flags(a_inf), flags(b_inf)
    pk = cp.Groth16Pk()
    pk.n_wires, pk.n_private, pk.log_domain = m, m - n_pub, log_n
    pk.a_g1, pk.b_g1, pk.b_g2, pk.k_g1, pk.z_g1 = (s.buf.ptr for s in sets)
    pk.a_inf, pk.b_inf = d_ainf.ptr, d_binf.ptr
    for name, P in (("alpha_g1", O.bls_g1_mul(G1, alpha)), ("beta_g1", O.bls_g1_mul(G1, beta)), ("delta_g1", O.bls_g1_mul(G1, delta))):
        getattr(pk, name)[:] = limbs(P[0], 6) + limbs(P[1], 6)
    for name, P in (("beta_g2", O.bls_g2_mul(G2, beta)), ("delta_g2", O.bls_g2_mul(G2, delta))):
        getattr(pk, name)[:] = limbs(P[0][0], 6) + limbs(P[0][1], 6) + limbs(P[1][0], 6) + limbs(P[1][1], 6)
    to_dev = lambda vals: prover.to_device(np.array([limbs(x, 4) for x in vals], dtype=np.uint64))
    dw, da, db, dc = to_dev(w), to_dev(a_ev), to_dev(b_ev), to_dev(c_ev)
    A_pt, B_pt, C_pt = cp.groth16_prove(prover, pk, dw.ptr, da.ptr, db.ptr, dc.ptr, rr, ss)
    assert A_pt == O.bls_g1_mul(G1, a_log)
    assert B_pt == O.bls_g2_mul(G2, b_log)
    assert C_pt == O.bls_g1_mul(G1, c_log)
    for d in (dw, da, db, dc, d_ainf, d_binf):
        d.free()
    for s in sets:
        s.free()


@pytest.mark.parametrize("log_n", [12, 18, 20])
def test_groth16_proof_matches_the_trapdoor_at_the_sizes_it_is_timed(prover, log_n):
    """The tiled digit sort, the heavy-bucket chunks (60 % of the witness is 0 / 1) and the two MSM chains on two contexts only run
    at large sizes: the assembled (A, B, C) at 2^18 and 2^20 constraints against the closed-form discrete logarithms of the
    synthetic key (tools/bench_groth16.py `run(check=True)`: every key point is [a i + b] G, so the expected logs are sums over
    the witness and the quotient coefficients; the quotient itself is held against the polynomial identity at 2^18 by
    tests/test_gpu_fr_ntt.py)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_groth16
    res = bench_groth16.run(prover, log_n, reps=1, check=True)
    assert res["checked"] is True
    # the exact big-integer sums behind the check, against plain Python on a small slice
    rng = np.random.default_rng(3)
    k = rng.integers(0, 2**64, (3000, 4), dtype=np.uint64)
    ints = [sum(int(k[i, j]) << (64 * j) for j in range(4)) for i in range(3000)]
    assert bench_groth16.limbs_sum(k) == (sum(ints), sum(i * v for i, v in enumerate(ints)))
    assert bench_groth16.limbs_sum(k, 17, 2999) == (sum(ints[17:2999]), sum(i * v for i, v in enumerate(ints[17:2999])))


def test_groth16_prove_rejects_bad_arguments(prover):
    import cityprover as cp
    _, r, G1 = O.bls_constants()
    pk = cp.Groth16Pk()
    pk.n_wires, pk.n_private, pk.log_domain = 4, 2, 2
    d = prover.to_device(np.zeros((4, 4), np.uint64))
    with pytest.raises(cp.CityProverError, match="NULL point set"):
        cp.groth16_prove(prover, pk, d.ptr, d.ptr, d.ptr, d.ptr, 1, 1)
    pts = cp.G1Points.synthetic(prover, G1, 1, 1, 4)
    g2 = cp.G2Points.synthetic(prover, O.bls_g2_generator(), 1, 1, 4)
    pk.a_g1 = pk.b_g1 = pk.k_g1 = pk.z_g1 = pts.buf.ptr
    pk.b_g2 = g2.buf.ptr
    with pytest.raises(cp.CityProverError, match="canonical"):
        cp.groth16_prove(prover, pk, d.ptr, d.ptr, d.ptr, d.ptr, r, 1)          # r is not below the group order
    with pytest.raises(cp.CityProverError, match="not on the curve"):
        cp.groth16_prove(prover, pk, d.ptr, d.ptr, d.ptr, d.ptr, 1, 1)          # alpha = (0, 0)
    pk.log_domain = 40
    with pytest.raises(cp.CityProverError, match="log_domain"):
        cp.groth16_prove(prover, pk, d.ptr, d.ptr, d.ptr, d.ptr, 1, 1)
    d.free(); pts.free(); g2.free()
