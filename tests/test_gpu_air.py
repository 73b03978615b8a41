"""GPU parity of the generic AIR machinery (include/cityprover.h cp_air_* / cp_cubic_batch_inverse_dev / cp_column_prefix_sum_dev /
cp_stark_prove; SURVEY.md §8(a) A13, §8(f) N3 second slice) against the CPU oracle (oracle/stark_air.c), through the C ABI:
  * the device interpreter on seeded random constraint programs — >= 10^4 ops over 418 + 912 columns
    (city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:55-79) at 2^10 and 2^14 rows, and small shapes with 1-4
    challenges, several rates and quotient degrees, traces shorter than a wave: coefficients and cap of the committed quotient;
  * map programs, the batched cubic inversion and the column prefix sum;
  * the toy AIR WITH a lookup proved end to end: bytes == the oracle's prover, both verifiers accept both, cheating traces and
    tampered proofs are refused by cp_stark_verify;
  * cp_stark_prove at the SHA-256 STARK's width (418 + 912 columns, a 10^4-op constraint program) — bytes == oracle.
Parity of A13 itself stays UNPINNED (smartgadget.rs:505-513 asserts digests only; the AIR lives in an absent crate)."""
import os

import numpy as np
import pytest

import air_programs as A
import oracle_lib as O

pytestmark = pytest.mark.gpu
P = O.P


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    O.lib().or_set_threads(min(16, os.cpu_count() or 1))
    yield p
    O.lib().or_set_threads(1)
    p.close()


def test_cubic_inverse_and_prefix_sum_match_oracle(prover):
    import cityprover
    rng = np.random.default_rng(3)
    for count, n in ((1, 1), (1, 5), (8, 64), (9, 300), (20, 1000), (3, 1 << 14), (304, 1 << 10)):
        for m in (A.CUBIC_MODULUS, (1, 1)):
            cols = rng.integers(0, P, (3 * count, n), dtype=np.uint64)
            cols[0:3, n // 2] = 0                                   # the zero element -> 0
            got = cityprover.cubic_batch_inverse(prover, m, cols)
            assert (got == O.cubic_batch_inverse(m, cols)).all(), (count, n, m)
    for k, n in ((1, 1), (3, 7), (5, 256), (2, 257), (912, 1 << 10), (7, 1 << 16), (3, 100000)):
        cols = rng.integers(0, P, (k, n), dtype=np.uint64)
        for ex in (False, True):
            assert (cityprover.column_prefix_sum(prover, cols, ex) == O.column_prefix_sum(cols, ex)).all(), (k, n, ex)


def test_map_programs_match_oracle(prover):
    import cityprover
    _, ma, mb = A.lookup_programs()
    rng = np.random.default_rng(4)
    beta = rng.integers(0, P, 3, dtype=np.uint64)
    for n in (1, 3, 64, 65, 1000, 1 << 12):
        cols = rng.integers(0, P, (22, n), dtype=np.uint64)
        for b in (ma, mb):
            g, o = b.gpu(prover), b.oracle()
            try:
                assert (cityprover.air_map(prover, g, cols, challenges=beta) == o.map(cols, challenges=beta)).all(), n
            finally:
                g.close()
    # next-row loads wrap around, INV maps 0 to 0, columns that are not stored stay as they were
    b = A.Builder(A.MAP, 2, n_public=1, n_out_columns=3)
    b.store(0, b.add(b.next(0), b.public(0)))
    b.store(2, b.inv(b.sub(b.local(1), b.local(0))))
    cols = rng.integers(0, P, (2, 200), dtype=np.uint64)
    cols[1, 17] = cols[0, 17]
    g, o = b.gpu(prover), b.oracle()
    got = cityprover.air_map(prover, g, cols, publics=[5])
    g.close()
    assert (got == o.map(cols, publics=[5])).all() and got[2, 17] == 0 and (got[1] == 0).all()


def quotient_case(prover, b, ks, db, rb, q, n_alphas, ch, seed):
    """commit random traces, run the program on both sides, compare the committed quotient"""
    import cityprover
    rng = np.random.default_rng(seed)
    n = 1 << db
    traces = [rng.integers(0, P, (k, n), dtype=np.uint64) for k in ks]
    pub, glo, cha = (rng.integers(0, P, k, dtype=np.uint64) for k in (b.n_public, b.n_global, b.n_challenge))
    alphas = rng.integers(0, P, n_alphas, dtype=np.uint64)
    G = [cityprover.PolyBatch(prover, t, rb, ch) for t in traces]
    Ob = [O.Batch(t, rb, ch) for t in traces]
    g, o = b.gpu(prover), b.oracle()
    try:
        want = O.air_quotient(o, Ob, q, alphas, pub, glo, cha)
        Q = cityprover.air_quotient_commit(prover, g, G, q, alphas, pub, glo, cha)
        try:
            assert Q.k == n_alphas << q and Q.degree_bits == db and Q.rate_bits == rb and Q.cap_height == ch
            got = Q.coeffs()
            assert (got == want).all(), "quotient coefficients differ"
            oq = O.Batch(want, rb, ch, True)
            assert (Q.cap() == oq.cap()).all()
            oq.close()
        finally:
            Q.close()
        return g.info()
    finally:
        g.close()
        for x in G + Ob:
            x.close()


@pytest.mark.parametrize("seed", range(12))
def test_quotient_commit_matches_oracle_on_small_random_programs(prover, seed):
    rng = np.random.default_rng(1000 + seed)
    db = int(rng.integers(2, 9))                  # 4 .. 256 rows: traces shorter than a wave included
    rb = int(rng.integers(1, 4))
    q = int(rng.integers(1, rb + 1))
    n_alphas = int(rng.integers(1, 5))
    ks = [int(rng.integers(1, 9)) for _ in range(int(rng.integers(1, 4)))]
    max_degree = (1 << q) + 1
    b = A.random_program(seed, sum(ks), int(rng.integers(20, 800)), n_public=int(rng.integers(0, 3)), n_global=int(rng.integers(0, 3)),
                         n_challenge=int(rng.integers(0, 4)), max_degree=max_degree)
    info = quotient_case(prover, b, ks, db, rb, q, n_alphas, int(rng.integers(0, min(db + rb, 4) + 1)), seed)
    assert info["max_constraint_degree"] <= max_degree


@pytest.mark.parametrize("log_rows,seed,kind", [(10, 21, "gadgets"), (14, 22, "gadgets"), (10, 23, "tangle"), (14, 25, "tangle"), (12, 24, "tangle-far")])
def test_quotient_commit_at_the_sha256_stark_width(prover, log_rows, seed, kind):
    """>= 10^4 ops over 418 + 912 columns in two oracles, rate_bits 1 (starky's fast configuration: UPSTREAM-MEMORY), degree-3
    constraints, 2 challenges. "gadgets": the shape of an instruction-list AIR (shallow expressions, little sharing); "tangle": one
    connected web of values where every constraint reaches far back (the compiler then cuts coarser instead of recomputing the web
    in every segment), "tangle-far" with half of all operands from anywhere in the program (long-lived temporaries: slots spill)."""
    if kind == "gadgets":
        b = A.gadget_program(seed, 418 + 912, 10500, max_degree=3)
    else:
        b = A.random_program(seed, 418 + 912, 10500, max_degree=3, far=0.5 if kind == "tangle-far" else 0.1)
    info = quotient_case(prover, b, [418, 912], log_rows, 1, 1, 2, 4, seed)
    assert info["n_ops"] >= 10000 and info["n_constraints"] > 100
    print("rows 2^%d: %s" % (log_rows, info))


def test_quotient_commit_refuses_bad_arguments(prover):
    import cityprover
    b = A.Builder(A.CONSTRAINTS, 3, n_challenge=1)
    x = b.mul(b.mul(b.local(0), b.local(1)), b.mul(b.local(2), b.challenge(0)))
    b.assert_zero(x)
    g = b.gpu(prover)
    t = O.splitmix64_felts(3, 3 * 16).reshape(3, 16)
    T = cityprover.PolyBatch(prover, t, 1, 2)
    T2 = cityprover.PolyBatch(prover, t[:2], 1, 2)
    try:
        with pytest.raises(cityprover.CityProverError, match="degree 3 exceeds"):
            cityprover.air_quotient_commit(prover, g, [T], 0, [1], challenges=[2])
        with pytest.raises(cityprover.CityProverError, match="columns"):
            cityprover.air_quotient_commit(prover, g, [T2], 1, [1], challenges=[2])
        with pytest.raises(cityprover.CityProverError, match="out of range"):
            cityprover.air_quotient_commit(prover, g, [T], 2, [1], challenges=[2])
        with pytest.raises(cityprover.CityProverError, match="canonical"):
            cityprover.air_quotient_commit(prover, g, [T], 1, [P], challenges=[2])
        with pytest.raises(cityprover.CityProverError, match="NULL"):
            cityprover.air_quotient_commit(prover, g, [T], 1, [1])
        Q = cityprover.air_quotient_commit(prover, g, [T], 1, [1], challenges=[2])   # and the handles are still good
        Q.close()
        with pytest.raises(cityprover.CityProverError, match="earlier value"):
            cityprover.AirProgram(prover, A.CONSTRAINTS, [(A.ADD, 0, 0, 0)], n_columns=1)
        m = A.Builder(A.MAP, 3, n_out_columns=1)
        m.store(0, m.local(0))
        gm = m.gpu(prover)
        with pytest.raises(cityprover.CityProverError, match="CP_AIR_CONSTRAINTS"):
            cityprover.air_quotient_commit(prover, gm, [T], 1, [1])
        gm.close()
    finally:
        g.close()
        T.close()
        T2.close()


def lookup_descs(prover, db, rb=1, ch=2, pow_bits=5, nq=12, arity=(2,)):
    import cityprover
    c, ma, mb = A.lookup_programs()
    gp = [x.gpu(prover) for x in (c, ma, mb)]
    op = [x.oracle() for x in (c, ma, mb)]
    gd = cityprover.stark_desc(db, 1, 2, cityprover.fri_params(db, rb, ch, pow_bits, nq, arity), A.LOOKUP_K0, gp[0], A.LOOKUP_K1, 3,
                               steps=A.lookup_steps(gp[1], gp[2]))
    od = O.stark_desc(db, 1, 2, O.fri_params(db, rb, ch, pow_bits, nq, arity), A.LOOKUP_K0, op[0], A.LOOKUP_K1, 3, steps=A.lookup_steps(op[1], op[2]))
    return gd, od, gp


@pytest.mark.parametrize("db,device_transcript", [(4, 0), (7, 0), (10, 1)])
def test_toy_air_with_a_lookup_end_to_end(prover, db, device_transcript):
    import cityprover
    (gd, gkeep), (od, okeep), progs = lookup_descs(prover, db)
    trace = A.lookup_trace(1 << db)
    prover.set_device_transcript(device_transcript)
    try:
        oc = O.challenger_new()
        O.challenger_observe(oc, [1, 2, 3, 4, 5])
        want = O.stark_prove(od, trace, oc)
        gc = cityprover.ChallengerState()
        gc.observe([1, 2, 3, 4, 5])
        got = cityprover.stark_prove(prover, gd, trace, gc)
        assert got == want
        assert gc.as_tuple() == O.challenger_tuple(oc)
        # both verifiers accept, and end where the prover ended
        v = cityprover.ChallengerState()
        v.observe([1, 2, 3, 4, 5])
        cityprover.stark_verify(gd, v, got)
        assert v.as_tuple() == gc.as_tuple()
        w = O.challenger_new()
        O.challenger_observe(w, [1, 2, 3, 4, 5])
        assert O.stark_verify(od, w, got) == 0
        # another transcript prefix, tampered bytes, truncation: refused, the caller's transcript untouched
        fresh = cityprover.ChallengerState()
        with pytest.raises(cityprover.CityProverError):
            cityprover.stark_verify(gd, fresh, got)
        assert fresh.as_tuple() == cityprover.ChallengerState().as_tuple()
        rng = np.random.default_rng(db)
        for off in [21, 500, len(got) // 2, len(got) - 9] + [int(x) for x in rng.integers(0, len(got), 12)]:
            bad = bytearray(got)
            bad[off] ^= 1 << int(rng.integers(0, 8))
            v = cityprover.ChallengerState()
            v.observe([1, 2, 3, 4, 5])
            with pytest.raises(cityprover.CityProverError):
                cityprover.stark_verify(gd, v, bytes(bad))
        v = cityprover.ChallengerState()
        v.observe([1, 2, 3, 4, 5])
        with pytest.raises(cityprover.CityProverError, match="malformed"):
            cityprover.stark_verify(gd, v, got[:300])
        # a looked-up value that is not in the table / a broken transition: proved all the same, refused at zeta
        for cheat in ("value", "fib"):
            gc2 = cityprover.ChallengerState()
            p2 = cityprover.stark_prove(prover, gd, A.lookup_trace(1 << db, cheat=cheat), gc2)
            with pytest.raises(cityprover.CityProverError, match="quotient identity fails at zeta"):
                cityprover.stark_verify(gd, cityprover.ChallengerState(), p2)
        # an injected proof-of-work witness gives the oracle's bytes too
        oc = O.challenger_new()
        gc = cityprover.ChallengerState()
        assert cityprover.stark_prove(prover, gd, trace, gc, pow_override=77) == O.stark_prove(od, trace, oc, pow_override=77)
        # a trace element >= p (the scan runs on the device, behind the upload): refused, the transcript untouched, and the next
        # call with a good trace gives the same bytes as before (the flag does not stick)
        for where in ((0, 0), (A.LOOKUP_K0 - 1, (1 << db) - 1)):
            badt = np.array(trace, dtype=np.uint64, copy=True)
            badt[where] = O.P + 5
            gc3 = cityprover.ChallengerState()
            gc3.observe([1, 2, 3, 4, 5])
            before = gc3.as_tuple()
            with pytest.raises(cityprover.CityProverError, match="not canonical"):
                cityprover.stark_prove(prover, gd, badt, gc3)
            assert gc3.as_tuple() == before
        gc = cityprover.ChallengerState()
        gc.observe([1, 2, 3, 4, 5])
        assert cityprover.stark_prove(prover, gd, trace, gc) == want
    finally:
        prover.set_device_transcript(-1)
        for g in progs:
            g.close()


def test_stark_prove_bytes_at_the_sha256_stark_width(prover):
    """the whole prover at 418 + 912 columns, 2^10 rows, a 10^4-op constraint program, the extended columns filled by a map
    program, 304 cubic inversions and prefix sums of all 912 columns; 84 queries, 16-bit PoW, arity 16 (the shape of
    tests/test_gpu_fri_generic.py::stark_spec). The random program is not SATISFIED by the random trace — the prover does not
    care (its quotient is simply not low-degree and no verifier would accept): what is held here is that every byte the device
    produces is the byte the oracle produces."""
    import cityprover
    db, k0, k1 = 10, 418, 912
    cons = A.gadget_program(77, k0 + k1, 10500, n_public=4, n_global=0, n_challenge=6, max_degree=3)
    m = A.Builder(A.MAP, k0 + k1, n_public=4, n_challenge=6, n_out_columns=k1)
    ch = [m.challenge(i) for i in range(6)]
    for j in range(k1):                                   # every extended column from a couple of trace columns and a challenge
        v = m.add(m.mul(m.local(j % k0), ch[j % 6]), m.next((7 * j + 1) % k0))
        m.store(j, m.sub(v, m.public(j % 4)) if j % 3 else v)
    steps_of = lambda mp: [("map", mp), ("cubic_inverse", 0, 304, A.CUBIC_MODULUS), ("prefix_sum", 0, k1, False)]
    gp = [cons.gpu(prover), m.gpu(prover)]
    op = [cons.oracle(), m.oracle()]
    arity = (4,)
    gd, gk = cityprover.stark_desc(db, 1, 2, cityprover.fri_params(db, 1, 4, 16, 84, arity), k0, gp[0], k1, 6, n_public=4, steps=steps_of(gp[1]))
    od, ok = O.stark_desc(db, 1, 2, O.fri_params(db, 1, 4, 16, 84, arity), k0, op[0], k1, 6, n_public=4, steps=steps_of(op[1]))
    rng = np.random.default_rng(8)
    trace = rng.integers(0, P, (k0, 1 << db), dtype=np.uint64)
    pub = rng.integers(0, P, 4, dtype=np.uint64)
    try:
        oc, gc = O.challenger_new(), cityprover.ChallengerState()
        want = O.stark_prove(od, trace, oc, publics=pub)
        got = cityprover.stark_prove(prover, gd, trace, gc, publics=pub)
        assert got == want
        assert gc.as_tuple() == O.challenger_tuple(oc)
    finally:
        for g in gp:
            g.close()


def random_map_program(rng, k0, k1, n_public, n_global, n_challenge, inv_cols):
    """a map program that stores every extended column: products / sums of a few trace columns, publics, globals and round
    challenges; the columns in `inv_cols` hold an inverse (INV: zero maps to zero)"""
    m = A.Builder(A.MAP, k0 + k1, n_public=n_public, n_global=n_global, n_challenge=n_challenge, n_out_columns=k1)
    for j in range(k1):
        v = m.local(int(rng.integers(0, k0))) if rng.random() < 0.7 else m.next(int(rng.integers(0, k0)))
        for _ in range(int(rng.integers(0, 3))):
            kind = int(rng.integers(0, 5))
            other = (m.local(int(rng.integers(0, k0))) if kind == 0 else m.public(int(rng.integers(0, n_public))) if kind == 1 and n_public else
                     m.glob(int(rng.integers(0, n_global))) if kind == 2 and n_global else
                     m.challenge(int(rng.integers(0, n_challenge))) if kind == 3 and n_challenge else m.const(int(rng.integers(1, P, dtype=np.uint64))))
            v = (m.mul, m.add, m.sub)[int(rng.integers(0, 3))](v, other)
        if j in inv_cols:
            v = m.inv(v)
        m.store(j, v)
    return m


@pytest.mark.parametrize("seed", range(10))
def test_stark_prove_bytes_on_small_random_shapes(prover, seed):
    """the whole prover == the oracle's on seeded small shapes: with and without an extended round, 1 - 4 alphas, quotient degree
    bits 1 - 2 at rates 2 - 8, publics / globals / round challenges present or not, one to three steps of every kind in any order,
    cap heights from 0 up, with and without FRI reduction layers, host and device transcripts, an injected PoW witness."""
    import cityprover
    rng = np.random.default_rng(9000 + seed)
    db = int(rng.integers(3, 9))
    rb = int(rng.integers(1, 4))
    q = int(rng.integers(1, min(rb, 2) + 1))
    na = int(rng.integers(1, 5))
    k0 = int(rng.integers(1, 12))
    k1 = 0 if seed % 3 == 0 else 3 * int(rng.integers(1, 4))
    n_pub, n_glob = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    n_rch = int(rng.integers(1, 4)) if k1 else 0
    ch = int(rng.integers(0, min(db + rb, 4) + 1))
    cons = A.random_program(seed + 50, k0 + k1, int(rng.integers(30, 400)), n_public=n_pub, n_global=n_glob, n_challenge=n_rch, max_degree=(1 << q) + 1)
    steps_g, steps_o, progs = [], [], []
    gcons, ocons = cons.gpu(prover), cons.oracle()
    progs.append(gcons)
    if k1:
        kinds = ["map"] + [("cubic_inverse", "prefix_sum", "map")[int(x)] for x in rng.integers(0, 3, int(rng.integers(0, 3)))]
        for kind in kinds:
            if kind == "map":
                m = random_map_program(rng, k0, k1, n_pub, n_glob, n_rch, set(int(x) for x in rng.integers(0, k1, 2)))
                g, o = m.gpu(prover), m.oracle()
                progs.append(g)
                steps_g.append(("map", g))
                steps_o.append(("map", o))
            elif kind == "cubic_inverse":
                cnt = int(rng.integers(1, k1 // 3 + 1))
                first = 3 * int(rng.integers(0, k1 // 3 - cnt + 1))
                st = ("cubic_inverse", first, cnt, A.CUBIC_MODULUS if rng.random() < 0.5 else (int(rng.integers(1, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64))))
                steps_g.append(st)
                steps_o.append(st)
            else:
                cnt = int(rng.integers(1, k1 + 1))
                st = ("prefix_sum", int(rng.integers(0, k1 - cnt + 1)), cnt, bool(rng.integers(0, 2)))
                steps_g.append(st)
                steps_o.append(st)
    arity = ()
    d = db
    while d > 3 and d + rb - 2 >= ch and rng.random() < 0.7 and len(arity) < 3:
        a = int(rng.integers(1, 3))
        if d - a < 1 or d - a + rb < ch:
            break
        arity += (a,)
        d -= a
    pow_bits, nq = int(rng.integers(0, 9)), int(rng.integers(1, 12))
    gd, gk = cityprover.stark_desc(db, q, na, cityprover.fri_params(db, rb, ch, pow_bits, nq, arity), k0, gcons, k1, n_rch, n_public=n_pub, n_global=n_glob, steps=steps_g)
    od, okk = O.stark_desc(db, q, na, O.fri_params(db, rb, ch, pow_bits, nq, arity), k0, ocons, k1, n_rch, n_public=n_pub, n_global=n_glob, steps=steps_o)
    trace = rng.integers(0, P, (k0, 1 << db), dtype=np.uint64)
    pub = rng.integers(0, P, n_pub, dtype=np.uint64)
    glob = rng.integers(0, P, n_glob, dtype=np.uint64)
    prefix = rng.integers(0, P, int(rng.integers(0, 11)), dtype=np.uint64)
    prover.set_device_transcript(seed % 2)
    try:
        for pow_override in (None, 12345):
            oc, gc = O.challenger_new(), cityprover.ChallengerState()
            if prefix.size:
                O.challenger_observe(oc, prefix)
                gc.observe(prefix)
            want = O.stark_prove(od, trace, oc, publics=pub, globals_=glob, pow_override=pow_override)
            got = cityprover.stark_prove(prover, gd, trace, gc, publics=pub if n_pub else None, globals_=glob if n_glob else None, pow_override=pow_override)
            assert got == want, (db, rb, q, na, k0, k1, ch, arity, [s[0] for s in steps_g])
            assert gc.as_tuple() == O.challenger_tuple(oc)
    finally:
        prover.set_device_transcript(-1)
        for g in progs:
            g.close()
