"""CPU tests of the generic AIR machinery (second slice; include/cityprover.h cp_air_*):
  * the oracle's direct interpreter (oracle/stark_air.c) against constraints written once over an abstract field and evaluated
    in Python integers — the same definition that is RECORDED into the program (tests/air_programs.py);
  * the product's host half — analyser, segmenting compiler, slot allocation — and the instruction semantics its device
    interpreter shares (city-rollup_amd/csrc/air.h run_segment), executed on the CPU through tests/hostsim against the oracle on
    seeded random programs of 10^4 ops over 418 + 912 columns (smartgadget.rs:55-79), for 1 .. 256 segments;
  * the cubic-extension primitives and the prefix sum (oracle, and ext3.h on the host);
  * the oracle's whole STARK prover / verifier on the toy AIR with a lookup: accepted; a value outside the table, a broken
    transition and tampered proofs are refused; the quotient identity re-checked at zeta in Python over F_p^2.
Parity of A13 itself stays UNPINNED (no STARK vector in the reference: smartgadget.rs:505-513 asserts digests only)."""
import ctypes
import os
import struct
import sys

import numpy as np
import pytest

import air_programs as A
import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim"))
P = O.P


@pytest.fixture(scope="module")
def hs():
    import build as hb
    lib = ctypes.CDLL(hb.build())
    u64p, u32p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)
    lib.hs_air_point.argtypes = [ctypes.c_int, u32p, ctypes.c_size_t, u64p, ctypes.c_size_t, u32p, ctypes.c_uint32, ctypes.c_int] + [u64p] * 6 + [
        ctypes.c_int, u64p, u64p, u64p, u32p, ctypes.c_char_p]
    lib.hs_cubic.argtypes = [ctypes.c_int] + [u64p] * 4
    return lib


def hs_point(hs, b, want_segments, local, nxt, publics=(), globals_=(), challenges=(), alphas=(), sel=(0, 0, 0), prefetch=1):
    ops, consts = b.arrays()
    u64p, u32p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)
    dims = np.array([b.n_columns, b.n_public, b.n_global, b.n_challenge, b.n_out_columns], dtype=np.uint32)
    arrs = [O.arr(x) for x in (local, nxt, publics, globals_, challenges, alphas, sel)]
    acc = np.zeros(max(len(alphas), 1), np.uint64)
    stores = np.zeros(max(b.n_out_columns, 1), np.uint64)
    info = np.zeros(5, np.uint32)
    err = ctypes.create_string_buffer(256)
    rc = hs.hs_air_point(b.kind, ops.ctypes.data_as(u32p), len(ops), consts.ctypes.data_as(u64p), len(consts), dims.ctypes.data_as(u32p), want_segments, prefetch,
                         *[a.ctypes.data_as(u64p) for a in arrs[:6]], len(alphas), arrs[6].ctypes.data_as(u64p), acc.ctypes.data_as(u64p),
                         stores.ctypes.data_as(u64p), info.ctypes.data_as(u32p), err)
    if rc != 0:
        raise ValueError(err.value.decode())
    return [int(v) for v in acc[:len(alphas)]], stores[:b.n_out_columns], dict(zip(("segments", "slots", "instructions", "live", "max_degree"), (int(v) for v in info)))


def horner(values_kinds, alpha, sel):
    acc = 0
    for v, k in values_kinds:
        f = {A.ASSERT_ZERO: 1, A.ASSERT_ZERO_TRANSITION: sel[0], A.ASSERT_ZERO_FIRST_ROW: sel[1], A.ASSERT_ZERO_LAST_ROW: sel[2]}[k]
        acc = (acc * alpha + v * f) % P
    return acc


def sinks_of(b, vals):
    return [(int(vals[a]), op) for (op, a, _, _) in b.ops if A.ASSERT_ZERO <= op <= A.ASSERT_ZERO_LAST_ROW]


# ---- the toy AIR with a lookup: one definition, three interpretations ------------------------------------------------------
def lookup_extended(trace, beta):
    """the 15 extended columns in Python integers (cubic arithmetic from air_programs, inverses through the oracle's elementwise one)"""
    n = trace.shape[1]
    F = A.IntField
    m = A.CUBIC_MODULUS

    def cinv(x):
        out = np.zeros(3, np.uint64)
        O.lib().or_cubic_inverse(O.ptr(O.arr(m)), O.ptr(O.arr(x)), O.ptr(out))
        r = tuple(int(v) for v in out)
        assert A.cubic_mul(F, m, x, r) == (1, 0, 0)
        return r
    ext = np.zeros((A.LOOKUP_K1, n), dtype=np.uint64)
    run = (0, 0, 0)
    for i in range(n):
        e = [cinv(((beta[0] - int(trace[c, i])) % P, beta[1], beta[2])) for c in (3, 4, 5)]
        row = A.cubic_sub(F, A.cubic_add(F, e[0], e[1]), A.cubic_scale(F, e[2], int(trace[6, i])))
        for j, v in enumerate(e + [row, run]):
            ext[3 * j:3 * j + 3, i] = v
        run = A.cubic_add(F, run, row)
    return ext, run


def test_lookup_air_one_definition_three_ways():
    n = 32
    trace = A.lookup_trace(n)
    beta = (123456789, 987654321, 555)
    ext, total = lookup_extended(trace, beta)
    assert total == (0, 0, 0)                              # the log-derivative sums of values and table agree
    full = np.vstack([trace, ext])
    c, ma, mb = A.lookup_programs()
    oc = c.oracle()
    assert oc.check() == 0 and oc.num_constraints() == 6 + 9 + 3 + 9
    # (1) Python integers, row by row: every constraint holds where it must
    for i in range(n):
        loc, nxt = [int(v) for v in full[:, i]], [int(v) for v in full[:, (i + 1) % n]]
        got = []
        A.lookup_constraints(A.IntField, loc, nxt, beta, lambda v, when: got.append((v, when)))
        for v, when in got:
            if when == "all" or (when == "transition" and i < n - 1) or (when == "first" and i == 0) or (when == "last" and i == n - 1):
                assert v == 0, (i, when)
        # (2) the recorded program through the oracle's interpreter gives the very same values
        vals = oc.eval_row(loc, nxt, challenges=beta)
        assert [v for v, _ in sinks_of(c, vals)] == [v for v, _ in got]
    # on random rows too (nothing vanishes there), and over F_p^2
    rng = np.random.default_rng(1)
    for _ in range(5):
        loc = [int(v) for v in rng.integers(0, P, 22, dtype=np.uint64)]
        nxt = [int(v) for v in rng.integers(0, P, 22, dtype=np.uint64)]
        got = []
        A.lookup_constraints(A.IntField, loc, nxt, beta, lambda v, when: got.append(v))
        assert [v for v, _ in sinks_of(c, oc.eval_row(loc, nxt, challenges=beta))] == got
        le = [(int(a), int(b)) for a, b in rng.integers(0, P, (22, 2), dtype=np.uint64)]
        ne = [(int(a), int(b)) for a, b in rng.integers(0, P, (22, 2), dtype=np.uint64)]
        be = [(x, 0) for x in beta]
        gote = []
        A.lookup_constraints(A.ExtField, le, ne, be, lambda v, when: gote.append(v))
        out, kinds = oc.eval_ext(np.array(le, dtype=np.uint64), np.array(ne, dtype=np.uint64), challenges=np.array(be, dtype=np.uint64))
        assert [tuple(int(x) for x in r) for r in out] == gote
    # (3) the map programs + the two primitives reproduce the extended columns
    oa, ob = ma.oracle(), mb.oracle()
    assert oa.check() == 0 and ob.check() == 0
    e = np.zeros_like(ext)
    work = np.vstack([trace, e])
    e[:9] = oa.map(work, challenges=beta)[:9]
    e[:9] = O.cubic_batch_inverse(A.CUBIC_MODULUS, e[:9])
    work = np.vstack([trace, e])
    e[9:15] = ob.map(work, challenges=beta)[9:15]
    e[12:15] = O.column_prefix_sum(e[12:15], exclusive=True)
    assert (e == ext).all()


def test_cubic_and_prefix_sum_primitives(hs):
    rng = np.random.default_rng(2)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    for m in (A.CUBIC_MODULUS, (1, 1), (5, 0), (int(rng.integers(1, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64)))):
        for _ in range(20):
            a = tuple(int(v) for v in rng.integers(0, P, 3, dtype=np.uint64))
            b = tuple(int(v) for v in rng.integers(0, P, 3, dtype=np.uint64))
            want = A.cubic_mul(A.IntField, m, a, b)
            assert tuple(int(v) for v in O.cubic_mul(m, a, b)) == want
            out = np.zeros(3, np.uint64)
            hs.hs_cubic(0, O.arr(m).ctypes.data_as(u64p), O.arr(a).ctypes.data_as(u64p), O.arr(b).ctypes.data_as(u64p), out.ctypes.data_as(u64p))
            assert tuple(int(v) for v in out) == want
            hs.hs_cubic(1, O.arr(m).ctypes.data_as(u64p), O.arr(a).ctypes.data_as(u64p), O.arr(a).ctypes.data_as(u64p), out.ctypes.data_as(u64p))
            inv = tuple(int(v) for v in out)
            prod = A.cubic_mul(A.IntField, m, a, inv)
            # X^3 - m1 X - m0 may be reducible for an arbitrary m: then some a have no inverse and norm = 0 -> 0
            assert prod == (1, 0, 0) or inv == (0, 0, 0)
        cols = rng.integers(0, P, (6, 17), dtype=np.uint64)
        cols[0:3, 4] = 0                                     # the zero element maps to zero
        inv = O.cubic_batch_inverse(m, cols)
        assert (inv[0:3, 4] == 0).all()
        for e in range(2):
            for i in range(17):
                a = tuple(int(v) for v in cols[3 * e:3 * e + 3, i])
                r = tuple(int(v) for v in inv[3 * e:3 * e + 3, i])
                assert r == (0, 0, 0) or A.cubic_mul(A.IntField, m, a, r) == (1, 0, 0)
    cols = rng.integers(0, P, (4, 100), dtype=np.uint64)
    inc, exc = O.column_prefix_sum(cols), O.column_prefix_sum(cols, exclusive=True)
    for c in range(4):
        run = 0
        for i in range(100):
            assert int(exc[c, i]) == run
            run = (run + int(cols[c, i])) % P
            assert int(inc[c, i]) == run


# ---- the product's compiler + instruction semantics on the host, against the oracle ----------------------------------------
@pytest.mark.parametrize("seed,n_columns,n_ops", [(0, 12, 300), (1, 40, 2000), (2, 418 + 912, 10000), (3, 418 + 912, 12000), (4, 5, 60), (5, 418 + 912, 10500),
                                                  (6, 30, 3000)])
def test_compiled_program_equals_the_direct_interpreter(hs, seed, n_columns, n_ops):
    if seed >= 5:
        b = A.gadget_program(seed, n_columns, n_ops)       # the shape of an instruction-list AIR
    else:
        b = A.random_program(seed, n_columns, n_ops, far=0.02 if seed == 3 else 0.1)   # one connected web of values
    op = b.oracle()
    assert op.check() == 0
    rng = np.random.default_rng(100 + seed)
    loc, nxt = rng.integers(0, P, n_columns, dtype=np.uint64), rng.integers(0, P, n_columns, dtype=np.uint64)
    pub, glo, cha = (rng.integers(0, P, k, dtype=np.uint64) for k in (b.n_public, b.n_global, b.n_challenge))
    alphas = [int(v) for v in rng.integers(0, P, 2, dtype=np.uint64)]
    sel = [int(v) for v in rng.integers(0, P, 3, dtype=np.uint64)]
    vals = op.eval_row(loc, nxt, pub, glo, cha)
    want = [horner(sinks_of(b, vals), a, sel) for a in alphas]
    seen = set()
    for segs in (1, 2, 7, 32, 256):
        got, _, info = hs_point(hs, b, segs, loc, nxt, pub, glo, cha, alphas, sel)
        assert got == want, (segs, info)
        got0, _, info0 = hs_point(hs, b, segs, loc, nxt, pub, glo, cha, alphas, sel, prefetch=0)   # column operands loaded at their use
        assert got0 == want and info0["instructions"] <= info["instructions"], (segs, info0)
        assert info["segments"] <= max(1, min(segs, op.num_constraints()))
        seen.add(info["segments"])
        if segs == 1:
            one = info
    assert len(seen) > 1 or op.num_constraints() < 2
    # dead values are dropped, loads and uniform values cost no instruction, temporaries are few
    # dead values are dropped; uniform values cost no instruction; a column costs one per prefetch (again when it is needed after its
    # window): never more than the live ops plus a reload here and there
    assert one["live"] <= len(b.ops) and one["instructions"] <= 1.25 * one["live"]
    assert one["slots"] < 0x3FFF and one["max_degree"] <= 3
    if n_ops >= 10000:
        print("program %d: %d ops, %d live, %d instructions, %d slots in one segment" % (seed, len(b.ops), one["live"], one["instructions"], one["slots"]))


def test_map_program_on_the_host(hs):
    _, ma, mb = A.lookup_programs()
    rng = np.random.default_rng(9)
    loc, nxt = rng.integers(1, P, 22, dtype=np.uint64), rng.integers(1, P, 22, dtype=np.uint64)
    beta = [int(v) for v in rng.integers(0, P, 3, dtype=np.uint64)]
    for b in (ma, mb):
        ob = b.oracle()
        vals = ob.eval_row(loc, nxt, challenges=beta)
        want = {a: int(vals[v]) for (op, a, v, _) in b.ops if op == A.STORE}
        _, stores, _ = hs_point(hs, b, 1, loc, nxt, challenges=beta)
        for col, v in want.items():
            assert int(stores[col]) == v
    # INV inside a map program
    b = A.Builder(A.MAP, 2, n_out_columns=2)
    x = b.local(0)
    b.store(0, b.inv(b.add(x, b.next(1))))
    b.store(1, b.inv(b.sub(x, x)))              # 0 -> 0
    _, stores, _ = hs_point(hs, b, 1, [5, 0], [0, 9])
    assert int(stores[0]) * 14 % P == 1 and int(stores[1]) == 0
    assert int(b.oracle().eval_row([5, 0], [0, 9])[3]) == int(stores[0])


def test_analyser_refuses_malformed_programs(hs):
    def bad(build, match, kind=A.CONSTRAINTS, n_out=0):
        b = A.Builder(kind, 3, 1, 1, 1, n_out)
        build(b)
        with pytest.raises(ValueError, match=match):
            hs_point(hs, b, 1, [0] * 3, [0] * 3, [0], [0], [0], [1])
        assert b.oracle().check() != 0                    # the oracle's checker agrees that it is malformed
    bad(lambda b: b.local(3), "column out of range")
    bad(lambda b: b.public(1), "public input")
    bad(lambda b: b.ops.append((A.ADD, 0, 0, 0)), "earlier value")
    bad(lambda b: (b.local(0), b.ops.append((A.ADD, 0, 5, 0))), "earlier value")
    bad(lambda b: (b.assert_zero(b.local(0)), b.ops.append((A.ADD, 1, 1, 0))), "earlier value")      # a sink defines no value
    bad(lambda b: b.inv(b.local(0)), "map programs only")
    bad(lambda b: b.store(0, b.local(0)), "map programs only")
    bad(lambda b: b.assert_zero(b.local(0)), "no constraints", kind=A.MAP, n_out=1)
    bad(lambda b: b.store(1, b.local(0)), "output column out of range", kind=A.MAP, n_out=1)
    bad(lambda b: (b.store(0, b.local(0)), b.store(0, b.local(1))), "stored twice", kind=A.MAP, n_out=1)
    bad(lambda b: b.ops.append((99, 0, 0, 0)), "unknown op")
    bad(lambda b: b.ops.append((A.LOCAL, 0, 0, 7)), "reserved")
    bad(lambda b: (b.consts.append(P), b.ops.append((A.CONST, 0, 0, 0))), "canonical")


def test_constraint_degree(hs):
    def deg(build):
        b = A.Builder(A.CONSTRAINTS, 2)
        build(b)
        return hs_point(hs, b, 1, [1, 2], [3, 4], alphas=[1])[2]["max_degree"]
    assert deg(lambda b: b.assert_zero(b.mul(b.mul(b.local(0), b.local(1)), b.next(0)), "transition")) == 3
    assert deg(lambda b: b.assert_zero(b.mul(b.local(0), b.local(1)), "first")) == 3     # a Lagrange factor is one more unit of n
    assert deg(lambda b: b.assert_zero(b.add(b.mul(b.local(0), b.const(5)), b.next(1)))) == 1
    assert deg(lambda b: b.assert_zero(b.mul(b.mul(b.local(0), b.local(0)), b.mul(b.local(0), b.local(0))))) == 4


# ---- the oracle's whole prover / verifier on the toy AIR with a lookup ------------------------------------------------------
def lookup_desc_oracle(db, rb=1, ch=2, pow_bits=5, nq=12, arity=(2,)):
    c, ma, mb = A.lookup_programs()
    oc, oa, ob = c.oracle(), ma.oracle(), mb.oracle()
    fri = O.fri_params(db, rb, ch, pow_bits, nq, arity)
    return O.stark_desc(db, 1, 2, fri, A.LOOKUP_K0, oc, A.LOOKUP_K1, 3, steps=A.lookup_steps(oa, ob))


def split_stark_proof(proof, ch, kt, kq, n_tr):
    """(caps, local, next, quotient, fri bytes) of the layout in include/cityprover.h cp_stark_prove"""
    o = 0

    def u64():
        nonlocal o
        v = struct.unpack_from("<Q", proof, o)[0]
        o += 8
        return v

    def words(n):
        nonlocal o
        a = np.frombuffer(proof, dtype="<u8", count=n, offset=o).copy()
        o += 8 * n
        return a
    assert u64() == n_tr
    caps = []
    for _ in range(n_tr + 1):
        assert u64() == 1 << ch
        caps.append(words(4 << ch))
    assert u64() == kt
    loc = words(2 * kt).reshape(kt, 2)
    assert u64() == kt
    nxt = words(2 * kt).reshape(kt, 2)
    assert u64() == kq
    qz = words(2 * kq).reshape(kq, 2)
    return caps, loc, nxt, qz, proof[o:]


def python_check_at_zeta(db, proof, ch=2):
    """the quotient identity at zeta re-derived in Python: transcript through the oracle's challenger, the constraints through the
    F_p^2 interpretation of air_programs.lookup_constraints (NOT through any recorded program)"""
    n, kt, kq = 1 << db, A.LOOKUP_K0 + A.LOOKUP_K1, 4
    caps, loc, nxt, qz, _ = split_stark_proof(proof, ch, kt, kq, 2)
    c = O.challenger_new()
    O.challenger_observe(c, caps[0])
    beta = [(int(v), 0) for v in O.challenger_challenges(c, 3)]
    O.challenger_observe(c, caps[1])
    alphas = [int(v) for v in O.challenger_challenges(c, 2)]
    O.challenger_observe(c, caps[2])
    zeta = tuple(int(v) for v in O.challenger_challenges(c, 2))
    E = A.ExtField
    import fri_instances as FI
    g = FI.root_of_unity(db)
    g_last = pow(g, n - 1, P)
    zn = FI.e_pow(zeta, n)
    zh = E.sub(zn, E.one)
    z_last = E.sub(zeta, (g_last, 0))
    n_inv = pow(n, P - 2, P)
    l_first = E.mul(FI.e_scale(zh, n_inv), FI.e_inv(E.sub(zeta, E.one)))
    l_last = E.mul(FI.e_scale(zh, n_inv * g_last % P), FI.e_inv(z_last))
    cons = []
    A.lookup_constraints(E, [tuple(int(x) for x in r) for r in loc], [tuple(int(x) for x in r) for r in nxt], beta,
                         lambda v, when: cons.append(E.mul(v, {"all": E.one, "transition": z_last, "first": l_first, "last": l_last}[when])))
    for a, al in enumerate(alphas):
        acc = E.zero
        for v in cons:
            acc = E.add(FI.e_scale(acc, al), v)
        t = E.zero
        for k in reversed(range(2)):
            t = E.add(E.mul(t, zn), tuple(int(x) for x in qz[2 * a + k]))
        if E.mul(t, zh) != acc:
            return False
    return True


@pytest.mark.parametrize("db", [4, 6])
def test_oracle_stark_with_a_lookup_end_to_end(db):
    d, keep = lookup_desc_oracle(db)
    trace = A.lookup_trace(1 << db)
    c = O.challenger_new()
    O.challenger_observe(c, [7, 8, 9])                       # whatever the caller's protocol observes first
    proof = O.stark_prove(d, trace, c)
    after = O.challenger_tuple(c)
    v = O.challenger_new()
    O.challenger_observe(v, [7, 8, 9])
    assert O.stark_verify(d, v, proof) == 0
    assert O.challenger_tuple(v) == after                    # the verifier's transcript ends where the prover's did
    caps, loc, nxt, qz, fri = split_stark_proof(proof, 2, A.LOOKUP_K0 + A.LOOKUP_K1, 4, 2)
    assert len(caps) == 3 and len(fri) > 0                   # the documented layout parses to the last byte of the openings
    # a different transcript prefix, a flipped opening, a flipped cap: refused
    w = O.challenger_new()
    assert O.stark_verify(d, w, proof) != 0
    for off in (8 + 8 + 5, len(proof) // 3, len(proof) - 9):
        bad = bytearray(proof)
        bad[off] ^= 1
        w = O.challenger_new()
        O.challenger_observe(w, [7, 8, 9])
        assert O.stark_verify(d, w, bytes(bad)) != 0, off
    # a value that is not in the table / a broken transition: the prover still produces bytes, the verifier refuses them at zeta
    for cheat in ("value", "fib"):
        c2 = O.challenger_new()
        O.challenger_observe(c2, [7, 8, 9])
        p2 = O.stark_prove(d, A.lookup_trace(1 << db, cheat=cheat), c2)
        w = O.challenger_new()
        O.challenger_observe(w, [7, 8, 9])
        assert O.stark_verify(d, w, p2) == 2, cheat


def test_quotient_identity_at_zeta_in_python():
    db = 5
    d, keep = lookup_desc_oracle(db)
    c = O.challenger_new()
    proof = O.stark_prove(d, A.lookup_trace(1 << db), c)
    assert python_check_at_zeta(db, proof)
    c = O.challenger_new()
    proof = O.stark_prove(d, A.lookup_trace(1 << db, cheat="value"), c)
    assert not python_check_at_zeta(db, proof)
