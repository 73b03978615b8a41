"""GPU parity of the two plonky2-level seams (include/cityprover.h cp_batch_* / cp_fri_prove): `PolynomialBatch` handles and
`PolynomialBatch::prove_openings` over ARBITRARY oracles and opening batches, against the CPU oracle (or_batch_* / or_fri_prove):
  * commitments (values / coefficients, with and without blinding): cap, openings, `get_lde_values` rows;
  * seeded random FRI instances (1-5 oracles, 1-4 batches given as up to 39 polynomial runs, 0-4 reduction layers): FriProof
    bytes and the challenger state handed back, both verifiers accept;
  * the SHA-256 STARK's shapes (city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:55-79,310-312: 418 free + 912
    extended columns, 2^k rows; starky's fast config: rate_bits 1, cap height 4, 16-bit PoW, 84 queries) with three opening batches,
    bytes == oracle at 2^14 and 2^16 rows;
  * a toy AIR (cubic-extension-free) proved end to end through the C ABI and verified.
Parity for A13 itself stays UNPINNED: the reference asserts SHA digests only (smartgadget.rs:505-513), no STARK proof bytes exist."""
import os

import numpy as np
import pytest

import oracle_lib as O
import fri_instances as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    O.lib().or_set_threads(min(16, os.cpu_count() or 1))
    yield p
    O.lib().or_set_threads(1)
    p.close()


@pytest.mark.parametrize("k,db,rb,ch,coeffs,salt", [(5, 6, 1, 2, False, False), (135, 12, 3, 4, False, False), (7, 5, 2, 0, True, False),
                                                    (20, 8, 3, 4, False, True), (3, 3, 1, 4, True, True), (1, 0, 1, 0, False, False),
                                                    (33, 13, 1, 4, False, False)])
def test_batch_commit_matches_oracle(prover, k, db, rb, ch, coeffs, salt):
    import cityprover
    n, N = 1 << db, 1 << (db + rb)
    polys = O.splitmix64_felts(k * 31 + db, k * n).reshape(k, n)
    salts = O.splitmix64_felts(77 + k, 4 * N).reshape(4, N) if salt else None
    g = cityprover.PolyBatch(prover, polys, rb, ch, coeffs, salts)
    o = O.Batch(polys, rb, ch, coeffs, salts)
    try:
        assert (g.cap() == o.cap()).all()
        pt = np.array([0x1234567890ABCDEF % O.P, 0x0FEDCBA987654321], dtype=np.uint64)
        assert (g.eval_ext(pt) == o.eval_ext(pt)).all()
        if k > 2:
            assert (g.eval_ext(pt, 1, k - 2) == o.eval_ext(pt, 1, k - 2)).all()
        cnt = min(N, 37)
        for first, step in ((0, 1), (N - cnt, 1), (3 % N, 1 << rb), (1, 3)):
            if (first + cnt > N and step == 1) or step > N:
                continue
            assert (g.lde_rows(first, cnt, step) == o.lde_rows(first, cnt, step)).all()
        # merkle_tree.leaves: leaf order, salt included
        lv = o.lde()
        assert (g.leaves(0, min(N, 50)) == lv[:, :min(N, 50)].T).all()
        assert (g.leaves(N - 1, 1) == lv[:, N - 1:].T).all()
        assert (g.coeffs() == o.coeffs()).all() and (g.coeffs(k - 1, 1) == o.coeffs()[k - 1:]).all()
        # device views: the coefficient array and the bit-reversed LDE the handle keeps
        cptr, lptr = g.device_ptrs()
        co = np.empty(k * n, np.uint64)
        prover._check(prover.lib.cp_d2h(prover.ctx, co.ctypes.data, cptr, co.size * 8))
        assert (co.reshape(k, n) == o.coeffs()).all()
        ld = np.empty(k * N, np.uint64)
        prover._check(prover.lib.cp_d2h(prover.ctx, ld.ctypes.data, lptr, ld.size * 8))
        assert (ld.reshape(k, N) == o.lde()[:k]).all()
    finally:
        g.close()
        o.close()


def test_batch_commit_refuses_bad_arguments(prover):
    import cityprover
    ok = O.splitmix64_felts(1, 4 * 16).reshape(4, 16)
    bad = ok.copy()
    bad[2, 3] = O.P
    with pytest.raises(cityprover.CityProverError, match="canonical"):
        cityprover.PolyBatch(prover, bad, 1, 2)
    with pytest.raises(cityprover.CityProverError, match="cap_height"):
        cityprover.PolyBatch(prover, ok, 1, 9)
    # the scan for elements >= p runs on the device behind the upload: the message still names the element, the last element of the
    # last polynomial is seen, a bad salt is seen, and the flag does not stick to the next call
    for r, c in ((0, 0), (3, 15)):
        bad = ok.copy()
        bad[r, c] = 2**64 - 1
        with pytest.raises(cityprover.CityProverError, match="polynomial element %d is not canonical" % (16 * r + c)):
            cityprover.PolyBatch(prover, bad, 1, 2)
    salts = O.splitmix64_felts(2, 4 * 32).reshape(4, 32)
    bs = salts.copy()
    bs[3, 31] = O.P
    with pytest.raises(cityprover.CityProverError, match="salt element 127 is not canonical"):
        cityprover.PolyBatch(prover, ok, 1, 2, salts=bs)
    cityprover.PolyBatch(prover, ok, 1, 2, salts=salts).close()
    b = cityprover.PolyBatch(prover, ok, 1, 2)
    with pytest.raises(cityprover.CityProverError, match="out of range"):
        b.eval_ext(np.array([3, 0], dtype=np.uint64), 2, 3)
    with pytest.raises(cityprover.CityProverError, match="canonical"):
        b.eval_ext(np.array([O.P, 0], dtype=np.uint64))
    # FriParams that do not match the batch, a batch list pointing outside an oracle, a point on the LDE coset
    st = cityprover.ChallengerState()
    with pytest.raises(cityprover.CityProverError, match="differ"):
        cityprover.fri_prove(prover, [b], [((3, 0), [(0, 0, 4)])], cityprover.fri_params(4, 2, 2, 0, 2, ()), st)
    with pytest.raises(cityprover.CityProverError, match="out of range"):
        cityprover.fri_prove(prover, [b], [((3, 0), [(0, 2, 3)])], cityprover.fri_params(4, 1, 2, 0, 2, ()), st)
    with pytest.raises(cityprover.CityProverError, match="LDE coset"):
        cityprover.fri_prove(prover, [b], [((7, 0), [(0, 0, 4)])], cityprover.fri_params(4, 1, 2, 0, 2, ()), st)
    assert st.as_tuple() == cityprover.ChallengerState().as_tuple()   # a refused call leaves the transcript alone
    b.close()


def test_batch_pool_two_contexts_exported_handles_oom_flush_and_orphans(prover):
    """the buffer pool behind cp_batch_destroy (csrc/dev_pool.h; its logic alone: tests/test_hostsim.py): commitments made and
    destroyed alternately from two contexts of one device keep giving the oracle's caps while they recycle each other's
    buffers; a handle whose device pointers were handed out is not recycled; an out-of-memory from the runtime
    (CP_FAULT_DEVMEM) empties the pool and the call succeeds, with nothing parked it is CP_ERR_OOM and the context stays
    usable; a handle may outlive its context."""
    import cityprover
    lib = cityprover.load_library()
    p2 = cityprover.Prover(0)
    try:
        cityprover.batch_pool_trim(prover)
        s0 = cityprover.batch_pool_stats(0)
        assert s0["pooled"] == 1 and s0["bytes"] == 0
        assert cityprover.batch_pool_stats(4096)["pooled"] == 0   # no pool for a device index outside the table
        shapes = [(9, 7, 1, 3), (5, 6, 2, 2)]
        want = {}
        for k, db, rb, ch in shapes:
            polys = O.splitmix64_felts(1000 + k, k << db).reshape(k, 1 << db)
            o = O.Batch(polys, rb, ch)
            want[(k, db, rb, ch)] = (polys, o.cap().copy())
            o.close()
        for it in range(8):
            ctx = prover if it % 2 == 0 else p2
            k, db, rb, ch = shapes[(it // 2) % 2]
            polys, cap = want[(k, db, rb, ch)]
            g = cityprover.PolyBatch(ctx, polys, rb, ch)
            assert (g.cap() == cap).all(), it
            pt = np.array([12345, 678], dtype=np.uint64)
            assert g.eval_ext(pt).shape == (k, 2)
            g.close()
        s1 = cityprover.batch_pool_stats(0)
        assert s1["hits"] - s0["hits"] >= 4 * 5 and s1["bytes"] > 0 and s1["buffers"] == 8   # 4 buffers per shape parked
        # exported pointers: this handle's buffers leave through hipFree
        k, db, rb, ch = shapes[0]
        g = cityprover.PolyBatch(prover, want[shapes[0]][0], rb, ch)
        g.device_ptrs()
        g.close()
        s2 = cityprover.batch_pool_stats(0)
        assert s2["buffers"] == 4 and s2["bytes"] < s1["bytes"]
        # the runtime reports out-of-memory once: the pool is given back, the commitment (a shape the pool does not hold) succeeds
        polys3 = O.splitmix64_felts(5, 3 << 5).reshape(3, 32)
        o3 = O.Batch(polys3, 1, 1)
        lib.cp_fault_inject(3, 0)
        g = cityprover.PolyBatch(p2, polys3, 1, 1)
        assert (g.cap() == o3.cap()).all()
        s3 = cityprover.batch_pool_stats(0)
        assert s3["trims"] == s2["trims"] + 1 and s3["bytes"] == 0
        g.close()
        cityprover.batch_pool_trim(prover)
        # nothing parked: the same fault is an out-of-memory status, and the context carries on
        lib.cp_fault_inject(3, 0)
        with pytest.raises(cityprover.CityProverError, match="(?i)memory"):
            cityprover.PolyBatch(p2, polys3, 1, 1)
        lib.cp_fault_inject(3, -1)
        g = cityprover.PolyBatch(p2, polys3, 1, 1)
        assert (g.cap() == o3.cap()).all()
        o3.close()
        # a handle outlives its context: only destroy is left
        p3 = cityprover.Prover(0)
        h = cityprover.PolyBatch(p3, polys3, 1, 1)
        p3.close()
        with pytest.raises(cityprover.CityProverError, match="destroyed"):
            h.cap()
        h.close()
        g.close()
    finally:
        lib.cp_fault_inject(3, -1)
        p2.close()


def test_twin_on_demand_of_a_cpu_committed_oracle(prover):
    """what rust/plonky2-hwa-patch/cityprover.rs::twin_on_demand does for an oracle that was committed on the CPU (a
    from_values oracle meeting an un-hooked / small from_coeffs one): its coefficients and the salts its host leaves end in,
    committed with CP_BATCH_FROM_COEFFS, reproduce the same cap — so a mixed set of oracles can be proved on the device."""
    import cityprover
    k, db, rb, ch = 6, 7, 2, 3
    n, N = 1 << db, 1 << (db + rb)
    vals = O.splitmix64_felts(4711, k * n).reshape(k, n)
    salts = O.splitmix64_felts(4712, 4 * N).reshape(4, N)
    host = O.Batch(vals, rb, ch, False, salts)                      # the "CPU-committed" oracle
    leaves = host.lde()                                              # (k + 4) x N in leaf order: the host mirror's leaves
    recovered = np.ascontiguousarray(leaves[k:k + 4])                # SALT_SIZE x N, indexed by leaf
    twin = cityprover.PolyBatch(prover, host.coeffs(), rb, ch, True, recovered)
    try:
        assert (twin.cap() == host.cap()).all()
        assert (twin.leaves(0, N) == leaves.T).all()
    finally:
        twin.close()
        host.close()


def test_switches_are_per_context_and_give_the_same_bytes(prover):
    """cp_ctx_set_option (the CITYPROVER_* switches used to be process-wide statics): a second context of the
    same process with other forms selected - leaves hashed twelve lanes per leaf never / always, no fused Merkle level, other
    cooperative-level switches - commits the same caps and proves the same FRI bytes as the first, which keeps its defaults."""
    import cityprover
    p2 = cityprover.Prover(0)
    try:
        with pytest.raises(cityprover.CityProverError, match="unknown option"):
            p2.set_option("NO_SUCH_SWITCH", 1)
        for name, v in (("COOP_LEAF_MAX", 0), ("MERKLE_FUSE", 0), ("COOP_MAX", 64), ("COOP_FRI_MAX", 1 << 20), ("COOP_FUSE", 2), ("DEVICE_TRANSCRIPT", 1)):
            p2.set_option(name, v)
        for seed in (1, 6, 11):
            spec = F.random_instance(seed, degree_bits=10 if seed == 11 else None)
            a = F.run_instance(F.GpuBackend(prover), spec)
            b = F.run_instance(F.GpuBackend(p2), spec)
            assert [c.tolist() for c in a["caps"]] == [c.tolist() for c in b["caps"]]
            assert a["proof"] == b["proof"] and a["state_after"] == b["state_after"]
        p2.set_option("COOP_LEAF_MAX", 1 << 30)          # and the other way round: twelve lanes per leaf always
        p2.set_option("MERKLE_FUSE", 3)
        spec = F.random_instance(4, degree_bits=9)
        assert F.run_instance(F.GpuBackend(prover), spec)["proof"] == F.run_instance(F.GpuBackend(p2), spec)["proof"]
    finally:
        p2.close()


N_RANDOM = int(os.environ.get("CITY_RANDOM_FRI", "40"))


@pytest.mark.parametrize("seed", range(N_RANDOM))
def test_fri_prove_matches_oracle_on_random_instances(prover, seed):
    spec = F.random_instance(seed)
    want = F.run_instance(F.OracleBackend(), spec)
    prover.set_device_transcript(seed % 2)   # the transcript hashed on the device (odd seeds) / on the host (even): same bytes
    try:
        got = F.run_instance(F.GpuBackend(prover), spec)
    finally:
        prover.set_device_transcript(-1)
    assert got["state_before"] == want["state_before"]
    assert got["proof"] == want["proof"], spec
    assert got["state_after"] == want["state_after"]
    F.verify_both(spec, got)


def test_fri_prove_with_injected_pow_witness(prover):
    spec = F.random_instance(3)
    spec["pow_bits"] = 5
    want = F.run_instance(F.OracleBackend(), spec, pow_override=12345)
    got = F.run_instance(F.GpuBackend(prover), spec, pow_override=12345)
    assert got["proof"] == want["proof"] and got["state_after"] == want["state_after"]


def stark_spec(log_rows, seed):
    # trace rounds of the SHA-256 STARK (418 free + 912 extended columns) + a quotient oracle; everything opened at zeta, the
    # trace also at g*zeta (next row), and a third batch at g^2*zeta over a sub-range (what a multi-row window would open)
    arity = []
    bits, cap = log_rows + 1, 4
    while bits - 4 >= max(cap, 1) and log_rows - 4 * (len(arity) + 1) >= 0 and (log_rows - 4 * len(arity)) > 5:
        arity.append(4)   # ConstantArityBits(4, 5): reduce by 16 while the degree is above 2^5 and the layer still has a cap
        bits -= 4
    return dict(seed=seed, degree_bits=log_rows, rate_bits=1, cap_height=cap, arity_bits=tuple(arity), pow_bits=16, num_query_rounds=84,
                ks=[418, 912, 8], blinding=[False, False, False],
                batches=[[(0, 0, 418), (1, 0, 912), (2, 0, 8)], [(0, 0, 418), (1, 0, 912)], [(1, 100, 300), (0, 17, 5)]])


@pytest.mark.parametrize("log_rows", [14, 16])
def test_fri_prove_at_the_sha256_stark_shapes(prover, log_rows):
    spec = stark_spec(log_rows, 4242 + log_rows)
    polys, salts = F.instance_inputs(spec)
    want = F.run_instance(F.OracleBackend(), spec, polys, salts)
    prover.set_device_transcript(1 if log_rows == 16 else 0)
    try:
        got = F.run_instance(F.GpuBackend(prover), spec, polys, salts)
    finally:
        prover.set_device_transcript(-1)
    assert [c.tolist() for c in got["caps"]] == [c.tolist() for c in want["caps"]]
    assert all((a == b).all() for a, b in zip(got["opened"], want["opened"]))
    assert got["proof"] == want["proof"]
    assert got["state_after"] == want["state_after"]
    F.verify_both(spec, got)


def test_toy_air_end_to_end_through_the_c_abi(prover):
    want = F.toy_stark_prove(F.OracleBackend(), degree_bits=8, rate_bits=1, cap_height=3, pow_bits=8, num_query_rounds=20, arity_bits=(3, 2))
    got = F.toy_stark_prove(F.GpuBackend(prover), degree_bits=8, rate_bits=1, cap_height=3, pow_bits=8, num_query_rounds=20, arity_bits=(3, 2))
    assert got["proof"] == want["proof"] and got["state_after"] == want["state_after"]
    assert F.toy_stark_verify(got, use_product=True) is None
    assert F.toy_stark_verify(got, use_product=False) is None
