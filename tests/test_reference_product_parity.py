"""PRODUCT code held against reference-held data (VERDICT r1 "what's weak" #1-#3): the verifier cp_verify runs, the HIP
FRI kernels and the HIP Merkle path, driven with the ten reference proofs of qbench_data/example.bin and the challenges
recovered from them (tests/reference_challenges.py); the sighash whitelist roots (P5); and BASELINE configs[1]'s full-size
Merkle cap as a test. Expected values come from the reference's own bytes, not from the oracle."""
import json
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
import reference_challenges as R
from proof_format import parse_proof, serialize_proof

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "city-rollup_amd"))
P = O.P


def ref_shape(cp, n_pi):
    # standard_recursion_config at degree 2^12: read off the reference proofs
    return cp.standard_recursion_shape(num_public_inputs=n_pi)


def whitelist_trees(golden_dir):
    out = []
    for fx in json.load(open(os.path.join(golden_dir, "sighash_whitelist_tree.json"))):
        leaves = np.zeros((1 << fx["height"], 4), np.uint64)
        leaves[:len(fx["leaves"])] = np.array(fx["leaves"], dtype=np.uint64)
        out.append((leaves, fx["root"]))
    assert [len(json.load(open(os.path.join(golden_dir, "sighash_whitelist_tree.json")))[i]["leaves"]) for i in range(2)] == [1875, 162]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# cp_verify's query phase on the reference proofs (host code of the product: runs without a GPU)
@pytest.mark.parametrize("which", range(10))
def test_product_verifier_query_phase_accepts_reference_proofs(golden_dir, which):
    import cityprover as cp
    blob, pf, ch = R.challenges(golden_dir, which)
    sh = ref_shape(cp, len(pf["public_inputs"]))
    args = (ch["alpha"], ch["zeta"], ch["betas"], ch["x_indices"])
    cp.verify_fri_queries_with_challenges(sh, blob, *args)            # 28 queries x (3 oracle paths + combine + 2 folds + final)
    if which > 1:
        return
    # and it is a real check: any perturbation of the reference bytes or of a challenge is refused, naming the failing step
    def refused(mutate, match, a=args):
        d = parse_proof(blob)
        mutate(d)
        with pytest.raises(cp.CityProverError, match=match):
            cp.verify_fri_queries_with_challenges(sh, serialize_proof(d), *a)
    refused(lambda d: d["queries"][3]["initial"][1][0].__setitem__(7, (d["queries"][3]["initial"][1][0][7] + 1) % P), "Merkle path of oracle 1")
    refused(lambda d: d["queries"][0]["initial"][3][1][2].__setitem__(0, 5), "Merkle path of oracle 3")
    refused(lambda d: d["openings"]["wires"][9].__setitem__(0, (d["openings"]["wires"][9][0] + 1) % P), "FRI layer 0 value")
    refused(lambda d: d["queries"][5]["steps"][1][0][3].__setitem__(1, 11), "FRI layer 1|Merkle path of FRI layer 1")
    refused(lambda d: d["final_poly"][2].__setitem__(0, (d["final_poly"][2][0] + 1) % P), "final polynomial")
    refused(lambda d: d["commit_caps"][1][4].__setitem__(1, 9), "Merkle path of FRI layer 1")
    bump = lambda e: ((e[0] + 1) % P, e[1])
    for k in (0, 1):                                                  # alpha / zeta off by one
        a = list(args)
        a[k] = bump(a[k])
        refused(lambda d: None, "FRI layer 0 value", tuple(a))
    a = list(args)
    a[2] = [bump(args[2][0]), args[2][1]]
    refused(lambda d: None, "FRI layer 1 value", tuple(a))
    a = list(args)
    a[2] = [args[2][0], bump(args[2][1])]
    refused(lambda d: None, "final polynomial", tuple(a))
    a = list(args)
    a[3] = [args[3][0] ^ 1] + list(args[3][1:])
    refused(lambda d: None, "Merkle path", tuple(a))


def test_whitelist_roots_oracle(golden_dir):
    """P5 on the oracle: the height-16 tree over the sorted fingerprints (empty leaves = zero) has the reference's root, for
    the live configuration (1875 circuits) and the earlier one the file keeps (162)."""
    for leaves, root in whitelist_trees(golden_dir):
        assert [int(x) for x in O.merkle_tree(leaves, 0)[0]] == root


# ---------------------------------------------------------------------------------------------------------------------
pytest_gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    yield p
    p.close()


@pytest_gpu
def test_whitelist_roots_on_gpu(prover, golden_dir):
    """P5 on the GPU: MerkleTree::new over 2^16 four-element leaves (hash_or_noop: a leaf of <= 4 elements is its own
    digest) with cap height 0 == SIGHASH_WHITELIST_TREE_ROOT (sighash_wrapper_config.rs:16-23)."""
    for leaves, root in whitelist_trees(golden_dir):
        assert [int(x) for x in prover.merkle_cap(leaves, 0)[0]] == root
        assert [int(x) for x in prover.merkle_cols(np.ascontiguousarray(leaves.T), 0)[0]] == root


@pytest_gpu
def test_reference_merkle_paths_on_gpu_all_proofs(prover, golden_dir):
    """Every Merkle path of every reference proof — 10 proofs x 28 queries x (wires, Z, quotient oracle + 2 FRI layers) =
    1400 paths — recomputed with the GPU's Poseidon (cp_hash_no_pad, cp_two_to_one) up to the caps the proofs carry."""
    leaves = {}   # leaf length -> list of (leaf values, index, siblings, cap)
    for which in range(10):
        _, pf, ch = R.challenges(golden_dir, which)
        caps = [None, pf["wires_cap"], pf["zs_pp_cap"], pf["quotient_cap"]]
        for x, q in zip(ch["x_indices"], pf["queries"]):
            for b in (1, 2, 3):
                vals, sib = q["initial"][b]
                leaves.setdefault(len(vals), []).append((vals, x, sib, caps[b]))
            xi = x
            for l, (evals, sib) in enumerate(q["steps"]):
                flat = [c for e in evals for c in e]
                leaves.setdefault(len(flat), []).append((flat, xi >> 4, sib, pf["commit_caps"][l]))
                xi >>= 4
    total = 0
    for ln, items in leaves.items():
        cur = prover.hash_no_pad(np.array([it[0] for it in items], dtype=np.uint64))
        idx = np.array([it[1] for it in items], dtype=np.int64)
        by_depth = {}
        for k, it in enumerate(items):
            by_depth.setdefault(len(it[2]), []).append(k)
        for depth, ks in by_depth.items():
            ks = np.array(ks)
            c, ix = cur[ks], idx[ks].copy()
            for lvl in range(depth):
                sib = np.array([items[k][2][lvl] for k in ks], dtype=np.uint64)
                right = (ix & 1).astype(bool)[:, None]
                c = prover.two_to_one(np.where(right, sib, c), np.where(right, c, sib))
                ix >>= 1
            for j, k in enumerate(ks):
                assert [int(v) for v in c[j]] == list(items[k][3][int(ix[j])]), (ln, depth, k)
            total += len(ks)
    assert total == 10 * 28 * 5


@pytest_gpu
@pytest.mark.parametrize("which", range(10))
def test_hip_fri_combine_on_reference_openings(prover, golden_dir, which):
    """The HIP batch-combine kernel (fri::k_combine, cp_fri_combine_dev) on the reference's data: 'polynomial' j holds, at
    position q, the j-th opened leaf value of query q (256 values in the batch order constants, sigmas, wires, Z,
    partial products, quotient chunks), so that its output at q is the alpha-reduction the verifier's
    fri_combine_initial needs; with the openings reduced the same way and the recovered alpha / zeta, the result must be
    the value the reference opened in the first FRI layer — for all 28 queries."""
    import cityprover as cp
    from reference_challenges import LOG_DEG, LOG_N, base, eadd, einv, emul, esub, rev
    _, pf, ch = R.challenges(golden_dir, which)
    alpha, zeta = ch["alpha"], ch["zeta"]
    nq = len(pf["queries"])
    vals = np.array([[v for e in q["initial"] for v in e[0]] for q in pf["queries"]], dtype=np.uint64)     # (28, 256)
    o = pf["openings"]
    open0 = [e for k in ("constants", "plonk_sigmas", "wires", "plonk_zs", "partial_products", "quotient_polys") for e in o[k]]
    assert vals.shape == (nq, 256) and len(open0) == 256
    # columns 0..27: the queries' leaf values; 28, 29: real / imaginary parts of the openings (reduced by linearity)
    polys = np.zeros((256, 32), np.uint64)
    polys[:, :nq] = vals.T
    polys[:, nq] = [e[0] for e in open0]
    polys[:, nq + 1] = [e[1] for e in open0]
    red = cp.fri_combine(prover, polys, alpha)                       # (32, 2): sum_j alpha^j column[j]
    zs = np.zeros((2, 32), np.uint64)
    zs[:, :nq] = np.array([q["initial"][2][0][:2] for q in pf["queries"]], dtype=np.uint64).T
    zs[:, nq] = [e[0] for e in o["plonk_zs_next"]]
    zs[:, nq + 1] = [e[1] for e in o["plonk_zs_next"]]
    red1 = cp.fri_combine(prover, zs, alpha)
    W7 = (0, 1)                                                      # the extension generator: re + im * X

    def opened(r):   # reduction of extension-valued openings from their real / imaginary columns
        return eadd(tuple(int(v) for v in r[nq]), emul(W7, tuple(int(v) for v in r[nq + 1])))
    O0, O1 = opened(red), opened(red1)
    omega = pow(7, (P - 1) >> LOG_N, P)
    g = pow(7, (P - 1) >> LOG_DEG, P)
    a2 = emul(alpha, alpha)
    for qi, (x_index, q) in enumerate(zip(ch["x_indices"], pf["queries"])):
        x = 7 * pow(omega, rev(x_index, LOG_N), P) % P
        t0 = emul(esub(tuple(int(v) for v in red[qi]), O0), einv(esub(base(x), zeta)))
        t1 = emul(esub(tuple(int(v) for v in red1[qi]), O1), einv(esub(base(x), emul(zeta, base(g)))))
        assert eadd(emul(t0, a2), t1) == tuple(q["steps"][0][0][x_index & 15]), qi


@pytest_gpu
@pytest.mark.parametrize("which", range(10))
def test_hip_fri_fold_on_reference_layers(prover, golden_dir, which):
    """The HIP fold kernel (fri::k_fold, cp_fri_fold_dev) on the reference's layer evaluations: the 16 values a query opens
    in a layer are the restriction of the layer polynomial to one coset; their interpolant's coefficients c_0..c_15 are
    f_j(y) of the decomposition f(X) = sum_j X^j f_j(X^16), which is exactly what the kernel folds: sum_j beta^j c_j. With
    the recovered betas the result must be the value the reference opened in the NEXT layer (layer 0 -> 1) and the
    final polynomial evaluated at the folded point (layer 1 -> final), for all 28 queries."""
    import cityprover as cp
    from reference_challenges import LOG_N, fold_poly, peval, rev
    _, pf, ch = R.challenges(golden_dir, which)
    omega = pow(7, (P - 1) >> LOG_N, P)
    nq = len(pf["queries"])
    for layer in (0, 1):
        coeffs = np.zeros((2, 16 * nq), np.uint64)
        want = []
        for qi, (x_index, q) in enumerate(zip(ch["x_indices"], pf["queries"])):
            x = 7 * pow(omega, rev(x_index, LOG_N), P) % P
            xl = pow(x, 16 ** layer, P)
            c = fold_poly(xl, (x_index >> (4 * layer)) & 15, 4, q["steps"][layer][0])     # interpolation: test-side, Python ints
            coeffs[0, 16 * qi:16 * qi + 16] = [e[0] for e in c]
            coeffs[1, 16 * qi:16 * qi + 16] = [e[1] for e in c]
            if layer == 0:
                want.append(tuple(q["steps"][1][0][(x_index >> 4) & 15]))
            else:
                want.append(peval([tuple(e) for e in pf["final_poly"]], (pow(xl, 16, P), 0)))
        got = cp.fri_fold(prover, coeffs, 4, ch["betas"][layer])
        assert [(int(got[0, i]), int(got[1, i])) for i in range(nq)] == want, layer


@pytest_gpu
def test_full_size_merkle_cap_matches_oracle(prover):
    """BASELINE.json configs[1] at full size: the Poseidon Merkle cap (height 4) over 2^20 rows x 135 columns, GPU vs the
    CPU oracle on the same seeded input (the check bench.py makes before timing, as a test)."""
    n, k = 1 << 20, 135
    cols = O.splitmix64_felts(0x243F6A8885A308D3 + 1, k * n).reshape(k, n)
    O.lib().or_set_threads(min(os.cpu_count() or 1, 64))
    want = np.zeros((16, 4), np.uint64)
    O.lib().or_merkle_tree_cols(O.ptr(cols), n, k, n, 4, None, O.ptr(want))
    O.lib().or_set_threads(1)
    got = prover.merkle_cols(cols, 4)
    assert (got == want).all()
