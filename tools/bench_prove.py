#!/usr/bin/env python3
"""proofs/s of the whole GPU prover (cp_prove_batch: wires -> ProofWithPublicInputs) on synthetic
qbench-shaped jobs: standard_recursion_config, n = 2^12, 135 wires / 80 routed, 28 queries, 16-bit PoW
(SURVEY.md §8(d) M1). Circuits carry the whole city-common gate set (pad_circuit.rs:31-55: 14 gate types in 4 selector
groups, 6 constants columns) with the recursion-circuit row mix: Poseidon ~60 %, Arithmetic / ArithmeticExtension /
MulExtension ~25 %, Reducing / ReducingExtension / RandomAccess / BaseSum / CosetInterpolation ~10 %, Noop padding.
One block of the example workload = 64 plonky2 proofs."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import cityprover as cp  # noqa: E402


class ProductBackend:
    """Witness-side helpers for the synthetic circuits WITHOUT the oracle: public-input hash through the
    product ABI, PoseidonGate rows through tools/witgen (product headers compiled for the host)."""

    def __init__(self, prover):
        self.prover = prover

    def hash_no_pad(self, xs):
        return [int(v) for v in self.prover.hash_no_pad(np.array(xs, dtype=np.uint64))]

    def poseidon_rows(self, inputs, swaps):
        import witgen
        return witgen.poseidon_gate_rows(inputs, swaps)


_CASES = {}


def recursion_mix(poseidon_fraction):
    import synth_gates as SG
    rest = 1.0 - poseidon_fraction
    return {SG.POSEIDON: poseidon_fraction, SG.ARITHMETIC: 0.25 * rest, SG.ARITHMETIC_EXT: 0.25 * rest, SG.MUL_EXT: 0.125 * rest,
            SG.REDUCING: 0.05 * rest, SG.REDUCING_EXT: 0.05 * rest, SG.RANDOM_ACCESS: 0.075 * rest, SG.BASE_SUM: 0.05 * rest,
            SG.COSET_INTERPOLATION: 0.025 * rest, SG.POSEIDON_MDS: 0.025 * rest, SG.COMPARISON: 0.025 * rest}


def cases_for(prover, n_circuits, poseidon_fraction, gate_set="city_common"):
    key = (n_circuits, poseidon_fraction, gate_set)
    if key not in _CASES:
        be = ProductBackend(prover)
        kw = dict(db=12, num_routed=80, num_wires=135, chunk=8, rate_bits=3, arity_bits=(4, 4), cap_height=4, pow_bits=16,
                  num_query_rounds=28, n_copies=64, backend=be)
        if gate_set == "basic":   # Noop / Constant / PublicInput / Arithmetic (+ Poseidon): the first-milestone circuits
            from synth_circuit import build
            _CASES[key] = [build(seed=i, poseidon_fraction=poseidon_fraction, **kw) for i in range(n_circuits)]
        else:
            import synth_gates as SG
            _CASES[key] = [SG.build_gate_set(SG.CITY_COMMON if gate_set == "city_common" else SG.ALL_GATES, seed=i,
                                             weights=recursion_mix(poseidon_fraction), noop_fraction=0.03, **kw)
                           for i in range(n_circuits)]
    return _CASES[key]


POSEIDON_FRACTION = 0.6  # recursion-circuit gate mix (Poseidon ~60 % of the rows)


def run(prover, B, iters, n_circuits=4, profile=False, poseidon_fraction=POSEIDON_FRACTION, gate_set="city_common",
        host_wires=False):
    cases = cases_for(prover, n_circuits, poseidon_fraction, gate_set)
    sh = cp.standard_recursion_shape(num_constants=cases[0]["num_constants"], num_public_inputs=len(cases[0]["public_inputs"]))  # selectors + 2 gate constants
    circs = []
    for i, c in enumerate(cases):
        circ = cp.Circuit(prover, sh, [i, 1, 2, 3], c["cs_values"])
        cp.set_gates(circ, c["gate_list"], c["num_selectors"])
        circs.append(circ)
    pick = [i % n_circuits for i in range(B)]
    dw = prover.to_device(np.stack([cases[i]["wires"] for i in pick]))
    pis = [cases[i]["public_inputs"] for i in pick]
    cs = [circs[i] for i in pick]
    proofs = cp.prove_batch_dev(prover, cs, pis, dw.ptr)  # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        proofs = cp.prove_batch_dev(prover, cs, pis, dw.ptr)
    t1 = time.perf_counter()
    out = {"B": B, "ms_per_batch": (t1 - t0) * 1e3 / iters, "proofs_per_s": B * iters / (t1 - t0),
           "blocks_per_s": B * iters / (t1 - t0) / 64.0, "proof_bytes": len(proofs[0]), "gate_set": gate_set,
           "n_gate_types": len(cases[0]["gate_list"]), "num_selectors": cases[0]["num_selectors"]}
    if host_wires:   # PCIe-inclusive: wires start in host memory (what the Rust shim hands over), proofs end there
        pin = [prover.pinned(c["wires"]) for c in cases]   # page-locked, as the shim would allocate them (cp_host_alloc)
        hw = [pin[i] for i in pick]
        cp.prove_batch(prover, cs, pis, hw)
        t0 = time.perf_counter()
        for _ in range(iters):
            cp.prove_batch(prover, cs, pis, hw)
        t1 = time.perf_counter()
        out["proofs_per_s_host_wires"] = B * iters / (t1 - t0)
        out["h2d_bytes_per_proof"] = int(hw[0].nbytes)
        pg = [cases[i]["wires"] for i in pick]
        t0 = time.perf_counter()
        for _ in range(iters):
            cp.prove_batch(prover, cs, pis, pg)
        t1 = time.perf_counter()
        out["proofs_per_s_pageable_host_wires"] = B * iters / (t1 - t0)
        prover.free_pinned()
    if profile:
        prover.profile_begin()
        cp.prove_batch_dev(prover, cs, pis, dw.ptr)
        prof = prover.profile_end()
        kern = {k: v for k, v in prof.items() if not k.startswith(("host:", "wait:"))}
        out["kernel_ms_per_batch"] = sum(v["total_ms"] for v in kern.values())
        out["kernels_ms"] = {k: round(v["total_ms"], 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["total_ms"])}
        # host phases of the call: wall ms, and the part NOT spent blocked on the stream (= host work the GPU waits for
        # unless another context fills the gap)
        out["host_phases_ms"] = {k[5:]: [round(v["total_ms"], 3), round(v["total_ms"] - prof["wait:" + k[5:]]["total_ms"], 3)]
                                 for k, v in prof.items() if k.startswith("host:")}
    dw.free()
    for c in circs:
        c.close()
    return out


def run_threads(T, B, iters, device=0, gate_set="city_common", host_wires=False):
    """T host threads, each with its own context (stream): host transcript work of one batch overlaps the
    kernels of the others. Wall clock from a common start (after every thread has loaded its circuits and run a
    warm-up batch) until the last thread finishes its `iters` batches."""
    import threading
    provers = [cp.Prover(device) for _ in range(T)]
    cases = cases_for(provers[0], 4, POSEIDON_FRACTION, gate_set)
    sh = cp.standard_recursion_shape(num_constants=cases[0]["num_constants"], num_public_inputs=len(cases[0]["public_inputs"]))
    start = threading.Barrier(T + 1)
    done = [None] * T
    pick = [i % len(cases) for i in range(B)]
    pis = [cases[i]["public_inputs"] for i in pick]

    def work(t):
        p = provers[t]
        circs = []
        for i, c in enumerate(cases):
            circ = cp.Circuit(p, sh, [i, 1, 2, 3], c["cs_values"])
            cp.set_gates(circ, c["gate_list"], c["num_selectors"])
            circs.append(circ)
        cs = [circs[i] for i in pick]
        if host_wires:
            pin = [p.pinned(c["wires"]) for c in cases]
            hw = [pin[i] for i in pick]
            go = lambda: cp.prove_batch(p, cs, pis, hw)
        else:
            dw = p.to_device(np.stack([cases[i]["wires"] for i in pick]))
            go = lambda: cp.prove_batch_dev(p, cs, pis, dw.ptr)
        go()
        start.wait()
        for _ in range(iters):
            go()
        done[t] = time.perf_counter()
        if not host_wires:
            dw.free()
        for c in circs:
            c.close()

    ths = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for t in ths:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in ths:
        t.join()
    for pr in provers:
        pr.close()
    wall = max(done) - t0
    return {"threads": T, "B": B, "iters_per_thread": iters, "wall_s": wall, "host_wires": host_wires,
            "proofs_per_s_steady": T * B * iters / wall, "blocks_per_s_steady": T * B * iters / wall / 64.0}


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    batches = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 16, 64]
    threads = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else []
    gate_set = sys.argv[4] if len(sys.argv) > 4 else "city_common"
    p = cp.Prover(0)
    t0 = time.perf_counter()
    cases_for(p, 4, POSEIDON_FRACTION, gate_set)
    print("witness generation for 4 circuits: %.1f s" % (time.perf_counter() - t0), file=sys.stderr)
    threads_only = len(sys.argv) > 5 and sys.argv[5] == "threads_only"
    out = [] if threads_only else [run(p, B, iters, profile=True, gate_set=gate_set, host_wires=True) for B in batches]
    p.close()
    for T in threads:
        out.append(run_threads(T, 32, max(iters, 8), gate_set=gate_set))
        if not threads_only:
            out.append(run_threads(T, 32, max(iters, 8), gate_set=gate_set, host_wires=True))
    print(json.dumps(out))
