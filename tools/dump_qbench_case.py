#!/usr/bin/env python3
"""Writes the input of the native harness tools/cityprover_qbench: qbench-shaped synthetic circuits (the same ones
tools/bench_prove.py proves: n = 2^12, 135 wires / 80 routed, city-common gate set, recursion gate mix), their witnesses,
the proof bytes the CPU oracle produces for them (so that the harness checks parity without Python), and the example
block's proof-level DAG (tools/qbench_replay.py, pinned by tests/golden/example_job_dag.json).

File layout (little-endian):  magic "CPQBENCH" | u32 version = 1 | cp_shape as 22 x i32 | u32 n_gates, u32 num_selectors,
n_gates x 7 x i32 | u32 n_circuits | per circuit: 4 x u64 digest, u64 rows, u64 cols, rows*cols x u64 constants+sigmas values,
u32 n_pi, n_pi x u64, u64 wire rows, u64 wire cols, wires, u64 proof_len, proof bytes | u32 n_tasks | per task: u32 n_deps,
n_deps x u32.      usage: dump_qbench_case.py [out = tools/qbench_case.bin] [n_circuits = 4]"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("city-rollup_amd", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
import synth_gates as SG  # noqa: E402


def recursion_mix(poseidon_fraction=0.6):
    rest = 1.0 - poseidon_fraction
    return {SG.POSEIDON: poseidon_fraction, SG.ARITHMETIC: 0.3 * rest, SG.ARITHMETIC_EXT: 0.2 * rest, SG.MUL_EXT: 0.125 * rest,
            SG.REDUCING: 0.05 * rest, SG.REDUCING_EXT: 0.05 * rest, SG.RANDOM_ACCESS: 0.075 * rest, SG.BASE_SUM: 0.05 * rest,
            SG.COSET_INTERPOLATION: 0.05 * rest, SG.POSEIDON_MDS: 0.05 * rest, SG.COMPARISON: 0.05 * rest}


def block_dag():
    import qbench_replay
    return qbench_replay.block_dag()


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "qbench_case.bin")
    n_circuits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    kw = dict(db=12, num_routed=80, num_wires=135, chunk=8, rate_bits=3, arity_bits=(4, 4), cap_height=4, pow_bits=16,
              num_query_rounds=28, n_copies=64)
    cases = [SG.build_gate_set(SG.CITY_COMMON, seed=i, weights=recursion_mix(), noop_fraction=0.03, **kw) for i in range(n_circuits)]
    sh = cases[0]["shape"]
    O.lib().or_set_threads(os.cpu_count() or 1)
    with open(out, "wb") as f:
        f.write(b"CPQBENCH" + struct.pack("<I", 1))
        ab = list(sh.arity_bits)[:8]
        f.write(struct.pack("<22i", sh.degree_bits, sh.num_constants, sh.num_routed_wires, sh.num_wires, sh.num_challenges,
                            sh.num_partial_products, sh.quotient_degree_factor, sh.rate_bits, sh.cap_height, sh.pow_bits,
                            sh.num_query_rounds, sh.n_arity, *ab, 0, len(cases[0]["public_inputs"])))
        gl = cases[0]["gate_list"]
        f.write(struct.pack("<II", len(gl), cases[0]["num_selectors"]))
        for g in gl:
            f.write(struct.pack("<7i", *g))
        f.write(struct.pack("<I", n_circuits))
        for i, c in enumerate(cases):
            assert c["gate_list"] == gl
            digest = [i, 1, 2, 3]
            proof, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
            cs, w = np.ascontiguousarray(c["cs_values"], np.uint64), np.ascontiguousarray(c["wires"], np.uint64)
            pi = np.asarray(c["public_inputs"], np.uint64)
            f.write(struct.pack("<4Q", *digest) + struct.pack("<QQ", *cs.shape) + cs.tobytes())
            f.write(struct.pack("<I", len(pi)) + pi.tobytes())
            f.write(struct.pack("<QQ", *w.shape) + w.tobytes())
            f.write(struct.pack("<Q", len(proof)) + bytes(proof))
        dag = block_dag()
        index = {name: k for k, (name, _) in enumerate(dag)}
        f.write(struct.pack("<I", len(dag)))
        for name, deps in dag:
            f.write(struct.pack("<I", len(deps)) + struct.pack("<%dI" % len(deps), *[index[d] for d in deps]))
    print("wrote %s (%d bytes, %d circuits, %d DAG tasks)" % (out, os.path.getsize(out), n_circuits, len(dag)))


if __name__ == "__main__":
    main()
