#!/usr/bin/env python3
"""Writes a SYNTHETIC circuit pack for tools/cityprover_qbench (layout: tools/qbench/pack.h): shape-equivalent stand-ins
for the worker's circuits until real `CircuitData` can be dumped by the Rust side of the bridge —
n = 2^12, 135 wires / 80 routed, the 14-gate city-common set of pad_circuit.rs:31-55 in plonky2's selector grouping, rows
drawn with the recursion-circuit mix (~60 % Poseidon), random satisfying witnesses. Every witness file records the proof
bytes the CPU oracle produces for it, so the harness checks byte parity of every proof it makes without Python.

Every job type of the example block is bound (round-robin) to one of `n_circuits` distinct circuits; jobs that prove
several circuits in the reference (root aggregators + minifier, sighash = inner + 3 minifiers + wrapper, ...) get that many
stages (tools/qbench/jobs.h proofs_per_job).

usage: make_circuit_pack.py [out_dir = tools/qbench_pack] [n_circuits = 4] [degree_bits = 12]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("city-rollup_amd", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
import synth_gates as SG  # noqa: E402
from cityprover import files  # noqa: E402

# (circuit_type, proofs per job): the job types plan_jobs can schedule (job_id.rs:87-125)
JOB_TYPES = [(t, 1) for t in list(range(12)) + [36] + list(range(48, 54))] + [(40, 2), (41, 2), (32, 2), (34, 2), (33, 5)]


def recursion_mix(poseidon_fraction=0.6):
    rest = 1.0 - poseidon_fraction
    return {SG.POSEIDON: poseidon_fraction, SG.ARITHMETIC: 0.3 * rest, SG.ARITHMETIC_EXT: 0.2 * rest, SG.MUL_EXT: 0.125 * rest,
            SG.REDUCING: 0.05 * rest, SG.REDUCING_EXT: 0.05 * rest, SG.RANDOM_ACCESS: 0.075 * rest, SG.BASE_SUM: 0.05 * rest,
            SG.COSET_INTERPOLATION: 0.05 * rest, SG.POSEIDON_MDS: 0.05 * rest, SG.COMPARISON: 0.05 * rest}


class ShapeView:
    """cp_shape fields from the oracle's shape + the number of public inputs (files.shape_ints reads attributes)"""

    def __init__(self, s, n_pi):
        for f in ("degree_bits", "num_constants", "num_routed_wires", "num_wires", "num_challenges", "num_partial_products",
                  "quotient_degree_factor", "rate_bits", "cap_height", "pow_bits", "num_query_rounds", "n_arity", "zero_knowledge"):
            setattr(self, f, int(getattr(s, f)))
        self.arity_bits = [int(s.arity_bits[i]) for i in range(8)]
        self.num_public_inputs = n_pi


def make_pack(out, n_circuits=4, db=12, small=False):
    os.makedirs(out, exist_ok=True)
    if small:   # test-sized circuits: same gate set, fewer rows / queries
        kw = dict(db=db, num_routed=80, num_wires=135, chunk=8, rate_bits=3, arity_bits=(2,), cap_height=2, pow_bits=4,
                  num_query_rounds=4, n_copies=2)
    else:
        kw = dict(db=db, num_routed=80, num_wires=135, chunk=8, rate_bits=3, arity_bits=(4, 4), cap_height=4, pow_bits=16,
                  num_query_rounds=28, n_copies=64)
    O.lib().or_set_threads(os.cpu_count() or 1)
    names = []
    for i in range(n_circuits):
        c = SG.build_gate_set(SG.CITY_COMMON, seed=i, weights=recursion_mix(), noop_fraction=0.03, **kw)
        digest = [i, 1, 2, 3]
        proof, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
        sh = ShapeView(c["shape"], len(c["public_inputs"]))
        cname, wname = "synthetic_%d.cpcirc" % i, "synthetic_%d.cpwit" % i
        files.write_circuit_file(os.path.join(out, cname), sh, digest, c["gate_list"], c["num_selectors"], c["cs_values"],
                                 k_is=[int(x) for x in c["k_is"]])
        files.write_witness_file(os.path.join(out, wname), digest, c["wires"], c["public_inputs"], proof=proof)
        names.append((cname, wname))
    with open(os.path.join(out, "pack.manifest"), "w") as f:
        f.write("# synthetic shape-equivalent circuits (tools/make_circuit_pack.py): <circuit_type> <stage> <circuit> <witness>\n")
        k = 0
        for t, stages in JOB_TYPES:
            for s in range(stages):
                f.write("%d %d %s %s\n" % (t, s, *names[k % n_circuits]))
                k += 1
    return out


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "qbench_pack")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    db = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    make_pack(out, n, db, small=db < 12)
    print("wrote", out, sorted(os.listdir(out)))
