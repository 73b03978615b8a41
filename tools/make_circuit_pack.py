#!/usr/bin/env python3
"""Writes a SYNTHETIC circuit pack for tools/cityprover_qbench (layout: tools/qbench/pack.h): shape-equivalent stand-ins
for the worker's circuits until real `CircuitData` can be dumped by the Rust side of the bridge —
n = 2^12, 135 wires / 80 routed, the 14-gate city-common set of pad_circuit.rs:31-55 in plonky2's selector grouping, rows
drawn with the recursion-circuit mix (~60 % Poseidon).

The workload is SURVEY.md §8(d)'s: ONE CIRCUIT PER (job type, stage) the block schedules — jobs that prove several circuits
in the reference (root aggregators + minifier, sighash = inner + 3 minifiers + wrapper, ...) get that many stages
(tools/qbench/jobs.h proofs_per_job) — and ONE WITNESS PER JOB, "random satisfying witness from seed = job index": the
example block's 46 jobs are 64 distinct proofs of 26 distinct circuits. `n_checked` of the witness files (spread over the
job types) also record the proof bytes the CPU oracle produces for them; the harness requires those bytes, runs cp_verify
on every other distinct proof once, and compares every later proof of the same job with the verified bytes.
`n_circuits` > 0 keeps the old reduced pack (that many circuits, one witness each, shared round-robin) for quick tests.

usage: make_circuit_pack.py [out_dir = tools/qbench_pack] [n_circuits = 0 (the full workload)] [degree_bits = 12]"""
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


# jobs of each circuit type in the example block (qbench_data/example.bin: job_config register 4 / claim 2 / transfer 4 /
# add-withdrawal 4 / process-withdrawal 4 / add-deposit 2, three sighash inputs): tools/cityprover_qbench --dry-run --trace
EXAMPLE_BLOCK_JOBS = {0: 4, 1: 3, 2: 2, 3: 1, 4: 2, 5: 1, 6: 4, 7: 3, 8: 4, 9: 3, 10: 4, 11: 3, 32: 1, 33: 3, 34: 3, 36: 3, 40: 1, 41: 1}


def _shape_kw(db, small):
    if small:   # test-sized circuits: same gate set, fewer rows / queries
        return dict(db=db, num_routed=80, num_wires=135, chunk=8, rate_bits=3, arity_bits=(2,), cap_height=2, pow_bits=4,
                    num_query_rounds=4, n_copies=2)
    return dict(db=db, num_routed=80, num_wires=135, chunk=8, rate_bits=3, arity_bits=(4, 4), cap_height=4, pow_bits=16,
                num_query_rounds=28, n_copies=64)


def _one_witness(args):
    """worker of the process pool: (circuit seed, witness seed or None, kw, digest, out dir, names, with_circuit, with_proof)"""
    cseed, wseed, kw, digest, out, cname, wname, with_circuit, with_proof, threads = args
    c = SG.build_gate_set(SG.CITY_COMMON, seed=cseed, weights=recursion_mix(), noop_fraction=0.03, witness_seed=wseed, **kw)
    proof = None
    if with_proof:
        O.lib().or_set_threads(threads)
        proof, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    if with_circuit:
        files.write_circuit_file(os.path.join(out, cname), ShapeView(c["shape"], len(c["public_inputs"])), digest, c["gate_list"],
                                 c["num_selectors"], c["cs_values"], k_is=[int(x) for x in c["k_is"]])
    files.write_witness_file(os.path.join(out, wname), digest, c["wires"], c["public_inputs"], proof=proof)
    return wname


def make_pack(out, n_circuits=0, db=12, small=False, jobs_per_type=None, n_checked=8, processes=None):
    os.makedirs(out, exist_ok=True)
    kw = _shape_kw(db, small)
    cpus = len(os.sched_getaffinity(0))
    if n_circuits > 0:   # the reduced pack: n_circuits circuits with one oracle-proved witness each, bound round-robin
        names = []
        for i in range(n_circuits):
            cname, wname = "synthetic_%d.cpcirc" % i, "synthetic_%d.cpwit" % i
            _one_witness((i, None, kw, [i, 1, 2, 3], out, cname, wname, True, True, cpus))
            names.append((cname, wname))
        with open(os.path.join(out, "pack.manifest"), "w") as f:
            f.write("# synthetic shape-equivalent circuits (tools/make_circuit_pack.py, reduced): <circuit_type> <stage> <circuit> <witness>\n")
            k = 0
            for t, stages in JOB_TYPES:
                for s in range(stages):
                    f.write("%d %d %s %s\n" % (t, s, *names[k % n_circuits]))
                    k += 1
        return out
    jobs_per_type = dict(EXAMPLE_BLOCK_JOBS if jobs_per_type is None else jobs_per_type)
    stages_of = dict(JOB_TYPES)
    tasks, lines = [], []
    bindings = [(t, s) for t, _ in JOB_TYPES if t in jobs_per_type for s in range(stages_of[t])]
    n_wit = sum(jobs_per_type[t] for t, _ in bindings)
    # the oracle-proved sample: every (n_wit / n_checked)-th witness in binding order, so that it spreads over the job types
    checked = set(range(0, n_wit, max(1, n_wit // max(1, n_checked)))) if n_checked > 0 else set()
    checked = set(sorted(checked)[:n_checked])
    procs = processes or max(1, min(cpus, 16))
    wi = 0
    for ci, (t, s) in enumerate(bindings):
        digest = [1000 + ci, t, s, 3]
        cname = "type%d_stage%d.cpcirc" % (t, s)
        wnames = []
        for k in range(jobs_per_type[t]):
            wname = "type%d_stage%d_job%d.cpwit" % (t, s, k)
            # witness seed = the job's index among the block's proofs
            tasks.append((1000 + ci, wi, kw, digest, out, cname, wname, k == 0, wi in checked, max(1, cpus // procs)))
            wnames.append(wname)
            wi += 1
        lines.append("%d %d %s %s\n" % (t, s, cname, " ".join(wnames)))
    # the job types plan_jobs can schedule but this block does not: bound to the first circuit, so that the manifest is complete
    first = tasks[0]
    if procs > 1 and len(tasks) > 1:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(procs) as pool:   # not fork: the parent may already run OpenMP threads (the oracle)
            pool.map(_one_witness, tasks, chunksize=1)
    else:
        for a in tasks:
            _one_witness(a)
    with open(os.path.join(out, "pack.manifest"), "w") as f:
        f.write("# synthetic shape-equivalent circuits (tools/make_circuit_pack.py): one circuit per (job type, stage), one witness per job\n"
                "# <circuit_type> <stage> <circuit> <witness of job 0> [<witness of job 1> ...]\n")
        f.writelines(lines)
        f.write("default 0 %s %s\n" % (first[5], first[6]))
    return out


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "qbench_pack")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    db = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    make_pack(out, n, db, small=db < 12)
    print("wrote", out, sorted(os.listdir(out)))
