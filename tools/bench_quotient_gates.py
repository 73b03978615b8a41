#!/usr/bin/env python3
"""Per-gate cost of A8: quotient-kernel ms per batch of B proofs (n = 2^12, product shape) for the city-common gate set
and for all 21 gate types. There is one kernel per gate (k_quot_gate<TYPE>), so the per-gate cost is read directly off
the per-kernel HIP-event times."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("city-rollup_amd", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np  # noqa: E402
import cityprover as cp  # noqa: E402
import synth_gates as SG  # noqa: E402
from bench_prove import ProductBackend  # noqa: E402


def measure(prover, gate_set, B):
    kw = dict(db=12, num_routed=80, num_wires=135, chunk=8, rate_bits=3, arity_bits=(4, 4), cap_height=4, pow_bits=16,
              num_query_rounds=28, n_copies=64, backend=ProductBackend(prover))
    c = SG.build_gate_set(gate_set, seed=1, **kw)
    sh = cp.standard_recursion_shape(num_constants=c["num_constants"], num_public_inputs=len(c["public_inputs"]))
    circ = cp.Circuit(prover, sh, [1, 1, 2, 3], c["cs_values"])
    cp.set_gates(circ, c["gate_list"], c["num_selectors"])
    dw = prover.to_device(np.stack([c["wires"]] * B))
    args = ([circ] * B, [c["public_inputs"]] * B, dw.ptr)
    cp.prove_batch_dev(prover, *args)
    prover.profile_begin()
    cp.prove_batch_dev(prover, *args)
    prof = prover.profile_end()
    dw.free()
    circ.close()
    m = {k: round(v["total_ms"], 3) for k, v in prof.items() if k.startswith("quotient")}
    m["total"] = round(sum(m.values()), 3)
    return m


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    p = cp.Prover(0)
    out = {"B": B, "CITY_COMMON": measure(p, SG.CITY_COMMON, B), "ALL_GATES": measure(p, SG.ALL_GATES, B)}
    p.close()
    print(json.dumps(out))
