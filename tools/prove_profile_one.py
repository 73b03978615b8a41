"""Per-kernel and per-host-phase milliseconds of ONE proof proved alone (the reference's single-threaded loop: one job per
call) — where the latency of a proof goes when nothing else shares the GPU. Prints one JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("city-rollup_amd", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, d))
import cityprover as cp
import bench_prove
p = cp.Prover(0)
r = bench_prove.run(p, 1, 20, profile=True, host_wires=True)
print(json.dumps(r))
p.close()
