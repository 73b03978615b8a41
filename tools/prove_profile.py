import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("city-rollup_amd", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, d))
import cityprover as cp
import bench_prove
p = cp.Prover(0)
r = bench_prove.run(p, 32, 3, profile=True)
print(json.dumps(r))
p.close()
rt = bench_prove.run_threads(3, 32, 8, device=0, host_wires=True)
print(json.dumps(rt))
