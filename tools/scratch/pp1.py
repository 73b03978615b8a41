import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("city-rollup_amd", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, d))
import cityprover as cp
import bench_prove
p = cp.Prover(0)
for B in (1, 4):
    r = bench_prove.run(p, B, 5, profile=True)
    print(json.dumps(r))
p.close()
