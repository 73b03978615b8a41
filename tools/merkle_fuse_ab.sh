#!/bin/bash
# headline step (bench.py --no-qbench) under the Merkle fusion knobs: "leaf,level" pairs. -> gpurun_out/r04_merkle_fuse_matrix.jsonl
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/r04_merkle_fuse_matrix.jsonl"; : > "$OUT"
for rnd in 1 2; do for pair in 0,0 1,0 2,0 3,0 0,1 0,2 3,3; do
  lf=${pair%,*}; vf=${pair#*,}
  CITYPROVER_MERKLE_FUSE=$lf CITYPROVER_MERKLE_LEVEL_FUSE=$vf python3 "$R/bench.py" --no-qbench --steps 20 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print(json.dumps({'leaf_fuse': $lf, 'level_fuse': $vf, 'round': $rnd, 'ms_per_step': d['ms_per_step'], 'leaf_hash_ms': k.get('leaf_hash_cols'), 'merkle_levels_ms': d['merkle_levels_ms'], 'levels': {n: v for n, v in k.items() if 'merkle' in n}}))" | tee -a "$OUT"
done; done
