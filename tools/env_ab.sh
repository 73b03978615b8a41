#!/bin/bash
# A/B of one environment knob of the library on (1) the headline step (bench.py --no-qbench: ms/NTT, leaf hash, tree levels) and
# (2) the q-bench harness (64 blocks in flight, the raw proofs/s loop, one block alone). Alternates the settings ROUNDS times on
# the one box. usage: tools/env_ab.sh VAR "v1 v2 ..." [rounds] [out.jsonl]   (run through gpurun from the repo root)
set -e
VAR="$1"; VALS="$2"; ROUNDS="${3:-2}"; OUT="${4:-gpurun_out/env_ab_${VAR}.jsonl}"
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
mkdir -p "$R/gpurun_out"
PACK=/tmp/env_ab_pack
[ -d $PACK ] || python3 "$R/tools/make_circuit_pack.py" $PACK 0 12 > /dev/null
Q="$R/tools/cityprover_qbench"; D="$R/tests/golden/qbench_example.bin"
: > "$OUT"
for r in $(seq 1 $ROUNDS); do for v in $VALS; do
  export $VAR=$v
  b=$(python3 "$R/bench.py" --no-qbench --steps 20 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels_ms_per_step']
print(json.dumps({'ms_per_step': d['ms_per_step'], 'leaf_hash_ms': k.get('leaf_hash_cols'), 'merkle_levels_ms': d['merkle_levels_ms'], 'perms_per_s': d['poseidon_perms_per_s'], 'levels': {n: v for n, v in k.items() if 'merkle' in n}}))")
  m=$($Q -i $D --pack $PACK --contexts 3 --batch 128 -n 128 --blocks-in-flight 64 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('blocks_per_s','proofs_per_s','mean_batch')}))")
  t=$($Q --mode throughput --pack $PACK --contexts 3 --batch 64 --iters 8 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['proofs_per_s'])")
  o=$($Q -i $D --pack $PACK --contexts 3 --batch 128 -n 8 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['mean_block_latency_ms'])")
  echo "{\"var\": \"$VAR\", \"value\": \"$v\", \"round\": $r, \"bench\": $b, \"qbench_64_in_flight\": $m, \"throughput_proofs_per_s\": $t, \"one_block_ms\": $o}" | tee -a "$OUT"
done; done
