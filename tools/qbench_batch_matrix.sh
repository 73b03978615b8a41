#!/bin/bash
# The q-bench DAG on the section 8(d) pack over max batch x blocks in flight x contexts.
# Run through gpurun from the repo root -> gpurun_out/qbench_batch_matrix.jsonl
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/qbench_batch_matrix.jsonl"
PACK=/tmp/qbench_bm_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 0 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
: > "$OUT"
for c in 3 2 4; do for b in 32 64 128; do for f in 32 64; do
  $Q -i $D --pack $PACK --contexts $c --batch $b -n $((f * 4)) --blocks-in-flight $f | tail -1 >> "$OUT"
done; done; done
for b in 32 64 128; do $Q --mode throughput --pack $PACK --contexts 3 --batch $b --iters 8 | tail -1 >> "$OUT"; done
wc -l "$OUT"
