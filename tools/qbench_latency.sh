#!/bin/bash
# One block alone (latency) under the batching worker and under one-job-per-call threads through cp_batcher.
# Run through gpurun from the repo root -> gpurun_out/qbench_latency.jsonl
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/qbench_latency.jsonl"
PACK=/tmp/qbench_latency_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 4 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
: > "$OUT"
for rep in 1 2; do
  $Q -i $D --pack $PACK --contexts 3 --batch 32 -n 4 | tail -1 >> "$OUT"
  $Q -i $D --pack $PACK --contexts 1 --lanes 4 --callers 32 --batch 32 -n 4 | tail -1 >> "$OUT"
  $Q -i $D --pack $PACK --contexts 1 --lanes 4 --callers 32 --batch 32 --linger-us 100 -n 4 | tail -1 >> "$OUT"
  $Q -i $D --pack $PACK --contexts 1 --lanes 2 --callers 24 --batch 32 -n 4 | tail -1 >> "$OUT"
  $Q -i $D --pack $PACK --contexts 3 --lanes 1 --callers 8 --batch 32 -n 4 | tail -1 >> "$OUT"
  $Q -i $D --pack $PACK --contexts 6 --batch 8 -n 4 | tail -1 >> "$OUT"
done
$Q -i $D --pack $PACK --contexts 1 --lanes 4 --callers 64 --batch 32 --linger-us 200 -n 8 --blocks-in-flight 8 | tail -1 >> "$OUT"
$Q -i $D --pack $PACK --contexts 3 --batch 32 -n 8 --blocks-in-flight 8 | tail -1 >> "$OUT"
wc -l "$OUT"
