#!/bin/bash
# contexts per GPU x blocks in flight with the round-3 scheduler -> gpurun_out/qbench_ctx_ab.jsonl
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/qbench_ctx_ab.jsonl"
PACK=/tmp/qbench_ctx_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 0 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
: > "$OUT"
export GPU_MAX_HW_QUEUES=8
for rep in 1 2; do for c in 2 3 4 5 6; do
  $Q -i $D --pack $PACK --contexts $c -n 8 | tail -1 >> "$OUT"
  $Q -i $D --pack $PACK --contexts $c -n 64 --blocks-in-flight 8 | tail -1 >> "$OUT"
  $Q -i $D --pack $PACK --contexts $c -n 256 --blocks-in-flight 64 | tail -1 >> "$OUT"
done; done
wc -l "$OUT"
