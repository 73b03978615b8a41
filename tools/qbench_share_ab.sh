#!/bin/bash
# A/B: the short-queue sharing rule (ready stages split among the workers that hold no work) with many blocks in flight.
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/qbench_share_ab.jsonl"
PACK=/tmp/qbench_share_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 0 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
: > "$OUT"
for rep in 1 2 3; do for f in 4 8 16 32 64; do
  n=$((f * 6)); [ $n -lt 48 ] && n=48
  $Q -i $D --pack $PACK --contexts 3 -n $n --blocks-in-flight $f | tail -1 >> "$OUT"
  CITYPROVER_QBENCH_SHARE_ALWAYS=1 $Q -i $D --pack $PACK --contexts 3 -n $n --blocks-in-flight $f | tail -1 >> "$OUT"
done; done
wc -l "$OUT"
