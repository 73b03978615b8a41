#!/bin/bash
# A/B: stages as the unit of scheduling (default) against whole jobs held by one worker (CITYPROVER_QBENCH_WHOLE_JOBS=1).
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/qbench_stage_ab.jsonl"
PACK=/tmp/qbench_stage_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 0 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
: > "$OUT"
for rep in 1 2; do for f in 1 4 8 16 32 64; do
  n=$((f * 4))
  CITYPROVER_QBENCH_WHOLE_JOBS=1 $Q -i $D --pack $PACK --contexts 3 -n $n --blocks-in-flight $f | tail -1 >> "$OUT"
  $Q -i $D --pack $PACK --contexts 3 -n $n --blocks-in-flight $f | tail -1 >> "$OUT"
done; done
$Q -i $D --pack $PACK --contexts 3 -n 3 --trace "$R/gpurun_out/stage_trace_c3.jsonl" | tail -1 >> "$OUT"
wc -l "$OUT"
