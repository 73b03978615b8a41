#!/bin/bash
# Waves of F blocks (a batch of F: BASELINE.json configs[3]) against a sliding window of F blocks (steady state), stage-unit scheduling,
# older block / longest chain first. -> gpurun_out/qbench_window_ab.jsonl
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/qbench_window_ab.jsonl"
PACK=/tmp/qbench_window_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 0 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
: > "$OUT"
for rep in 1 2; do for f in 2 4 8 16 32 64; do
  n=$((f * 8)); [ $n -lt 32 ] && n=32
  $Q -i $D --pack $PACK --contexts 3 -n $n --blocks-in-flight $f | tail -1 >> "$OUT"
  $Q -i $D --pack $PACK --contexts 3 -n $n --blocks-in-flight $f --sliding | tail -1 >> "$OUT"
done; done
$Q -i $D --pack $PACK --contexts 3 -n 4 | tail -1 >> "$OUT"
wc -l "$OUT"
