#!/bin/bash
# A/B: longest-dependent-chain-first ordering of the ready queue with many blocks in flight (it is the rule of the latency mode).
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/qbench_chain_ab.jsonl"
PACK=/tmp/qbench_chain_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 0 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
: > "$OUT"
for rep in 1 2; do for f in 4 8 16 32 64; do
  $Q -i $D --pack $PACK --contexts 3 -n $((f * 4)) --blocks-in-flight $f | tail -1 >> "$OUT"
  CITYPROVER_QBENCH_CHAIN_ALWAYS=1 $Q -i $D --pack $PACK --contexts 3 -n $((f * 4)) --blocks-in-flight $f | tail -1 >> "$OUT"
done; done
wc -l "$OUT"
