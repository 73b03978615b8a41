#!/bin/bash
# Soak of the harness's scheduler on the section 8(d) pack: every proof of every run is compared with the bytes that passed the gate.
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/r03_soak.jsonl"
PACK=/tmp/qbench_soak_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 0 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
: > "$OUT"
$Q -i $D --pack $PACK --contexts 3 -n 1024 --blocks-in-flight 64 | tail -1 >> "$OUT"
$Q -i $D --pack $PACK --contexts 3 -n 512 --blocks-in-flight 8 | tail -1 >> "$OUT"
$Q -i $D --pack $PACK --contexts 1 --lanes 4 --callers 192 --batch 64 --linger-us 300 -n 512 --blocks-in-flight 64 | tail -1 >> "$OUT"
$Q -i $D --pack $PACK --contexts 3 -n 256 | tail -1 >> "$OUT"
$Q -i $D --pack $PACK --contexts 3 -n 512 --blocks-in-flight 32 --sliding | tail -1 >> "$OUT"
wc -l "$OUT"
