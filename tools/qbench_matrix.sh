#!/bin/bash
# The q-bench harness over the configurations DESIGN.md section 6 tabulates (run through gpurun from the repo root):
# one JSON line per run -> gpurun_out/qbench_matrix.jsonl. Every proof is compared with the oracle's bytes in the pack.
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/qbench_matrix.jsonl"
PACK=/tmp/qbench_matrix_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 4 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
: > "$OUT"
$Q -i $D --pack $PACK --contexts 1 --batch 1 | tail -1 >> "$OUT"                                        # the reference's loop
$Q -i $D --pack $PACK --contexts 3 --batch 32 | tail -1 >> "$OUT"                                       # one block alone
$Q -i $D --pack $PACK --contexts 3 --batch 32 -n 8 --blocks-in-flight 8 | tail -1 >> "$OUT"
$Q -i $D --pack $PACK --contexts 3 --batch 32 -n 8 --blocks-in-flight 8 | tail -1 >> "$OUT"
$Q -i $D --pack $PACK --contexts 3 --batch 32 -n 32 --blocks-in-flight 32 | tail -1 >> "$OUT"
$Q -i $D --pack $PACK --contexts 3 --batch 32 -n 64 --blocks-in-flight 64 | tail -1 >> "$OUT"
$Q -i $D --pack $PACK --contexts 3 --batch 32 -n 64 --blocks-in-flight 64 | tail -1 >> "$OUT"
$Q --mode throughput --pack $PACK --contexts 3 --batch 32 --iters 8 | tail -1 >> "$OUT"
$Q -i $D --pack $PACK --contexts 3 --batch 32 -n 8 --blocks-in-flight 8 --groth16-log-size 20 | tail -1 >> "$OUT"
$Q -i $D --pack $PACK --contexts 3 --batch 32 -n 4 --blocks-in-flight 4 --groth16-log-size 22 | tail -1 >> "$OUT"
$Q -i $D --pack $PACK --contexts 3 --batch 32 --groth16-log-size 22 | tail -1 >> "$OUT"
wc -l "$OUT"
# reference-style workers sharing the GPU: N contexts, each one job at a time (the harness raises GPU_MAX_HW_QUEUES itself)
for c in 4 8 12; do
  $Q -i $D --pack $PACK --contexts $c --batch 1 -n 16 --blocks-in-flight 16 | tail -1 >> "$OUT"
done
wc -l "$OUT"
