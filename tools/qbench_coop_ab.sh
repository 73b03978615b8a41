#!/bin/bash
# A/B of the lane-per-state <-> twelve-lanes-per-state switch of the Merkle levels (CITYPROVER_COOP_MAX: a level with at most that
# many parents over the whole batch takes the cooperative permutation) on the section 8(d) workload.
# Run through gpurun from the repo root -> gpurun_out/qbench_coop_ab.jsonl
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/qbench_coop_ab.jsonl"
PACK=/tmp/qbench_coop_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 0 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
: > "$OUT"
for m in 16384 4096 1024 256 16384 1024; do
  export CITYPROVER_COOP_MAX=$m
  echo "{\"coop_max\": $m}" >> "$OUT"
  $Q --mode throughput --pack $PACK --contexts 3 --batch 32 --iters 8 | tail -1 >> "$OUT"
  $Q -i $D --pack $PACK --contexts 3 --batch 32 -n 32 --blocks-in-flight 32 | tail -1 >> "$OUT"
  $Q -i $D --pack $PACK --contexts 3 --batch 32 -n 4 | tail -1 >> "$OUT"
done
wc -l "$OUT"
