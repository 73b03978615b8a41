#!/bin/bash
# One-job-per-call threads through cp_batcher ("callers"): run through gpurun from the repo root;
# one JSON line per run -> gpurun_out/qbench_callers.jsonl. Every proof is compared with the oracle's bytes in the pack.
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/qbench_callers.jsonl"
PACK=/tmp/qbench_callers_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 4 12 > /dev/null
Q="$R/tools/cityprover_qbench"
: > "$OUT"
# contexts, lanes per context, caller threads, linger (us)
for cfg in "1 1 8 0" "1 1 32 0" "1 1 64 0" "1 3 64 0" "1 3 64 200" "1 3 64 500" "1 3 128 500" "1 4 128 500" "3 1 64 0" "3 1 128 0" "3 1 128 200" \
           "3 1 128 500" "3 1 192 500" "2 2 128 500"; do
  set -- $cfg
  $Q --mode callers --pack $PACK --contexts $1 --lanes $2 --callers $3 --batch 32 --linger-us $4 --iters 24 | tail -1 >> "$OUT"
done
"$R/tools/qbench_callers_dag.sh" "$OUT"
