#!/bin/bash
# The q-bench DAG drained by reference-style loops (one job per pop, one proof per call) sharing a context through cp_batcher;
# appended to gpurun_out/qbench_callers.jsonl by tools/qbench_callers.sh, or run alone -> gpurun_out/qbench_callers_dag.jsonl
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="${1:-$R/gpurun_out/qbench_callers_dag.jsonl}"
PACK=/tmp/qbench_callers_pack
[ -d $PACK ] || python3 "$R/tools/make_circuit_pack.py" $PACK 4 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
# contexts, lanes, caller threads per context, linger (us), blocks in flight
for cfg in "1 4 64 300 16" "1 4 128 300 32" "1 4 192 300 64" "3 1 64 300 32" "2 2 96 300 64" "1 4 128 0 32" "1 4 128 1000 64"; do
  set -- $cfg
  $Q -i $D --pack $PACK --contexts $1 --lanes $2 --callers $3 --linger-us $4 --batch 32 -n $5 --blocks-in-flight $5 | tail -1 >> "$OUT"
done
wc -l "$OUT"
