#!/bin/bash
# One block alone with a per-job timeline, over a few worker configurations -> gpurun_out/one_block/*.jsonl
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/one_block"; rm -rf "$OUT"; mkdir -p "$OUT"
PACK=/tmp/qbench_ob_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 0 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
for c in 3 4 6 8; do
  GPU_MAX_HW_QUEUES=8 $Q -i $D --pack $PACK --contexts $c -n 3 --trace "$OUT/trace_c$c.jsonl" | tail -1 > "$OUT/summary_c$c.json"
done
CITYPROVER_QBENCH_NO_SHARE=1 $Q -i $D --pack $PACK --contexts 3 -n 3 --trace "$OUT/trace_c3_noshare.jsonl" | tail -1 > "$OUT/summary_c3_noshare.json"
ls "$OUT"
