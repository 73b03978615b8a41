#!/bin/bash
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
PACK=/tmp/ctx_matrix_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 4 12 > /dev/null
Q="$R/tools/cityprover_qbench"; D="$R/tests/golden/qbench_example.bin"
for hq in 2 4 8 16; do
  for c in 4 8; do
    echo "{\"GPU_MAX_HW_QUEUES\": $hq}"
    GPU_MAX_HW_QUEUES=$hq $Q -i $D --pack $PACK --contexts $c --batch 1 -n 16 --blocks-in-flight 16 | tail -1
  done
done
echo '{"GPU_MAX_HW_QUEUES": 8, "batch": 32}'
GPU_MAX_HW_QUEUES=8 $Q -i $D --pack $PACK --contexts 3 --batch 32 -n 32 --blocks-in-flight 32 | tail -1
GPU_MAX_HW_QUEUES=8 $Q -i $D --pack $PACK --contexts 6 --batch 32 -n 32 --blocks-in-flight 32 | tail -1
