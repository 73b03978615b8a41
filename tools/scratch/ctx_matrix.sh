#!/bin/bash
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
PACK=/tmp/ctx_matrix_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 4 12 > /dev/null
Q="$R/tools/cityprover_qbench"; D="$R/tests/golden/qbench_example.bin"
for c in 2 4 8 12 16; do
  $Q -i $D --pack $PACK --contexts $c --batch 1 -n 16 --blocks-in-flight 16 | tail -1
done
for c in 4 8; do
  $Q -i $D --pack $PACK --contexts $c --batch 4 -n 16 --blocks-in-flight 16 | tail -1
done
