#!/bin/bash
# The q-bench harness with and without the three STARK proofs of a block (tools/qbench/stark_stage.h: --stark-log-rows K runs
# cp_stark_prove on a synthetic AIR of the reference's shape before each sighash job). One JSON line per setting.
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
python3 "$R/tools/make_circuit_pack.py" /tmp/ab_pack 0 12 > /dev/null
for k in 0 10 11 12; do
  "$R/tools/cityprover_qbench" -i "$R/tests/golden/qbench_example.bin" -n 128 --blocks-in-flight 64 --pack /tmp/ab_pack --contexts 3 --batch 128 --stark-log-rows $k
done
for c in 4 5; do
  "$R/tools/cityprover_qbench" -i "$R/tests/golden/qbench_example.bin" -n 128 --blocks-in-flight 64 --pack /tmp/ab_pack --contexts $c --batch 128 --stark-log-rows 10
done
# contexts that prove nothing but STARK stages, beside three that prove none
for c in 1 2 3; do
  "$R/tools/cityprover_qbench" -i "$R/tests/golden/qbench_example.bin" -n 128 --blocks-in-flight 64 --pack /tmp/ab_pack --contexts 3 --batch 128 --stark-log-rows 10 --stark-contexts $c
done
"$R/tools/cityprover_qbench" -i "$R/tests/golden/qbench_example.bin" -n 8 --blocks-in-flight 1 --pack /tmp/ab_pack --contexts 3 --batch 128 --stark-log-rows 10 --stark-contexts 3
"$R/tools/cityprover_qbench" -i "$R/tests/golden/qbench_example.bin" -n 8 --blocks-in-flight 1 --pack /tmp/ab_pack --contexts 3 --batch 128 --stark-log-rows 0
"$R/tools/cityprover_qbench" -i "$R/tests/golden/qbench_example.bin" -n 8 --blocks-in-flight 1 --pack /tmp/ab_pack --contexts 3 --batch 128 --stark-log-rows 10
