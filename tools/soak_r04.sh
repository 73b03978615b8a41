#!/bin/bash
# Soak of the last build: 512 example blocks in waves of 64 with every plonky2 proof byte-checked, the three STARKs of each block on the
# device and the three Groth16 proofs on a 2^16 synthetic key; then 256 blocks through cp_batcher with 160 one-job-per-call threads.
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
python3 "$R/tools/make_circuit_pack.py" /tmp/soak_pack 0 12 > /dev/null
D="$R/tests/golden/qbench_example.bin"
"$R/tools/cityprover_qbench" -i "$D" -n 512 --blocks-in-flight 64 --pack /tmp/soak_pack --contexts 3 --batch 128 --stark-log-rows 10 --groth16-log-size 16
"$R/tools/cityprover_qbench" -i "$D" -n 256 --blocks-in-flight 64 --pack /tmp/soak_pack --contexts 1 --lanes 4 --callers 160 --batch 64 --linger-us 300
"$R/tools/cityprover_qbench" -i "$D" -n 256 --blocks-in-flight 32 --pack /tmp/soak_pack --contexts 3 --batch 128 --sliding --stark-log-rows 11 --stark-contexts 2
