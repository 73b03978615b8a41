#!/bin/bash
# Fiat-Shamir transcript on the host / on the device / chosen per batch (default), over harness configurations.
set -e
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/qbench_transcript_ab.txt"
PACK=/tmp/qbench_tr_pack
python3 "$R/tools/make_circuit_pack.py" $PACK 0 12 > /dev/null
Q="$R/tools/cityprover_qbench"
D="$R/tests/golden/qbench_example.bin"
echo "# CITYPROVER_DEVICE_TRANSCRIPT = (unset: device from 8 proofs per launch) / 0 (host) / 1 (device); columns: mode contexts blocks_in_flight blocks_per_s proofs_per_s mean_block_latency_ms mean_launch" > "$OUT"
run() { # mode, args...
  local mode="$1"; shift
  if [ "$mode" = auto ]; then r=$($Q "$@" | tail -1); else r=$(CITYPROVER_DEVICE_TRANSCRIPT=$mode $Q "$@" | tail -1); fi
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('$mode', d['contexts_per_device'], d.get('blocks_in_flight'), d['blocks_per_s'], d['proofs_per_s'], d.get('mean_block_latency_ms'), d.get('mean_batch'))" "$r" >> "$OUT"
}
for mode in auto 0 1 auto 0 1; do
  run $mode -i $D --pack $PACK --contexts 3 -n 4
  run $mode -i $D --pack $PACK --contexts 3 -n 32 --blocks-in-flight 8
  run $mode -i $D --pack $PACK --contexts 3 -n 128 --blocks-in-flight 32
  run $mode -i $D --pack $PACK --contexts 3 -n 256 --blocks-in-flight 64
  run $mode --mode throughput --pack $PACK --contexts 3 --batch 64 --iters 8
done
cat "$OUT"
