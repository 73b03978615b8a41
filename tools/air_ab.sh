#!/bin/bash
# A/B of the AIR interpreter's knobs (column prefetch, LDS slot budget, waves per launch) on the SHA-256-STARK-shaped quotient:
# one JSON line per setting into gpurun_out/r04_air_ab.jsonl. usage: tools/air_ab.sh [log_rows ...]
set -e
mkdir -p gpurun_out
out=gpurun_out/r04_air_ab.jsonl
: > $out
sizes="${@:-10 14 16}"
for pf in 1 0; do for lds in 16 32 64; do for tw in 8192; do
  echo "{\"prefetch\": $pf, \"lds_slots\": $lds, \"target_waves\": $tw, \"run\": $(CITYPROVER_AIR_PREFETCH=$pf CITYPROVER_AIR_LDS_SLOTS=$lds CITYPROVER_AIR_TARGET_WAVES=$tw python tools/bench_stark_air.py $sizes)}" >> $out
  echo "done prefetch=$pf lds=$lds waves=$tw"
done; done; done
