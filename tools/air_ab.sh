#!/bin/bash
# A/B of the AIR interpreter's knobs (points per lane, LDS slot budget, waves per launch) on the SHA-256-STARK-shaped quotient:
# one JSON line per setting into gpurun_out/r04_air_ab3.jsonl. usage: tools/air_ab.sh [log_rows ...]
set -e
mkdir -p gpurun_out
out=gpurun_out/r04_air_ab3.jsonl
: > $out
sizes="${@:-10 14 16}"
for k in 1 2; do for lds in 4 6 8 10 12; do for tw in 8192 16384; do
  echo "{\"points_per_lane\": $k, \"lds_slots\": $lds, \"target_waves\": $tw, \"run\": $(CITYPROVER_AIR_POINTS_PER_LANE=$k CITYPROVER_AIR_LDS_SLOTS=$lds CITYPROVER_AIR_TARGET_WAVES=$tw python tools/bench_stark_air.py $sizes)}" >> $out
  echo "done k=$k lds=$lds tw=$tw"
done; done; done
