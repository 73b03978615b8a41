#!/bin/bash
# rocprofv3 evidence of round 4 on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the bench command                      -> r04_bench_kernel_stats.csv (+ the bench line under rocprof)
#   2-4. --pmc SQ group / FETCH_SIZE / WRITE_SIZE of the bench command  -> r04_pmc_bench.json
#   5. --kernel-trace --stats of the native q-bench harness (8 blocks in flight, 3 contexts: the section 8(d) workload) -> r04_prove_kernel_stats.csv
#   6-8. --pmc passes of the harness in throughput mode, ONE context    -> r04_pmc_qbench.json (per proof: VALU instructions, HBM bytes, quotient traffic)
#   9. --kernel-trace --stats of the G1 / G2 MSMs at 2^20               -> r04_msm_kernel_stats.csv; one proof alone -> r04_prove_profile_1.json;
#      STARK-shaped commit + FRI -> r04_stark_commit_fri.json; the AIR quotient / whole STARK prover -> r04_stark_air.json (+ kernel
#      trace and SQ counters of the interpreter launch)
# Counter passes run alone (no trace flags: gpurun refuses the combination). The program itself follows `--`.
set -e
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/prof_r04"
rm -rf "$OUT"; mkdir -p "$OUT"
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
if [ "$1" != "qbench-only" ]; then
BENCH="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-qbench"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
python3 "$R/tools/rocpd_top_kernels.py" "$OUT/trace" "rocprofv3 --kernel-trace --stats -- $BENCH" > "$OUT/r04_bench_kernel_stats.csv" || true
PMCB="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-qbench"
rocprofv3 --pmc $SQ -d "$OUT/pmc_sq" -o pmc --output-format csv -- $PMCB > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc --output-format csv -- $PMCB > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc --output-format csv -- $PMCB > /dev/null 2> "$OUT/pmc_write.err"
mkdir -p "$OUT/benchpmc" && mv "$OUT/pmc_sq" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/benchpmc/"
python3 "$R/tools/pmc_summary.py" "$OUT/benchpmc" "$OUT/r04_pmc_bench.json" "rocprofv3 --pmc <group> -- $PMCB"
fi
# the whole-proof path
python3 "$R/tools/make_circuit_pack.py" /tmp/prof_pack 0 12 > /dev/null
QB="$R/tools/cityprover_qbench -i $R/tests/golden/qbench_example.bin -n 8 --blocks-in-flight 8 --pack /tmp/prof_pack --contexts 3 --batch 32"
rocprofv3 --kernel-trace --stats -d "$OUT/trace_qbench" -o qbench -- $QB > "$OUT/qbench_under_rocprof.json" 2> "$OUT/trace_qbench.err"
python3 "$R/tools/rocpd_top_kernels.py" "$OUT/trace_qbench" "rocprofv3 --kernel-trace --stats -- tools/cityprover_qbench -i tests/golden/qbench_example.bin -n 8 --blocks-in-flight 8 --pack <section 8(d) pack> --contexts 3 --batch 32" > "$OUT/r04_prove_kernel_stats.csv" || true
ITERS=4
QT="$R/tools/cityprover_qbench --mode throughput --skip-gate --pack /tmp/prof_pack --contexts 1 --batch 32 --iters $ITERS"
PROOFS=$((32 + 32 * ITERS))   # one full warm-up batch and the timed batches: every launch of the run carries 32 proofs (no gate: the counter run measures, the other runs check)
rocprofv3 --pmc $SQ -d "$OUT/qpmc/sq" -o pmc --output-format csv -- $QT > "$OUT/qbench_under_pmc.json" 2> "$OUT/qpmc_sq.err"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/qpmc/fetch" -o pmc --output-format csv -- $QT > /dev/null 2> "$OUT/qpmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/qpmc/write" -o pmc --output-format csv -- $QT > /dev/null 2> "$OUT/qpmc_write.err"
python3 "$R/tools/pmc_summary_qbench.py" "$OUT/qpmc" "$OUT/r04_pmc_qbench.json" $PROOFS "rocprofv3 --pmc <group> -- tools/cityprover_qbench --mode throughput --skip-gate --pack <section 8(d) pack> --contexts 1 --batch 32 --iters $ITERS"
# the MSM kernels at 2^20 (G1 and G2): per-kernel time beside the code-object metadata of profiles/r04_msm_kernel_meta.csv
rocprofv3 --kernel-trace --stats -d "$OUT/trace_msm" -o msm -- python3 "$R/tools/bench_msm.py" 20 > "$OUT/r04_msm_bench.json" 2> "$OUT/trace_msm.err"
python3 "$R/tools/rocpd_top_kernels.py" "$OUT/trace_msm" "rocprofv3 --kernel-trace --stats -- python3 tools/bench_msm.py 20" > "$OUT/r04_msm_kernel_stats.csv" || true
# VALU instructions of the MSM kernels (the roof of the bucket method is the issue rate of v_mad_u64_u32): bench_msm.py 20 makes one
# warm-up + three timed MSMs per group
rocprofv3 --pmc $SQ -d "$OUT/msmpmc" -o pmc --output-format csv -- python3 "$R/tools/bench_msm.py" 20 > /dev/null 2> "$OUT/msmpmc.err" || true
python3 "$R/tools/pmc_summary_msm.py" "$OUT/msmpmc" "$OUT/r04_pmc_msm.json" 20 4 || true
# one proof alone: per-kernel and per-host-phase milliseconds (host transcript for a single proof, device transcript from 8 up)
python3 "$R/tools/prove_profile_one.py" > "$OUT/r04_prove_profile_1.json" 2> "$OUT/prove_profile_1.err"
python3 "$R/tools/bench_stark_fri.py" > "$OUT/r04_stark_commit_fri.json" 2> "$OUT/stark_fri.err" || true
# the STARK's own two steps on the generic AIR machinery: quotient alone + the whole prover, and the interpreter launch under rocprofv3
python3 "$R/tools/bench_stark_air.py" 10 12 14 16 > "$OUT/r04_stark_air.json" 2> "$OUT/stark_air.err" || true
rocprofv3 --kernel-trace --stats -d "$OUT/trace_air" -o air -- python3 "$R/tools/bench_stark_air.py" 14 > /dev/null 2> "$OUT/trace_air.err" || true
python3 "$R/tools/rocpd_top_kernels.py" "$OUT/trace_air" "rocprofv3 --kernel-trace --stats -- python3 tools/bench_stark_air.py 14" > "$OUT/r04_stark_air_kernel_stats.csv" || true
rocprofv3 --pmc $SQ -d "$OUT/airpmc" -o pmc --output-format csv -- python3 "$R/tools/bench_stark_air.py" 14 > /dev/null 2> "$OUT/airpmc.err" || true
f=$(find "$OUT/airpmc" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Kernel_Name|k_run|k_finish" "$f" > "$OUT/r04_air_pmc_sq_counter_collection.csv" || true
python3 "$R/tools/kernel_meta.py" msm --csv > "$OUT/r04_msm_kernel_meta.csv" 2>/dev/null || true
# keep the per-kernel counter CSVs small: one merged CSV per group
for g in sq fetch write; do f=$(find "$OUT/qpmc/$g" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/r04_qpmc_${g}_counter_collection.csv"; done
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
