#!/bin/bash
# rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the bench command          -> gpurun_out/prof_r02/trace   (+ the bench line under rocprof)
#   2. --pmc SQ counters (VALU instructions, busy, stalls, clock)   \
#   3. --pmc FETCH_SIZE                                             |-> gpurun_out/prof_r02/pmc_*  -> r02_pmc_bench.json
#   4. --pmc WRITE_SIZE                                             /
#   5. --kernel-trace --stats of the native q-bench harness (whole proofs)  -> prove_kernel_stats.csv
# Counter passes run alone (no trace flags: gpurun refuses the combination). The program itself follows `--`.
set -e
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
OUT="$R/gpurun_out/prof_r02"
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-qbench"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
python3 "$R/tools/rocpd_top_kernels.py" "$OUT/trace" "rocprofv3 --kernel-trace --stats -- $BENCH" > "$OUT/bench_kernel_stats.csv" || true
PMCB="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-qbench"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -o pmc --output-format csv -- $PMCB > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc --output-format csv -- $PMCB > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc --output-format csv -- $PMCB > /dev/null 2> "$OUT/pmc_write.err"
python3 "$R/tools/pmc_summary.py" "$OUT" "$OUT/r02_pmc_bench.json" "rocprofv3 --pmc <group> -- $PMCB"
# 5. --kernel-trace --stats of the whole-proof path: the native q-bench harness (8 example blocks in flight, 3 contexts)
python3 "$R/tools/make_circuit_pack.py" /tmp/prof_pack 4 12 > /dev/null
QB="$R/tools/cityprover_qbench -i $R/tests/golden/qbench_example.bin -n 8 --blocks-in-flight 8 --pack /tmp/prof_pack --contexts 3 --batch 32"
rocprofv3 --kernel-trace --stats -d "$OUT/trace_qbench" -o qbench -- $QB > "$OUT/qbench_under_rocprof.json" 2> "$OUT/trace_qbench.err"
python3 "$R/tools/rocpd_top_kernels.py" "$OUT/trace_qbench" "rocprofv3 --kernel-trace --stats -- tools/cityprover_qbench -i tests/golden/qbench_example.bin -n 8 --blocks-in-flight 8 --pack <synthetic pack> --contexts 3 --batch 32" > "$OUT/prove_kernel_stats.csv" || true
find "$OUT" -name "*.db" -delete   # the rocpd databases are scratch; the CSV / JSON summaries are what is kept   # the rocpd databases are scratch; the CSV / JSON summaries are what is kept
du -sh "$OUT"
