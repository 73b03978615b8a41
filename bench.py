#!/usr/bin/env python3
"""bench.py — BASELINE.json configs[1]: "single 2^20-row Goldilocks NTT + Poseidon Merkle cap".

One step = one commit-shaped pass over a synthetic 2^20-row x 135-column trace that is already
resident in HBM: 135 forward NTTs of size 2^20 (natural in -> bit-reversed out, in place,
column-major) followed by the Poseidon Merkle cap (height 4) over the 2^20 bit-reversed rows.
Metric: ms per 2^20 NTT, whole job = elapsed / (steps * 135 * n_gpus); weak scaling (every GPU
commits its own trace; the path has no data-path collective).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))

import numpy as np  # noqa: E402

LOG_N = 20
COLS = 135
CAP_H = 4
SEED = 0x243F6A8885A308D3
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# integer-VALU issue peak (MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, one wave64 VALU instruction per 2 cycles per SIMD, 2.4 GHz):
VALU_PEAK_TLOPS = 256 * 4 * 32 * 2.4e9 / 1e12  # = 78.6 T lane-ops/s; only an all-VOP2 stream reaches it (profiles/r01_ubench_valu.txt)
# the sources that define the kernels whose PMC counters are stored under profiles/: the stored counts are used only
# when the hash of these files is the one they were collected with (tools/pmc_summary.py writes it)
KERNEL_SOURCES = ["gl.h", "poseidon.h", "poseidon_tables.h", "merkle.h", "ntt.h", "ntt16.h"]
def _latest_profile(name):
    """profiles/rNN_<name> of the latest round that has one (the counter files are re-collected when the kernels change)"""
    for r in ("r04", "r03"):
        p = os.path.join(ROOT, "profiles", "%s_%s" % (r, name))
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", "r04_" + name)


PMC_JSON = _latest_profile("pmc_bench.json")
PMC_QBENCH_JSON = _latest_profile("pmc_qbench.json")


def permutations_per_proof():
    """Poseidon permutations of one proof at the product shape: leaf hashes of the three
    oracles over the 2^15 LDE rows, their tree levels down to the 16-entry cap, the two FRI layers, the expected proof-of-work
    search (2^16 candidates) and the transcript."""
    N, cap = 1 << 15, 16
    leaves = N * sum((k + 7) // 8 for k in (135, 20, 16))
    levels = 3 * (N - cap)
    fri = sum((n >> 4) * 4 + ((n >> 4) - cap) for n in (N, N >> 4))   # leaf = 16 ext = 32 felts = 4 permutations
    return leaves + levels + fri + (1 << 16) + 115


def qbench_roofline(qb, poseidon_rate):
    """M1 against the roofline that bounds it: the proving path is Poseidon — integer-VALU issue — so proofs/s is priced in
    permutation-equivalents against the permutation rate the leaf hash reaches in THIS run, with the whole path's VALU lane-ops
    and HBM bytes per proof from a rocprofv3 --pmc pass of the harness (tools/profile_r03.sh -> profiles/r03_pmc_qbench.json,
    used only when it was collected with the kernel sources as they are now)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    ppp = permutations_per_proof()
    rate = qb["proofs_per_s"] * ppp
    out = {"bound": "valu-issue (Poseidon)", "unit": "G permutation-equivalents/s", "permutations_per_proof": ppp,
           "achieved": rate / 1e9, "peak": poseidon_rate / 1e9 if poseidon_rate else None,
           "peak_note": "the permutation rate of merkle::k_leaf_hash_cols measured in this run (the headline step's dominant kernel)",
           "frac": rate / poseidon_rate if poseidon_rate else None}
    try:
        import pmc_summary_qbench as P
        d = json.load(open(PMC_QBENCH_JSON))
        if d.get("kernel_source_hash") != P.source_hash():
            out["pmc"] = "refused: " + os.path.relpath(PMC_QBENCH_JSON, ROOT) + " is for other kernel sources (%s, now %s)" % (d.get("kernel_source_hash"), P.source_hash())
            return out
    except (OSError, ValueError, ImportError) as e:
        out["pmc"] = "none: %s" % e
        return out
    pp = d["per_proof"]
    lane_ops_s = pp["valu_lane_ops"] * qb["proofs_per_s"]
    hbm_s = pp["hbm_bytes"] * qb["proofs_per_s"]
    out["pmc"] = {"source": os.path.relpath(PMC_QBENCH_JSON, ROOT) + " (kernel source hash %s)" % d["kernel_source_hash"],
                  "valu_lane_ops_per_proof": pp["valu_lane_ops"], "valu_T_lane_ops_per_s_at_this_rate": lane_ops_s / 1e12,
                  "valu_frac_of_issue_peak": lane_ops_s / 1e12 / VALU_PEAK_TLOPS,
                  "hbm_bytes_per_proof": pp["hbm_bytes"], "hbm_GBs_at_this_rate": hbm_s / 1e9, "hbm_frac": hbm_s / 1e9 / HBM_PEAK_GBS,
                  "algorithmic_bytes_per_proof": 0.16 * (1 << 30), "traffic_over_algorithmic": pp["hbm_bytes"] / (0.16 * (1 << 30)),
                  "kernel_busy_us_per_proof_one_context": pp["kernel_busy_us"], "quotient": d["quotient"],
                  # a SIMD issues one wave64 VALU instruction per four cycles (the leaf hash measures 4.0: roofline.cycles_per_valu_...):
                  # the proofs/s at which the instructions of a proof fill all 1 024 SIMDs at 2.4 GHz, and how close the runs come
                  "issue_ceiling_proofs_per_s": 1024 * 2.4e9 / 4.0 / pp["valu_instructions"],
                  "frac_of_issue_ceiling": qb["proofs_per_s"] * pp["valu_instructions"] * 4.0 / (1024 * 2.4e9),
                  "frac_of_issue_ceiling_throughput_mode": (qb.get("throughput_mode_proofs_per_s") or 0) * pp["valu_instructions"] * 4.0 / (1024 * 2.4e9) or None,
                  "quotient_kernels": {k: {kk: v.get(kk) for kk in ("hbm_bytes_per_proof", "algorithmic_bytes_per_proof", "traffic_over_algorithmic",
                                                                       "busy_us_per_proof")}
                                       for k, v in d["kernels"].items() if k.startswith("k_quot")}}
    return out


def kernel_source_hash():
    """SHA-256 over the CODE of the kernel sources: comments and white space are taken out first, so that rewording a
    comment does not orphan the counters (string literals in these headers hold no comment markers; '#error' texts count)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        src = open(os.path.join(ROOT, "city-rollup_amd", "csrc", f), "r").read()
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
        src = re.sub(r"//[^\n]*", " ", src)
        h.update(" ".join(src.split()).encode())
    return h.hexdigest()[:16]


def stored_pmc():
    """PMC counters of the bench kernels from a separate rocprofv3 --pmc run of this command (tools/profile_bench.sh), or
    ({}, reason) when there are none for the kernel sources as they are now."""
    try:
        d = json.load(open(PMC_JSON))
    except (OSError, ValueError):
        return {}, "no stored PMC summary"
    if d.get("kernel_source_hash") != kernel_source_hash():
        return {}, "stored PMC summary is for other kernel sources (hash %s, now %s): refused" % (d.get("kernel_source_hash"), kernel_source_hash())
    return d.get("kernels", {}), None


def splitmix64_felts(seed, n):
    P = np.uint64(0xFFFFFFFF00000001)
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + i + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z % P


def host_cpu_share():
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU quota when there is one (a GPU box
    shows all 64+ logical CPUs of its host in the mask but schedules the container on a share of them: threads beyond the share
    only take turns). Returns (usable, affinity, quota or None)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()          # cgroup v2
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:                                                             # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    usable = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return usable, aff, quota


def cpu_baseline(cols_host, log_n, cap_h):
    """The oracle (a port, not the Rust reference — which cannot be built offline) timed on this
    box's host cores on the same workload: ONE full step (135 NTTs of 2^20 + the Merkle cap)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    usable, affinity, quota = host_cpu_share()
    cores = min(usable, 64)
    L = O.lib()
    L.or_set_threads(cores)
    L.or_set_fast_poseidon(1)   # multiplier-free MDS planes, branch-free field ops: the faster form of the port
    k, n = cols_host.shape
    work = cols_host.copy()
    import ctypes
    from concurrent.futures import ThreadPoolExecutor

    def one(p):
        L.or_ntt(O.ptr(work[p]), log_n)  # ctypes releases the GIL
        L.or_bit_reverse(O.ptr(work[p]), log_n)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(one, range(k)))
    t1 = time.perf_counter()
    cap = np.zeros((1 << cap_h, 4), np.uint64)
    L.or_merkle_tree_cols(O.ptr(work), n, k, n, cap_h, None, O.ptr(cap))
    t2 = time.perf_counter()
    # ... and with eight states per AVX-512 permutation (oracle/poseidon_simd.c) where the host has it: the
    # same bytes (tests/test_oracle_simd.py; the cap is compared here too), what a CPU prover that uses its vector units reaches
    L.or_simd_available.restype = ctypes.c_int
    simd = bool(L.or_simd_available())
    simd_cap_s = simd_rate_one = None
    if simd:
        L.or_set_simd_poseidon(1)
        cap2 = np.zeros_like(cap)
        ts = time.perf_counter()
        L.or_merkle_tree_cols(O.ptr(work), n, k, n, cap_h, None, O.ptr(cap2))
        simd_cap_s = time.perf_counter() - ts
        assert (cap2 == cap).all(), "AVX-512 Merkle cap differs from the scalar port's"
        st8 = splitmix64_felts(12, 12 * 80000).reshape(-1, 12).copy()
        L.or_set_threads(1)
        ts = time.perf_counter()
        O.permute_many(st8)
        simd_rate_one = len(st8) / (time.perf_counter() - ts)
        L.or_set_simd_poseidon(0)
    L.or_set_threads(1)
    # single-thread permutation rate of the port, on a bounded sample: the honest scale of this baseline
    st = splitmix64_felts(11, 12 * 20000).reshape(-1, 12).copy()
    L.or_set_threads(1)
    t3 = time.perf_counter()
    O.permute_many(st)
    t4 = time.perf_counter()
    L.or_set_fast_poseidon(0)
    perms = n * ((k + 7) // 8) + (n - 16)
    all_rate, one_rate = perms / (t2 - t1), len(st) / (t4 - t3)
    step_s = (t1 - t0) + (simd_cap_s if simd else (t2 - t1))
    return {
        "value": step_s * 1e3 / k, "unit": "ms/NTT", "cores": cores, "kind": "port-simd" if simd else "port",
        "host": {"threads_used": cores, "cpus_in_affinity_mask": affinity, "cgroup_cpu_quota": quota, "nproc": os.cpu_count(), "avx512": simd},
        "poseidon_perms_per_s_all_cores": all_rate, "poseidon_perms_per_s_one_thread": one_rate,
        "poseidon_speedup_over_one_thread": all_rate / one_rate, "poseidon_parallel_efficiency": all_rate / one_rate / cores,
        "scalar_port": {"value": (t2 - t0) * 1e3 / k, "merkle_cap_s": t2 - t1},
        "simd": None if not simd else {
            "poseidon_perms_per_s_all_cores": perms / simd_cap_s, "poseidon_perms_per_s_one_thread": simd_rate_one, "merkle_cap_s": simd_cap_s,
            "speedup_over_the_scalar_port": (t2 - t1) / simd_cap_s,
            "note": "eight states per permutation in AVX-512 lanes (oracle/poseidon_simd.c), same round structure and bytes as the scalar port"},
        "note": "a C restatement (multiplier-free MDS on 32-bit planes, branch-free field ops, textbook round structure, OpenMP over "
                "columns / leaves; `value` uses the AVX-512 permutation over eight states where the host has it, `scalar_port` is the "
                "same step without it; the poseidon_* figures at this level are the scalar port's), NOT plonky2's own prover: no "
                "speed-up over the reference may be read off this number (the reference cannot be built here: no Rust toolchain)",
        "sample": f"one full step on the host: {k} x 2^{log_n} NTT (+bit-reverse) = {(t1 - t0):.2f} s, "
                  f"Poseidon Merkle cap over 2^{log_n} x {k} = {(t2 - t1):.2f} s scalar"
                  + (f", {simd_cap_s:.2f} s with AVX-512" if simd else "") + f"; C oracle, {cores} threads",
        "ntt_ms": (t1 - t0) * 1e3 / k, "merkle_cap_s": t2 - t1,
    }, work, cap


def cpu_port_proof(prover, cores):
    """M1's CPU figure: ONE whole proof of the qbench workload by the C oracle ("port": a plain
    restatement, OpenMP over polynomials / leaves / quotient points / PoW candidates — not the optimised Rust prover) on
    the host cores, and a live parity check: the GPU's bytes for the same job must equal the oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ctypes
    import oracle_lib as O
    import bench_prove
    import cityprover as cp
    c = bench_prove.cases_for(prover, 4, bench_prove.POSEIDON_FRACTION)[0]
    sh = cp.standard_recursion_shape(num_constants=c["num_constants"], num_public_inputs=len(c["public_inputs"]))
    osh = O.standard_shape(num_constants=c["num_constants"])
    og = O.make_gates(c["gate_list"], c["num_selectors"], c["k_is"])
    digest = [0, 1, 2, 3]
    circ = cp.Circuit(prover, sh, digest, c["cs_values"])
    cp.set_gates(circ, c["gate_list"], c["num_selectors"])
    got = cp.prove(circ, c["wires"], c["public_inputs"])
    circ.close()
    O.lib().or_set_threads(cores)
    O.lib().or_set_fast_poseidon(1)
    O.lib().or_simd_available.restype = ctypes.c_int
    simd = bool(O.lib().or_simd_available())
    scalar_sec = None
    if simd:   # one scalar proof for the record, then everything below with the AVX-512 permutation (same bytes: asserted)
        ts = time.perf_counter()
        O.prove_full(osh, og, digest, c["public_inputs"], c["cs_values"], c["wires"])
        scalar_sec = time.perf_counter() - ts
        O.lib().or_set_simd_poseidon(1)
    t0 = time.perf_counter()
    O.commit_batch(c["cs_values"], 3, 4, want=("cap",))   # circuit data: built once per circuit upstream, not per proof
    t1 = time.perf_counter()
    want, _ = O.prove_full(osh, og, digest, c["public_inputs"], c["cs_values"], c["wires"])
    t2 = time.perf_counter()
    O.lib().or_set_threads(1)
    assert got == want, "GPU proof bytes != CPU oracle proof bytes"
    sec = (t2 - t1) - (t1 - t0)
    # The reference scales its CPU path by running many proofs side by side (one worker process per core group,
    # city_rollup_core_worker/src/lib.rs:131-145), not by spreading one proof over every core: N independent proofs on N threads,
    # one thread each (the oracle's OpenMP regions run with one thread; ctypes releases the GIL), every one compared with the GPU's bytes
    from concurrent.futures import ThreadPoolExecutor

    def one_proof(_):
        b, _dbg = O.prove_full(osh, og, digest, c["public_inputs"], c["cs_values"], c["wires"])
        return b
    t3 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        many = list(ex.map(one_proof, range(cores)))
    t4 = time.perf_counter()
    O.lib().or_set_fast_poseidon(0)
    O.lib().or_set_simd_poseidon(0)
    assert all(b == got for b in many), "a side-by-side CPU proof differs from the GPU's bytes"
    return {"proofs_per_s": cores / (t4 - t3), "seconds_per_proof": sec, "cores": cores, "kind": "port-simd" if simd else "port",
            "seconds_per_proof_scalar_port_all_threads_incl_constants_commitment": scalar_sec,
            "proofs_per_s_one_proof_on_all_threads": 1.0 / sec,
            "proofs_per_s_independent_proofs_one_thread_each": cores / (t4 - t3), "seconds_per_proof_on_one_thread": t4 - t3,
            "throughput_note": "proofs_per_s = %d independent proofs on %d threads, one thread each (how the reference's worker processes scale); "
                               "the latency figure is one proof spread over all threads (constants/sigmas commitment included in the side-by-side "
                               "runs: +%.2f s of %.2f s each)" % (cores, cores, t1 - t0, t4 - t3),
            "sample": "1 proof of the qbench workload (n = 2^12, city-common gate set); constants/sigmas commitment "
                      "(%.2f s) subtracted: it belongs to circuit build" % (t1 - t0),
            "parity": "GPU proof bytes == oracle proof bytes (%d B)" % len(got)}


def native_qbench(device, rank, pack):
    """Second half of BASELINE.json's metric, "block proofs/sec (qbench)", from the native harness (tools/cityprover_qbench:
    the reference's q-bench loop on a worker pool above the C ABI): the example dump replayed with 64 (and 32) blocks in flight,
    one block alone, and the raw proofs/s mode — every proof compared with the oracle's bytes recorded in the pack."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "tools", "cityprover_qbench")
    srcs = [exe + ".cpp"] + [os.path.join(ROOT, "tools", "qbench", h) for h in ("jobs.h", "pack.h", "redis.h")] + [os.path.join(ROOT, "include", "cityprover.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(f) for f in srcs):
        # normally built by __graft_entry__.build(); a box that only received the sources builds it here (plain g++)
        tmp_exe = "%s.%d.tmp" % (exe, os.getpid())
        subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tools"), srcs[0],
                        "-L" + os.path.join(ROOT, "city-rollup_amd"), "-lcityprover_hip", "-Wl,-rpath,$ORIGIN/../city-rollup_amd",
                        "-lpthread", "-o", tmp_exe], check=True)
        os.replace(tmp_exe, exe)
    dump = os.path.join(ROOT, "tests", "golden", "qbench_example.bin")
    with tempfile.TemporaryDirectory(prefix="cpq%d_" % rank) as tmp:

        def run(args):
            r = subprocess.run([exe] + args + ["--pack", pack, "--devices", str(device)], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("cityprover_qbench failed: " + r.stderr[-500:])
            return json.loads(r.stdout.strip().splitlines()[-1])
        out_json = os.path.join(tmp, "out.json")
        # BASELINE.json configs[3] is a batch of 64 independent blocks: 64 in flight on this GPU. A worker takes up to 128 ready jobs
        # per launch: with the cap at 32 three workers cut a long ready queue into small launches (26.6 blocks/s, mean batch 13,
        # against 33.5 and 45: profiles/r03_qbench_batch_matrix.jsonl) — and a block is not finished LATER for it (1.18 -> 0.97 s at 32 in flight)
        many = run(["-i", dump, "-o", out_json, "-n", "256", "--blocks-in-flight", "64", "--contexts", "3", "--batch", "128", "--check-plan"])
        per_job = json.load(open(out_json))
        many32 = run(["-i", dump, "-n", "128", "--blocks-in-flight", "32", "--contexts", "3", "--batch", "128"])
        one = run(["-i", dump, "-n", "8", "--contexts", "3", "--batch", "128"])   # eight blocks one after the other: mean latency
        serial = run(["-i", dump, "--contexts", "1", "--batch", "1"])
        thr = run(["--mode", "throughput", "--contexts", "3", "--batch", "64", "--iters", "8"])
        # ... and with the three SHA-256 STARKs of a block proved on the device too (tools/qbench/stark_stage.h: a synthetic AIR of
        # the reference's shape, 418 + 912 columns, 2^10 rows, before each sighash job)
        starks = run(["-i", dump, "-n", "128", "--blocks-in-flight", "64", "--contexts", "3", "--batch", "128", "--stark-log-rows", "10"])
        one_starks = run(["-i", dump, "-n", "8", "--contexts", "3", "--batch", "128", "--stark-log-rows", "10"])
        # the reference's loops unchanged (one job per pop, one proof per call) as 192 threads sharing one context through cp_batcher
        # (rank 0 only: a side measurement, and 192 threads per rank would be fifteen hundred on an 8-GPU node)
        callers = run(["-i", dump, "-n", "128", "--blocks-in-flight", "64", "--contexts", "1", "--lanes", "4", "--callers", "192",
                       "--batch", "64", "--linger-us", "300"]) if rank == 0 else None
    return {"blocks_per_s": many["blocks_per_s"], "proofs_per_s": many["proofs_per_s"], "blocks_in_flight": many["blocks_in_flight"],
            "jobs_per_block": many["jobs_per_block"], "proofs_per_block": many["proofs_per_block"],
            "proofs_byte_checked": many["proofs_byte_checked"], "job_records_written": len(per_job),
            "distinct_proofs_per_block": many["distinct_proofs"], "distinct_circuits": many["circuits"],
            "distinct_proofs_equal_to_oracle_bytes": many["distinct_proofs_equal_to_recorded_bytes"],
            "distinct_proofs_cp_verified": many["distinct_proofs_cp_verified"], "mean_batch": many["mean_batch"],
            "launches": many["launches"], "mean_block_latency_ms": many["mean_block_latency_ms"],
            "with_32_blocks_in_flight": {"blocks_per_s": many32["blocks_per_s"], "proofs_per_s": many32["proofs_per_s"],
                                         "mean_batch": many32["mean_batch"], "mean_block_latency_ms": many32["mean_block_latency_ms"]},
            "one_block_mean_batch": one["mean_batch"],
            "one_block_latency_ms": one["mean_block_latency_ms"],
            "reference_loop_block_ms": serial["mean_block_latency_ms"],
            "throughput_mode_proofs_per_s": thr["proofs_per_s"], "contexts_per_gpu": 3, "max_batch": 128,
            "with_the_three_starks_of_a_block": {
                "blocks_per_s": starks["blocks_per_s"], "plonky2_proofs_per_s": starks["proofs_per_s"], "stark_proofs": starks["stark_proofs"],
                "stark_log_rows": starks["stark_log_rows"], "stark_proof_bytes": starks["stark_proof_bytes_mean"], "blocks": starks["blocks"],
                "one_block_latency_ms": one_starks["mean_block_latency_ms"],
                "note": "-n 128 --blocks-in-flight 64 --stark-log-rows 10: every GenerateSigHashIntrospectionProof job first runs cp_stark_prove on a "
                        "synthetic AIR of the reference's shape (418 + 912 columns, >= 10^4-op constraint program, extended columns filled on the "
                        "device, 84 queries): 3 STARK proofs per block beside its 64 plonky2 proofs. The AIR and its trace are stand-ins (the real "
                        "one lives in an absent crate); the row count of the reference's STARK (smartgadget.rs:310-312: 2^ceil(log2(cycle x rounds))) "
                        "for a sighash preimage of a few hundred bytes is 2^10 - 2^11"},
            "one_job_per_call_threads": None if callers is None else {
                "blocks_per_s": callers["blocks_per_s"], "proofs_per_s": callers["proofs_per_s"], "threads": 192, "lanes": 4, "max_batch": 64,
                "linger_us": 300, "proofs_byte_checked": callers["proofs_byte_checked"],
                "note": "rank 0's GPU only; 64 blocks in flight, --callers 192 --lanes 4 --batch 64: the DAG drained by one-job-per-call threads merged by "
                        "cp_batcher (include/cityprover.h) instead of a batching worker"},
            "harness": "tools/cityprover_qbench -i tests/golden/qbench_example.bin (the reference's own q-bench dump) -n 256 "
                       "--blocks-in-flight 64 --contexts 3 --batch 128 (BASELINE.json configs[3]: a batch of 64 independent blocks; "
                       "with_32_blocks_in_flight = -n 128 --blocks-in-flight 32); one_block = the same dump alone; reference_loop = one "
                       "context, one job at a time (the reference's single-threaded loop)",
            "workload": "SURVEY.md section 8(d) M1: the example block's 46 jobs = 64 DISTINCT plonky2 proofs per block — one synthetic "
                        "shape-equivalent circuit per (job type, stage) (26 circuits), one witness per job (seed = job index); n = 2^12, 135 "
                        "wires / 80 routed, 28 queries, 16-bit PoW, the 14-gate city-common set, rows ~60 % Poseidon. Before the clock starts "
                        "8 of the 64 must equal the CPU oracle's bytes and the other 56 pass cp_verify; every proof of the timed run is "
                        "compared with the bytes that passed. Ready jobs of any type share launches (one shape, one gate set); wires in "
                        "page-locked host memory (PCIe-inclusive), proofs end in host memory; witness generation and the 3 Groth16 "
                        "proofs of a block are not in this number (the reference's q-bench runs with GROTH16_DISABLED_DEV_MODE too); the 3 SHA-256 "
                        "STARKs are in with_the_three_starks_of_a_block, not in blocks_per_s"}


def power_and_clock(prover, cp, data_ptr, cap_ptr, k, log_n, seconds=1.5):
    """Board power and shader clock under the two loads of the step (is the 2.0-2.1 GHz the leaf hash runs at a
    power cap?): `rocm-smi --showpower --showclocks --json` sampled from a side thread while the leaf hash, then the NTT passes,
    run back to back for `seconds`. Returns {load: {power_W: [...], sclk_MHz: [...]}} or a note when rocm-smi is not usable."""
    import re
    import shutil
    import subprocess
    import threading
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return {"note": "rocm-smi not found"}
    n = 1 << log_n

    def sample():
        try:
            r = subprocess.run([exe, "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
            d = json.loads(r.stdout)
        except Exception as e:   # noqa: BLE001
            return {"error": str(e)[:200]}
        card = d.get("card0") or (list(d.values())[0] if d else {})
        out = {}
        for key, val in card.items():
            m = re.search(r"[-+]?[0-9]*[.]?[0-9]+", str(val))
            if not m:
                continue
            if "ower" in key and "W" in key:
                out.setdefault("power_W", float(m.group(0)))
            if key.lower().startswith("sclk"):
                out.setdefault("sclk_MHz", float(m.group(0)))
        return out

    res = {}
    for name, fn in (("idle", None), ("leaf_hash", lambda: prover.merkle_cols_dev(data_ptr, n, k, CAP_H, cap_ptr)),
                     ("ntt_passes", lambda: prover.ntt_dev(data_ptr, log_n, k, n, cp.NTT_BITREV_OUT))):
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                samples.append(sample())
        th = threading.Thread(target=poll)
        th.start()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < (seconds if fn else 0.5):
            if fn:
                for _ in range(8):
                    fn()
                prover.sync()
            else:
                time.sleep(0.05)
        stop.set()
        th.join()
        res[name] = {"power_W": [x["power_W"] for x in samples if "power_W" in x], "sclk_MHz": [x["sclk_MHz"] for x in samples if "sclk_MHz" in x],
                     "samples": len(samples)}
        if samples and "error" in samples[0] and not res[name]["power_W"]:
            res[name]["error"] = samples[0]["error"]
    return res


def all_ranks_or_exit(D, dist, fn, what, cleanup=None):
    """fn() on every rank; a rank whose fn raises must not leave the others waiting in the next collective: the
    failure flag is reduced first, and when any rank failed EVERY rank exits non-zero (the --gpus launcher then reports it)."""
    res, err = None, None
    try:
        res = fn()
    except Exception as e:   # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    failed = D.sum_over_ranks(dist, 0.0 if err is None else 1.0)
    if cleanup:
        cleanup()
    if failed:
        sys.exit("bench.py: %s failed on %d rank(s)%s" % (what, int(failed), "" if err is None else " — this rank: " + err))
    return res


_JSON_OUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout. Libraries below us write there too (gloo prints "[Gloo] Rank 0 is connected to
    ..." from C++ when a process group forms), so a rank keeps a private handle on the real stdout for its JSON line and points
    file descriptor 1 at stderr for everybody else, itself included."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _JSON_OUT


def emit(obj):
    out = claim_stdout()
    out.write(json.dumps(obj) + "\n")
    out.flush()


def stub_main(args, D):
    """CITYPROVER_BENCH_STUB=1: everything of a multi-rank run EXCEPT the GPU section — rank environment, rendezvous, the
    barrier / max / sum / broadcast of the control plane, the all-ranks-or-exit rule, rank-0-only output — so that the
    `--gpus N` path can be tested on a box without a GPU (tests/test_dist_gloo.py). CITYPROVER_BENCH_STUB_FAIL=<rank> makes that
    rank's side measurement raise; CITYPROVER_BENCH_STUB_EXIT=<rank> makes it exit 3 at the very end."""
    rank, local_rank, world = D.env_rank()
    dist = D.init("gloo")
    units = D.shard_units(world, rank, world)
    D.barrier(dist)
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))   # the slowest rank sets the time
    D.barrier(dist)
    elapsed = D.max_over_ranks(dist, time.perf_counter() - t0)
    token = D.broadcast_str(dist, "pack-of-rank-0" if rank == 0 else None)

    def side():
        if os.environ.get("CITYPROVER_BENCH_STUB_FAIL") == str(rank):
            raise RuntimeError("stub failure on rank %d" % rank)
        return {"units": len(units)}
    mine = all_ranks_or_exit(D, dist, side, "the stub side measurement")
    total_units = D.sum_over_ranks(dist, mine["units"])
    if rank == 0:
        emit({"metric": "stub", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "elapsed_s": elapsed,
              "units": total_units, "broadcast": token, "local_rank": local_rank})
    if dist is not None:
        dist.destroy_process_group()
    if os.environ.get("CITYPROVER_BENCH_STUB_EXIT") == str(rank):
        sys.exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-qbench", action="store_true", help="skip the side measurements (q-bench, NTT variants, Groth16 kernels)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: this process starts N ranks of itself (one per GPU) before anything
    # here touches the GPU, passes rank 0's JSON line through and fails when any rank fails. Under torchrun
    # (WORLD_SIZE set) the flag must agree with the launcher.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        codes = [pr.wait() for pr in procs]
        if any(codes):
            sys.exit("bench.py: rank exit codes %s" % codes)
        return
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))

    claim_stdout()
    from cityprover import dist as D
    if os.environ.get("CITYPROVER_BENCH_STUB"):
        return stub_main(args, D)
    import cityprover as cp

    rank, local_rank, world = D.env_rank()
    dist = D.init("gloo")  # control plane only: barrier + max-reduce of the timing, no data path

    # one rank per GPU; with more ranks than GPUs (a rehearsal of the N > 1 path on a one-GPU box) ranks share devices
    n_dev = cp.load_library().cp_device_count()
    device = local_rank % n_dev if n_dev > 0 else local_rank
    prover = cp.Prover(device)
    k, log_n, n = args.cols, args.log_n, 1 << args.log_n
    # weak scaling: rank r commits its own trace (unit r of `world` independent units)
    host = splitmix64_felts(D.unit_seed(SEED, D.shard_units(world, rank, world)[0]), k * n).reshape(k, n)
    data = prover.to_device(host)
    cap = prover.alloc(4 << CAP_H)

    def sync():
        prover.sync()
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except ImportError:
            pass

    def barrier():
        D.barrier(dist)

    def step():
        prover.ntt_dev(data.ptr, log_n, k, n, cp.NTT_BITREV_OUT)
        prover.merkle_cols_dev(data.ptr, n, k, CAP_H, cap.ptr)

    # correctness gate on this rank's first step (rank 0 also times the CPU baseline with it)
    base = None
    if rank == 0 and not args.no_cpu_baseline:
        base, want_cols, want_cap = cpu_baseline(host, log_n, CAP_H)
        step()
        sync()
        got_cap = cap.download().reshape(-1, 4)
        assert (got_cap == want_cap).all(), "GPU Merkle cap != CPU oracle"
        got_col = data.download(n, offset=(k - 1) * n)
        assert (got_col == want_cols[k - 1]).all(), "GPU NTT != CPU oracle"
        del want_cols
        data.upload(host)
    del host

    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    prover.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    barrier()
    t1 = time.perf_counter()
    prof = prover.profile_end()
    elapsed = D.max_over_ranks(dist, t1 - t0)

    # M2's other transforms: inverse NTT and the rate-8 coset LDE 2^20 -> 2^23 (shift 7), batch 16,
    # device-resident, HIP events around the whole call. Side measurement (not part of `value`).
    variants = None
    if (k, log_n) == (COLS, LOG_N) and not args.no_qbench:
        vb = 16
        e0, e1 = prover.event(), prover.event()
        src = prover.alloc(vb * n)
        prover.lib.cp_d2d(prover.ctx, src.ptr, data.ptr, vb * n * 8)
        dst = prover.alloc(vb * (n << 3))

        def timed(fn, reps=5):
            fn()
            prover.record(e0)
            for _ in range(reps):
                fn()
            prover.record(e1)
            return prover.elapsed_ms(e0, e1) / reps

        t_inv = timed(lambda: prover.ntt_dev(src.ptr, log_n, vb, n, cp.NTT_INVERSE))
        t_lde = timed(lambda: prover.lde_dev(src.ptr, log_n, 3, vb, dst.ptr, 7, cp.NTT_BITREV_OUT))
        variants = {"batch": vb, "intt_ms_per_poly": t_inv / vb, "lde_rate8_ms_per_poly": t_lde / vb,
                    "lde_GBs": vb * 8.0 * n * (1 + 8) / (t_lde * 1e-3) / 1e9,
                    "lde_algorithmic_bytes_per_poly": 8.0 * n * (1 + 8)}
        src.free()
        dst.free()
    pw = None
    if rank == 0 and (k, log_n) == (COLS, LOG_N) and not args.no_qbench:
        pw = power_and_clock(prover, cp, data.ptr, cap.ptr, k, log_n)
    data.free()
    data = None

    # "block proofs/sec (qbench)": every rank runs the native harness on its own GPU (jobs shard by block, no collective)
    qb = None
    if not args.no_qbench:
        # the synthetic circuit pack (26 circuits, 64 witnesses, the oracle's proofs of 8 of them) is written once, by rank 0,
        # into a fresh private directory whose name the other ranks learn through the control plane
        import shutil
        import tempfile
        pack = None
        if rank == 0:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import make_circuit_pack
            pack = tempfile.mkdtemp(prefix="cityprover_pack_")
            make_circuit_pack.make_pack(pack, db=12, n_checked=8)
        pack = D.broadcast_str(dist, pack)
        mine = all_ranks_or_exit(D, dist, lambda: native_qbench(device, rank, pack), "the q-bench harness",
                                 cleanup=(lambda: shutil.rmtree(pack, ignore_errors=True)) if rank == 0 else None)
        qb = dict(mine)
        for key in ("blocks_per_s", "proofs_per_s", "throughput_mode_proofs_per_s"):
            qb[key] = D.sum_over_ranks(dist, mine[key])
        if rank == 0 and not args.no_cpu_baseline:
            qb["cpu_baseline"] = cpu_port_proof(prover, min(host_cpu_share()[0], 64))

    # Groth16-wrap kernels, side measurement on rank 0: G1 MSM and F_r NTT at 2^20, both with a
    # correctness check inside (closed form resp. inverse round trip) — tools/bench_msm.py, tools/bench_fr_ntt.py
    g16 = None
    if rank == 0 and not args.no_qbench:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_msm
        import bench_fr_ntt
        m18 = bench_msm.run(prover, 18, reps=2)          # checked against (sum k_i (a i + b)) * G
        m20 = bench_msm.run(prover, 20, reps=2)
        g2 = bench_msm.run_g2(prover, 18, reps=2)
        g2_20 = bench_msm.run_g2(prover, 20, reps=2)
        f20 = bench_fr_ntt.run(prover, 20, reps=3)
        import bench_groth16
        pr20 = bench_groth16.run(prover, 20, reps=2)      # whole proof assembly: five MSMs + quotient + host part
        # the size of a wrap: a gnark circuit that verifies a plonky2 proof is 2^24-2^25 constraints - timed AND checked
        # against the trapdoor there too (CITYPROVER_BENCH_GROTH16_LOG overrides; 0 skips)
        big = int(os.environ.get("CITYPROVER_BENCH_GROTH16_LOG", "24"))
        pr_big = bench_groth16.run(prover, big, reps=1) if big > 20 else None
        g16 = {"msm_g1_2^18_ms": m18["ms"], "msm_g1_2^18_checked": m18["checked"], "msm_g1_2^20_ms": m20["ms"],
               "msm_g1_2^20_Mpoints_per_s": m20["Mpoints_per_s"], "msm_g2_2^18_ms": g2["ms"],
               "msm_g2_2^18_checked": g2["checked"], "msm_g2_2^20_ms": g2_20["ms"], "fr_ntt_2^20_forward_ms": f20["forward_ms"],
               "fr_ntt_2^20_inverse_ms": f20["inverse_ms"], "groth16_quotient_2^20_ms": f20["groth16_quotient_ms"],
               "groth16_prove_2^20_ms": pr20["prove_ms"], "groth16_prove_2^20_checked": pr20["checked"]}
        if pr_big:
            g16["groth16_prove_2^%d_ms" % big] = pr_big["prove_ms"]
            g16["groth16_prove_2^%d_checked" % big] = pr_big["checked"]
            # what fixed-base precomputation could remove (one bucket set for all windows shrinks the reduction tail, not the bucket
            # sums): the tail's share of the kernels of the calling context's MSM chain at this size, and at 2^20
            g16["reduction_tail_share"] = {"2^20": pr20.get("reduction_tail_share_of_main_chain"), "2^%d" % big: pr_big.get("reduction_tail_share_of_main_chain")}
            g16["groth16_prove_2^%d_main_chain_kernels_ms" % big] = pr_big.get("main_chain_kernels_ms")
        # the MSMs against the roof that bounds them: VALU issue of the quarter-rate multiply-add v_mad_u64_u32 (a Montgomery product
        # of the 381-bit field is 392 of them + ~95 other instructions). Lane-operations per MSM from the counter pass of
        # tools/profile_r04.sh (profiles/r04_pmc_msm.json), the durations of THIS run.
        try:
            pm = json.load(open(_latest_profile("pmc_msm.json")))
            roof_msm = {"bound": "valu-issue (v_mad_u64_u32, quarter rate)", "unit": "T lane-ops/s",
                        "peak": VALU_PEAK_TLOPS / 2.0,
                        "peak_note": "1024 SIMD-32 x 64 lanes per 4 cycles at 2.4 GHz = the nominal issue rate of a quarter-rate wave64 instruction; "
                                     "tools/ubench_valu.hip measures 31.5 T/s for a pure v_mad_u64_u32 stream (profiles/r01_ubench_valu.txt)",
                        "measured_mad_stream_TLOPS": 31.5, "counter_source": os.path.relpath(_latest_profile("pmc_msm.json"), ROOT)}
            for grp, ms_ in (("G1", m20["ms"]), ("G2", g2_20["ms"])):
                lo = pm["groups"][grp]["valu_lane_ops_per_msm"]
                ach = lo / (ms_ * 1e-3) / 1e12
                roof_msm[grp] = {"points": 1 << 20, "ms": ms_, "valu_lane_ops_per_point": pm["groups"][grp]["valu_lane_ops_per_point"], "achieved": ach,
                                 "frac": ach / (VALU_PEAK_TLOPS / 2.0), "frac_of_measured_mad_stream": ach / 31.5}
            g16["msm_roofline"] = roof_msm
        except (OSError, KeyError, ValueError) as e:
            g16["msm_roofline"] = "no counter file for the MSM kernels: %s" % e

    # A13's polynomial-commitment half through the generic seams (cp_batch_commit_dev / cp_fri_prove) at the SHA-256 STARK's
    # shapes, side measurement on rank 0 with cp_fri_verify as its check — tools/bench_stark_fri.py
    stark = None
    if rank == 0 and not args.no_qbench:
        import bench_stark_fri
        stark = {"2^10": bench_stark_fri.run(prover, 10, reps=3), "2^14": bench_stark_fri.run(prover, 14, reps=3),
                 "2^16": bench_stark_fri.run(prover, 16, reps=2)}

    # A13's own two steps on the generic device machinery (cp_air_quotient_commit / cp_stark_prove: a recorded constraint program
    # of >= 10^4 ops at the SHA-256 STARK's width evaluated on the quotient coset straight from the committed traces; the whole
    # prover with its extended columns filled on the device) - tools/bench_stark_air.py; tests/test_gpu_air.py holds the bytes
    stark_air = stark_sha = None
    if rank == 0 and not args.no_qbench:
        import bench_stark_air
        stark_air = {"2^10": bench_stark_air.run(prover, 10, reps=3), "2^14": bench_stark_air.run(prover, 14, reps=3),
                     "2^16": bench_stark_air.run(prover, 16, reps=2)}
        # ... and a SATISFIED AIR through the same prover, verified in full: SHA-256 of a 2^k / 64 - 1 block message, digest == hashlib
        stark_sha = {"2^10": bench_stark_air.run_sha256(prover, 10, reps=3), "2^14": bench_stark_air.run_sha256(prover, 14, reps=3)}

    if rank == 0:
        ms_step = elapsed * 1e3 / args.steps
        ntts = k * world
        value = ms_step / ntts

        def kern(name):
            d = prof.get(name)
            return (d["total_ms"] / d["launches"], d["launches"]) if d else (None, 0)

        leaf_ms, _ = kern("leaf_hash_cols")
        cols_ms, cols_l = kern("ntt16_cols")
        rows_ms, rows_l = kern("ntt16_rows")
        lvl_ms = sum(prof.get(nm, {"total_ms": 0.0})["total_ms"] for nm in ("merkle_level", "merkle_level_fused", "merkle_level_coop", "merkle_levels_coop"))
        # Dominant kernel by time: the Poseidon leaf hash (75 % of the step). It moves exactly its algorithmic bytes
        # (8*R*k read + 32*R written) but is bound by integer-VALU issue, not by HBM: ~15 K VALU
        # instructions per permutation. `roofline` is therefore priced in lane-operations against the VALU issue peak;
        # the HBM figures stay inside it as `hbm`. Instruction counts and HBM traffic come from separate rocprofv3 --pmc
        # passes of this command (tools/profile_bench.sh -> profiles/r02_pmc_bench.json) and are used only when that file
        # was collected with the kernel sources as they are now; the durations are this run's HIP events.
        default_shape = (k, log_n) == (COLS, LOG_N)
        pmc, pmc_refused = stored_pmc() if default_shape else ({}, "not the default workload shape")
        # round 4: the leaf-hash workgroups also compute the first level of their 256-leaf subtrees (csrc/merkle.h fused_levels;
        # CITYPROVER_MERKLE_FUSE = 0..3, default 1: deeper measured slower, profiles/r04_merkle_fuse_matrix.jsonl): those node
        # permutations and digests are work of the SAME launch and are counted with it
        fuse = max(0, min(3, int(os.environ.get("CITYPROVER_MERKLE_FUSE", "1"))))
        coop_max = int(os.environ.get("CITYPROVER_COOP_MAX", "16384"))
        while fuse > 0 and ((n >> fuse) < coop_max or (n >> (fuse + 0)) <= (1 << CAP_H) or n % 256):
            fuse -= 1
        fused_nodes = sum(n >> l for l in range(1, fuse + 1))
        leaf_bytes = 8.0 * n * k + 32.0 * (n + fused_nodes)
        perms = n * ((k + 7) // 8) + fused_nodes
        pl = pmc.get("leaf_hash_cols", {})
        hbm = {"achieved_GBs": leaf_bytes / (leaf_ms * 1e-3) / 1e9, "peak_GBs": HBM_PEAK_GBS,
               "frac": leaf_bytes / (leaf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": leaf_bytes,
               "traffic_bytes": pl.get("hbm_bytes")}
        if pl.get("SQ_INSTS_VALU"):
            lane_ops = pl["SQ_INSTS_VALU"] * 64.0
            clk = pl.get("clock_GHz_est")
            roof = {"kernel": "leaf_hash_cols", "bound": "valu-issue", "unit": "T lane-ops/s",
                    "achieved": lane_ops / (leaf_ms * 1e-3) / 1e12, "peak": VALU_PEAK_TLOPS,
                    "peak_note": "NOMINAL peak: 1024 SIMD-32 x one full-rate wave64 VALU instruction per 2 cycles at 2.4 GHz "
                                 "(MI355X_MICROARCH.md). Measured per instruction with eight waves per SIMD (tools/ubench_valu, "
                                 "profiles/r04_ubench_valu.txt, cycles per wave64 instruction at the nominal clock): only plain 32-bit add / sub / "
                                 "and / xor / mov and f32 multiply-add reach 2.4-3.0; everything a 64-bit modular product and the exact f64 "
                                 "limb planes are made of takes 4.2-5.0 (v_mad_u64_u32 5.0, v_fma_f64 / v_add_f64 4.3, carry chains 4.3-4.9, "
                                 "v_cndmask on an SGPR mask 4.5, shifts 4.1, three-operand integer forms 4.4-4.7): the ceiling of the MIX is about "
                                 "half this peak - see frac_of_mix_rate",
                    "frac_is_against": "the nominal full-rate peak (roofline.frac, frac_at_measured_clock); roofline.frac_of_mix_rate is "
                                       "against the measured issue rate of this instruction mix (4 cycles per instruction)",
                    "valu_instructions_per_launch": pl["SQ_INSTS_VALU"], "valu_instructions_per_permutation": lane_ops / perms,
                    "clock_GHz_under_load": clk,
                    "peak_at_measured_clock": VALU_PEAK_TLOPS * clk / 2.4 if clk else None,
                    "valu_frac_of_wave_cycles": pl.get("valu_frac_of_wave_cycles"), "wave_issue_stall_frac": pl.get("wait_inst_any_frac"),
                    "wave_memory_wait_frac": pl.get("wait_any_frac"),
                    "counter_source": os.path.relpath(PMC_JSON, ROOT) + " (kernel source hash %s)" % kernel_source_hash(),
                    "traffic": pl.get("hbm_bytes"), "hbm": hbm}
            roof["frac"] = roof["achieved"] / roof["peak"]
            if clk:
                roof["frac_at_measured_clock"] = roof["achieved"] / roof["peak_at_measured_clock"]
            # what the kernel actually issues: with the shader clock rocm-smi reports while it runs (the clock estimated under
            # the counter pass is the counter pass's), cycles a SIMD spends per wave64 VALU instruction. A SIMD-32 issues a
            # full-rate wave64 instruction in 2 cycles and a quarter-rate one (v_mad_u64_u32, v_fma_f64: this kernel's mix) in 4;
            # tools/ubench_valu.hip measures 4.3-5.0 for the VOP3 integer forms of this mix.
            sclk = sorted((pw or {}).get("leaf_hash", {}).get("sclk_MHz") or [])
            if sclk:
                mhz = sclk[len(sclk) // 2]
                roof["sclk_MHz_while_running"] = mhz
                roof["cycles_per_valu_instruction_per_simd"] = leaf_ms * 1e-3 * mhz * 1e6 / (pl["SQ_INSTS_VALU"] / 1024.0)
                # against the rate of the mix: 4 cycles per instruction = 1.0
                roof["frac_of_mix_rate"] = 4.0 / roof["cycles_per_valu_instruction_per_simd"]
        else:
            roof = {"kernel": "leaf_hash_cols", "bound": "hbm", "unit": "GB/s", "achieved": hbm["achieved_GBs"], "peak": HBM_PEAK_GBS,
                    "frac": hbm["frac"], "traffic": None, "algorithmic_bytes": leaf_bytes,
                    "note": "the kernel is integer-VALU bound (Poseidon), so this HBM fraction is low by nature; no VALU "
                            "instruction count is available for a valu-issue roofline: " + str(pmc_refused)}
        # NTT: one pass reads + writes the batch once -> 16 B per element per pass
        pass_bytes = 16.0 * n * k
        ntt_ms = (cols_ms or 0) * (cols_l / max(args.steps, 1)) + (rows_ms or 0) * (rows_l / max(args.steps, 1))
        roof_ntt = {"kernel": "ntt16_cols + ntt16_rows (k_dif_pass16)", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                    "algorithmic_bytes_per_ntt": 16.0 * n,
                    "achieved": 16.0 * n * k / (ntt_ms * 1e-3) / 1e9 if ntt_ms else None,
                    "per_pass_GBs": {"cols": pass_bytes / (cols_ms * 1e-3) / 1e9 if cols_ms else None,
                                     "rows": pass_bytes / (rows_ms * 1e-3) / 1e9 if rows_ms else None},
                    "ms_per_ntt_kernel_only": ntt_ms / k if ntt_ms else None,
                    "traffic": (pmc["ntt16_cols"]["hbm_bytes"] + pmc["ntt16_rows"]["hbm_bytes"]) / k
                    if "ntt16_cols" in pmc and "ntt16_rows" in pmc and pmc["ntt16_cols"].get("hbm_bytes") else None,
                    "valu": {nm: {kk: pmc[nm].get(kk) for kk in ("valu_lane_ops_per_s", "valu_issue_utilization", "wait_any_frac", "wait_inst_any_frac")}
                             for nm in ("ntt16_cols", "ntt16_rows") if nm in pmc},
                    "note": "two passes over HBM (8 + 12 stages): traffic = 2 x the algorithmic bytes by construction; both passes sit at "
                            "the VALU issue ceiling of their VOP3-heavy mix (~30-32 T lane-ops/s) while ~half of every wave's cycles are "
                            "memory waits hidden by occupancy"}
        if roof_ntt["achieved"]:
            roof_ntt["frac"] = roof_ntt["achieved"] / HBM_PEAK_GBS
        # The same two passes priced against VALU issue (VERDICT r3 "next" #5): wave-instructions per launch from the counter pass,
        # this run's HIP-event durations, the shader clock rocm-smi reports while the passes run. A SIMD-32 issues a quarter-rate
        # wave64 instruction (v_mad_u64_u32: the field product is made of them) every 4 cycles, a full-rate one every 2;
        # tools/ubench_valu.hip measures 4.3-5.0 cycles for the VOP3 forms this mix is made of. cycles_per_instruction near 4 =
        # the passes run at the issue rate of their instructions, whatever the HBM fraction says.
        sclk_ntt = sorted((pw or {}).get("ntt_passes", {}).get("sclk_MHz") or [])
        vi = {}
        for nm, ms_ in (("ntt16_cols", cols_ms), ("ntt16_rows", rows_ms)):
            insts = pmc.get(nm, {}).get("SQ_INSTS_VALU")
            if insts and ms_ and sclk_ntt:
                mhz = sclk_ntt[len(sclk_ntt) // 2]
                cyc = ms_ * 1e-3 * mhz * 1e6 / (insts / 1024.0)
                vi[nm] = {"valu_instructions_per_launch": insts, "valu_instructions_per_element": insts * 64.0 / (n * k),
                          "sclk_MHz_while_running": mhz, "cycles_per_valu_instruction_per_simd": cyc, "frac_of_mix_rate": 4.0 / cyc,
                          "lane_ops_per_s": insts * 64.0 / (ms_ * 1e-3), "frac_of_nominal_valu_peak": insts * 64.0 / (ms_ * 1e-3) / 1e12 / VALU_PEAK_TLOPS}
        roof_ntt["valu_issue"] = vi or None
        roof_ntt["valu_issue_note"] = ("priced against VALU issue the passes are NOT idle: frac_of_mix_rate is 4 cycles per wave64 instruction (the rate of "
                                       "the quarter-rate multiply-adds the field product is made of) over the cycles a SIMD actually spent per instruction. A "
                                       "10 + 10 split of the 20 stages moves no instruction out of the kernels; what would is a cheaper butterfly (DESIGN 4.3)")
        out = {
            "metric": "ms/NTT at 2^20 Goldilocks (commit-shaped step: 135-column 2^20-row NTT + Poseidon Merkle cap)",
            "value": value, "unit": "ms/NTT", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 (Goldilocks mod 2^64-2^32+1; the Poseidon MDS layers are evaluated exactly on 32-bit limbs held in f64)", "data": "synthetic",
            "config": {"workload": "configs[1]: 2^%d-row x %d-column trace, forward NTT per column "
                                   "(natural->bit-reversed) + Poseidon Merkle cap height %d" % (log_n, k, CAP_H),
                       "log_n": log_n, "columns": k, "cap_height": CAP_H, "sharding": "independent traces per GPU"},
            "roofline": roof, "roofline_ntt": roof_ntt, "ntt_variants": variants,
            "kernels_ms_per_step": {name: d["total_ms"] / args.steps for name, d in prof.items()},
            "poseidon_perms_per_s": (perms / (leaf_ms * 1e-3)) if leaf_ms else None,
            "merkle_levels_ms": lvl_ms / args.steps,
            "merkle_levels_fused_into_leaf_hash": {"levels": fuse, "node_permutations": fused_nodes,
                                                   "ms_at_this_run's_leaf_rate": (fused_nodes / (perms / (leaf_ms * 1e-3)) * 1e3) if leaf_ms else None,
                                                   "note": "merkle_levels_ms counts the separate level launches only; the levels the leaf-hash workgroups compute "
                                                           "themselves are inside leaf_hash_cols (their permutations are counted in poseidon_perms_per_s)"},
            "cpu_baseline": base,
            "qbench": dict(qb, roofline=qbench_roofline(qb, (perms / (leaf_ms * 1e-3)) if leaf_ms else None)) if qb else None,
            "groth16_kernels": g16,
            "stark_commit_fri": stark,
            "stark_quotient_ms": {k: v["stark_quotient_ms"] for k, v in stark_air.items()} if stark_air else None,
            "stark_prove_ms": {k: v["stark_prove_ms"] for k, v in stark_air.items()} if stark_air else None,
            "stark_air": stark_air,
            "stark_sha256": stark_sha,
            "power_and_clock": pw,
        }
        emit(out)
    cap.free()
    prover.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
