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


def splitmix64_felts(seed, n):
    P = np.uint64(0xFFFFFFFF00000001)
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + i + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z % P


def cpu_baseline(cols_host, log_n, cap_h):
    """The oracle (a port, not the Rust reference — which cannot be built offline) timed on this
    box's host cores on the same workload: ONE full step (135 NTTs of 2^20 + the Merkle cap)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    cores = min(len(os.sched_getaffinity(0)), 64)
    L = O.lib()
    L.or_set_threads(cores)
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
    L.or_set_threads(1)
    return {
        "value": (t2 - t0) * 1e3 / k, "unit": "ms/NTT", "cores": cores, "kind": "port",
        "sample": f"one full step on the host: {k} x 2^{log_n} NTT (+bit-reverse) = {(t1 - t0):.2f} s, "
                  f"Poseidon Merkle cap over 2^{log_n} x {k} = {(t2 - t1):.2f} s; C oracle, "
                  f"{cores} threads",
        "ntt_ms": (t1 - t0) * 1e3 / k, "merkle_cap_s": t2 - t1,
    }, work, cap


def cpu_port_proof(prover, cores):
    """M1's CPU figure: ONE whole proof of the qbench workload by the C oracle ("port": a plain
    restatement, OpenMP over polynomials / leaves / quotient points / PoW candidates — not the optimised Rust prover) on
    the host cores, and a live parity check: the GPU's bytes for the same job must equal the oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
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
    t0 = time.perf_counter()
    O.commit_batch(c["cs_values"], 3, 4, want=("cap",))   # circuit data: built once per circuit upstream, not per proof
    t1 = time.perf_counter()
    want, _ = O.prove_full(osh, og, digest, c["public_inputs"], c["cs_values"], c["wires"])
    t2 = time.perf_counter()
    O.lib().or_set_threads(1)
    assert got == want, "GPU proof bytes != CPU oracle proof bytes"
    sec = (t2 - t1) - (t1 - t0)
    return {"proofs_per_s": 1.0 / sec, "seconds_per_proof": sec, "cores": cores, "kind": "port",
            "sample": "1 proof of the qbench workload (n = 2^12, city-common gate set); constants/sigmas commitment "
                      "(%.2f s) subtracted: it belongs to circuit build" % (t1 - t0),
            "parity": "GPU proof bytes == oracle proof bytes (%d B)" % len(got)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-qbench", action="store_true", help="skip the proofs/s side measurement")
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

    import cityprover as cp
    from cityprover import dist as D

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

    # Second half of BASELINE.json's metric ("block proofs/sec (qbench)"), reported beside the headline:
    # whole-proof throughput of cp_prove_batch on synthetic qbench-shaped jobs (every rank proves its own
    # jobs; 64 plonky2 proofs = one example block). Not part of the timed region above.
    qb = None
    if not args.no_qbench:
        data.free()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import bench_prove
        barrier()
        r1 = bench_prove.run(prover, 32, 3)
        barrier()
        rt = bench_prove.run_threads(3, 32, 8, device=device, host_wires=True)
        import qbench_replay
        barrier()
        rp = qbench_replay.run(32, threads=3, max_batch=32, device=device)
        pps = D.sum_over_ranks(dist, rt["proofs_per_s_steady"])
        pps1 = D.sum_over_ranks(dist, r1["proofs_per_s"])
        bps = D.sum_over_ranks(dist, rp["blocks_per_s"])
        cpu_m1 = None
        if rank == 0 and not args.no_cpu_baseline:
            cpu_m1 = cpu_port_proof(prover, min(len(os.sched_getaffinity(0)), 64))
        qb = {"cpu_baseline": cpu_m1, "proofs_per_s": pps, "blocks_per_s": pps / 64.0, "proofs_per_s_single_context": pps1,
              "batch": 32, "contexts_per_gpu": 3, "proof_bytes": r1["proof_bytes"],
              "timing": "wall clock from a common start until the last of 3 contexts has finished 8 batches of 32; wire "
                        "matrices start in page-locked HOST memory (PCIe-inclusive), proofs end in host memory",
              "dag_replay": {"blocks_per_s": bps, "blocks_in_flight_per_gpu": rp["blocks"], "proofs_per_block": 64,
                             "critical_path_proofs": rp["critical_path_proofs"], "mean_batch": rp["mean_batch"],
                             "note": "tools/qbench_replay.py: the example block's job DAG (job_planner.rs:5-154) at proof "
                                     "level, ready-queue scheduler over 3 contexts; order constraints only (a parent does "
                                     "not consume its children's bytes: witness generation is outside the build)"},
              "workload": "synthetic standard_recursion_config jobs (n=2^12, 135 wires / 80 routed, 28 queries, 16-bit "
                          "PoW; the 14-gate city-common gate set of pad_circuit.rs:31-55 in 4 selector groups; rows ~60 % Poseidon, ~25 % "
                          "Arithmetic/ArithmeticExtension/MulExtension, ~10 % Reducing/RandomAccess/BaseSum/CosetInterpolation, Noop pad), "
                          "wires -> proof bytes, witness generation excluded"}
        data = None

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
        f20 = bench_fr_ntt.run(prover, 20, reps=3)
        import bench_groth16
        pr20 = bench_groth16.run(prover, 20, reps=2)      # whole proof assembly: five MSMs + quotient + host part
        g16 = {"msm_g1_2^18_ms": m18["ms"], "msm_g1_2^18_checked": m18["checked"], "msm_g1_2^20_ms": m20["ms"],
               "msm_g1_2^20_Mpoints_per_s": m20["Mpoints_per_s"], "msm_g2_2^18_ms": g2["ms"],
               "msm_g2_2^18_checked": g2["checked"], "fr_ntt_2^20_forward_ms": f20["forward_ms"],
               "fr_ntt_2^20_inverse_ms": f20["inverse_ms"], "groth16_quotient_2^20_ms": f20["groth16_quotient_ms"],
               "groth16_prove_2^20_ms": pr20["prove_ms"]}

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
        lvl = prof.get("merkle_level", {"total_ms": 0.0, "launches": 0})
        # dominant kernel by time: the Poseidon leaf hash. Algorithmic bytes per launch:
        # 8*R*k read + 32*R digests written.
        leaf_bytes = 8.0 * n * k + 32.0 * n
        perms = n * ((k + 7) // 8)
        # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes of this same command,
        # gfx950-corrected; committed under profiles/): only valid for the default workload shape
        pmc = {}
        try:
            if (k, log_n) == (COLS, LOG_N):
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))["kernels"]
        except (OSError, ValueError, KeyError):
            pmc = {}
        roof = {"kernel": "leaf_hash_cols", "bound": "hbm",
                "achieved": leaf_bytes / (leaf_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "traffic": pmc.get("leaf_hash_cols", {}).get("hbm_bytes"), "algorithmic_bytes": leaf_bytes,
                "note": "integer-VALU bound (Poseidon: no MFMA-shaped work); see roofline_ntt for the "
                        "HBM-bound kernel of this step"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        # what actually bounds the leaf hash: integer-VALU issue. Instruction counts and busy fractions come from a
        # separate rocprofv3 --pmc pass of this command (profiles/r01_pmc_valu_bench_v10.json); the rate is this run's.
        roof_valu = None
        try:
            pv = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_valu_bench_v10.json")))["merkle::k_leaf_hash_cols"]
            if (k, log_n) == (COLS, LOG_N):
                insts = pv["SQ_INSTS_VALU"]          # wave-level VALU instructions per launch
                simds, clk = 256 * 4, pv["clock_GHz_est"] * 1e9
                roof_valu = {"kernel": "leaf_hash_cols", "bound": "valu-issue", "unit": "T lane-ops/s",
                             "achieved": insts * 64 / (leaf_ms * 1e-3) / 1e12,
                             "peak": simds * clk / 2 * 64 / 1e12,
                             "peak_note": "1024 SIMDs x one wave64 VALU instruction per 2 cycles (MI355X_MICROARCH.md) at the "
                                          "clock measured under this load (GRBM_GUI_ACTIVE); only an all-VOP2 stream reaches it, "
                                          "VOP3 / v_mad_u64_u32 issue in 3+ cycles (profiles/r01_ubench_valu.txt)",
                             "valu_instructions_per_permutation": insts * 64 / perms,
                             "simd_valu_busy": pv["SQ_ACTIVE_INST_VALU"] / pv["SQ_WAVE_CYCLES"] * 4,
                             "busy_note": "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x 4 resident waves per SIMD (6 fit; the kernel "
                                          "launches 2^20 lanes = 4 waves per SIMD)",
                             "wave_issue_stall_frac": pv["wait_inst_any_frac"], "wave_memory_wait_frac": pv["wait_any_frac"]}
                roof_valu["frac"] = roof_valu["achieved"] / roof_valu["peak"]
        except (OSError, ValueError, KeyError):
            roof_valu = None
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
                    if "ntt16_cols" in pmc and "ntt16_rows" in pmc else None,
                    "traffic_note": "HBM bytes per NTT (both passes) from FETCH_SIZE/WRITE_SIZE; equals the bytes the "
                                    "two-pass structure must move (2 x 16 MiB), i.e. no wasted re-reads"}
        if roof_ntt["achieved"]:
            roof_ntt["frac"] = roof_ntt["achieved"] / HBM_PEAK_GBS
        out = {
            "metric": "ms/NTT at 2^20 Goldilocks (commit-shaped step: 135-column 2^20-row NTT + Poseidon Merkle cap)",
            "value": value, "unit": "ms/NTT", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 (Goldilocks mod 2^64-2^32+1)", "data": "synthetic",
            "config": {"workload": "configs[1]: 2^%d-row x %d-column trace, forward NTT per column "
                                   "(natural->bit-reversed) + Poseidon Merkle cap height %d" % (log_n, k, CAP_H),
                       "log_n": log_n, "columns": k, "cap_height": CAP_H, "sharding": "independent traces per GPU"},
            "roofline": roof, "roofline_valu": roof_valu, "roofline_ntt": roof_ntt, "ntt_variants": variants,
            "kernels_ms_per_step": {name: d["total_ms"] / args.steps for name, d in prof.items()},
            "poseidon_perms_per_s": (perms / (leaf_ms * 1e-3)) if leaf_ms else None,
            "merkle_levels_ms": lvl["total_ms"] / args.steps,
            "cpu_baseline": base,
            "qbench_proofs": qb,
            "groth16_kernels": g16,
        }
        print(json.dumps(out))
    if data is not None:
        data.free()
    cap.free()
    prover.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
