"""tools/cityprover_qbench — the q-bench harness: reads `BlockProofStoreDump` bincode natively
(the reference's own qbench_data/example.bin, kept as tests/golden/qbench_example.bin), re-plans the block with its
restatement of plan_jobs, drains the job queue with the reference's counter / goal / next-jobs semantics on a pool of
workers and writes the reference's `[{"job_id", "duration"}]` output. CPU tests use --dry-run (the whole schedule, no
proving, no GPU); the GPU tests prove every job and compare every proof with the oracle's bytes."""
import json
import os
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "cityprover_qbench")


def build_harness():
    sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
    from cityprover import build
    build.build()
    srcs = [EXE + ".cpp", os.path.join(ROOT, "tools", "qbench", "jobs.h"), os.path.join(ROOT, "tools", "qbench", "pack.h"),
            os.path.join(ROOT, "tools", "qbench", "redis.h"),
            os.path.join(ROOT, "include", "cityprover.h")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(s) for s in srcs):
        tmp = "%s.%d.tmp" % (EXE, os.getpid())
        subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tools"), srcs[0],
                        "-L" + os.path.join(ROOT, "city-rollup_amd"), "-lcityprover_hip", "-Wl,-rpath,$ORIGIN/../city-rollup_amd",
                        "-lpthread", "-o", tmp], check=True)
        os.replace(tmp, EXE)


def run(args, ok=True):
    build_harness()
    r = subprocess.run([EXE] + args, capture_output=True, text=True)
    if ok:
        assert r.returncode == 0, r.stderr
        return json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode != 0
    return r.stderr


def key_fields(hexkey):
    return struct.unpack("<BQBIIIBB", bytes.fromhex(hexkey))


def expected_pop_order(golden_dir):
    """Independent restatement of the queue semantics (actors/simple.rs:57-115, events.rs:29-48) over the DAG that
    make_golden.py decoded from the dump's own counter / goal / next-jobs records (not over the harness's planner)."""
    fx = json.load(open(os.path.join(golden_dir, "example_job_dag.json")))
    groups = {tuple(g["group"]): (g["goal"], [tuple(n) for n in g["next"]]) for g in fx}
    cfg = json.load(open(os.path.join(golden_dir, "example_dump_index.json")))["config"]
    reg, claim, transfer, addw, procw, addd = cfg["job_config"]
    gid = lambda ct: ct + 0xCF00
    # leaves in plan_jobs' order (job_planner.rs:141-151): introspections, then the op leaves
    queue = [(0, 33, gid(33), 0, i) for i in range(addd + 1)]
    for ct, n in ((0, reg), (4, claim), (6, transfer), (8, addw), (10, procw), (2, addd)):
        queue += [(0, ct, gid(ct), 0, i) for i in range(n)]
    counters, popped = {}, []
    while queue:
        j = queue.pop(0)
        popped.append(j)
        if j[0] == 3:
            continue
        goal, nxt = groups[j[:4]]
        counters[j[:4]] = counters.get(j[:4], 0) + 1
        if goal and counters[j[:4]] == goal:
            queue += nxt
    return popped


def test_dry_run_reproduces_the_reference_schedule(golden_dir, tmp_path):
    dump = os.path.join(golden_dir, "qbench_example.bin")
    out, trace = str(tmp_path / "out.json"), str(tmp_path / "trace.jsonl")
    res = run(["-i", dump, "-o", out, "--dry-run", "--check-plan", "--contexts", "1", "--batch", "1", "--trace", trace])
    assert res["blocks_complete"] == 1 and res["jobs"] == 46 and res["proofs"] == 64
    # (a) the output file is the reference's format: a JSON list of {"job_id": 48 hex chars, "duration": ms}
    bench = json.load(open(out))
    assert len(bench) == 46 and all(set(b) == {"job_id", "duration"} and len(b["job_id"]) == 48 and isinstance(b["duration"], int) for b in bench)
    assert open(out).read().startswith('[\n  {\n    "job_id": "')       # serde_json::to_vec_pretty's layout
    # (b) the job-id multiset == the GenerateStandardProof input witnesses the dump holds
    idx = json.load(open(os.path.join(golden_dir, "example_dump_index.json")))["entries"]
    want = sorted(e["key"] for e in idx if e["topic"] == 0 and e["data_type"] == 0)
    assert sorted(b["job_id"] for b in bench) == want
    # (c) pop order (barrier and notify jobs included) == the independent simulation over the dump's own DAG records
    popped = []
    for line in open(trace):
        d = json.loads(line)
        if "popped" in d:
            popped.append((d["topic"], d["circuit_type"], d["group_id"], d["sub_group_id"], d["task_index"]))
    assert popped == expected_pop_order(golden_dir)
    assert len(popped) == 60 and popped[-1][0] == 3     # 46 proving jobs + 13 AggregateJobs barriers + the notify job
    # (d) --check-plan passed: the harness's plan_jobs wrote exactly the records the dump carries (checked inside)


def test_worker_pool_iterations_and_blocks_in_flight(golden_dir, tmp_path):
    dump = os.path.join(golden_dir, "qbench_example.bin")
    out = str(tmp_path / "out.json")
    res = run(["-i", dump, dump, "-o", out, "-n", "3", "--dry-run", "--contexts", "4", "--batch", "8", "--blocks-in-flight", "4"])
    assert res["dumps"] == 2 and res["blocks"] == 6 and res["blocks_complete"] == 6 and res["jobs"] == 6 * 46 and res["proofs"] == 6 * 64
    bench = json.load(open(out))
    from collections import Counter
    assert set(Counter(b["job_id"] for b in bench).values()) == {6}


def test_sliding_window_of_blocks(golden_dir, tmp_path):
    """--sliding: a block that completes starts the next one (a window of --blocks-in-flight blocks instead of waves); every
    block completes, every job runs exactly once per block, and never more blocks than the window are open at a time."""
    dump = os.path.join(golden_dir, "qbench_example.bin")
    out, trace = str(tmp_path / "out.json"), str(tmp_path / "trace.jsonl")
    res = run(["-i", dump, "-o", out, "-n", "9", "--dry-run", "--dry-run-job-us", "200", "--contexts", "3", "--batch", "8",
               "--blocks-in-flight", "3", "--sliding", "--trace", trace])
    assert res["blocks"] == 9 and res["blocks_complete"] == 9 and res["jobs"] == 9 * 46 and res["proofs"] == 9 * 64
    from collections import Counter
    assert set(Counter(b["job_id"] for b in json.load(open(out))).values()) == {9}
    popped = [json.loads(l) for l in open(trace) if '"popped"' in l]
    open_blocks, worst = 0, 0
    for p in popped:          # a block opens with its first leaf (the first introspection job) and closes with its notify job
        if (p["topic"], p["circuit_type"], p["task_index"]) == (0, 33, 0):
            open_blocks += 1
        if p["topic"] == 3:
            open_blocks -= 1
        worst = max(worst, open_blocks)
    assert worst <= 3 and open_blocks == 0
    assert "--sliding and --ref-counters" in run(["-i", dump, "--dry-run", "--sliding", "--ref-counters"], ok=False)


def test_stage_units_in_the_dry_run(golden_dir, tmp_path):
    """--dry-run-stages: every stage of a job is a queue entry of its own (as in a real run). The block still completes with 46
    jobs = 64 proofs, each job is recorded once, the pop order lists every job once (first stages only), and a job's recorded
    duration spans all its stages."""
    dump = os.path.join(golden_dir, "qbench_example.bin")
    out, trace = str(tmp_path / "out.json"), str(tmp_path / "trace.jsonl")
    res = run(["-i", dump, "-o", out, "--dry-run", "--dry-run-stages", "--dry-run-job-us", "1000", "--contexts", "3", "--batch", "8",
               "-n", "2", "--trace", trace])
    assert res["blocks_complete"] == 2 and res["jobs"] == 92 and res["proofs"] == 128
    bench = json.load(open(out))
    from collections import Counter
    assert len(bench) == 92 and set(Counter(b["job_id"] for b in bench).values()) == {2}
    rows = [json.loads(l) for l in open(trace)]
    assert sum(1 for r in rows if "popped" in r) == 2 * 60
    timed = [r for r in rows if "job_id" in r]
    five = [r for r in timed if r["circuit_type"] == 33]          # the sighash introspection jobs: five proofs each
    one = [r for r in timed if r["circuit_type"] == 0]            # the register-user leaves: one proof each
    assert min(r["end_ms"] - r["start_ms"] for r in five) >= 4.5 and max(r["end_ms"] - r["start_ms"] for r in one) < 4.5


def test_reference_counter_quirk(golden_dir, tmp_path):
    """The reference never resets `counters` between iterations (memory_proof_store/mod.rs:77-83; qbench.rs:44-61): from
    the second iteration on no group reaches its goal and only the 23 leaf jobs run. --ref-counters reproduces that."""
    dump = os.path.join(golden_dir, "qbench_example.bin")
    out = str(tmp_path / "out.json")
    res = run(["-i", dump, "-o", out, "-n", "2", "--dry-run", "--ref-counters", "--contexts", "1", "--batch", "1"])
    assert res["blocks_complete"] == 1 and res["jobs"] == 46 + 23
    assert len(json.load(open(out))) == 46 + 23


def test_bad_inputs_fail_loudly(golden_dir, tmp_path):
    dump = open(os.path.join(golden_dir, "qbench_example.bin"), "rb").read()
    p = str(tmp_path / "bad.bin")
    open(p, "wb").write(dump[:-5])
    assert "truncated" in run(["-i", p, "--dry-run"], ok=False) or "trailing" in run(["-i", p, "--dry-run"], ok=False)
    open(p, "wb").write(dump + b"\0")
    assert "trailing" in run(["-i", p, "--dry-run"], ok=False)
    # a job whose input witness is missing fails like the reference's store ("Data not found")
    idx = json.load(open(os.path.join(golden_dir, "example_dump_index.json")))["entries"]
    victim = next(e for e in idx if e["topic"] == 0 and e["data_type"] == 0 and e["circuit_type"] == 7)
    key = bytes.fromhex(victim["key"])
    at = dump.index(key)
    cut = dump[:at] + dump[at + 24 + 8 + victim["len"]:]
    cut = cut[:44 + 16] + struct.pack("<Q", struct.unpack_from("<Q", dump, 60)[0] - 1) + cut[68:]   # entry count - 1
    open(p, "wb").write(cut)
    assert "Data not found" in run(["-i", p, "--dry-run"], ok=False)
    # a signature proof a leaf job names is missing: the job fails before proving
    sig = next(e for e in idx if e["topic"] == 2 and e["len"] > 0)
    at = dump.index(bytes.fromhex(sig["key"]))
    cut = dump[:at] + dump[at + 24 + 8 + sig["len"]:]
    cut = cut[:60] + struct.pack("<Q", struct.unpack_from("<Q", dump, 60)[0] - 1) + cut[68:]
    open(p, "wb").write(cut)
    assert "Data not found" in run(["-i", p, "--dry-run"], ok=False)
    assert "--pack" in run(["-i", os.path.join(golden_dir, "qbench_example.bin")], ok=False)


def test_mutated_dumps_are_refused_or_replayed_never_crash(golden_dir, tmp_path):
    """The dump parser against damaged input: truncations at every structural boundary, overwritten length prefixes,
    flipped bytes in the records' headers, random garbage. Each run must end by itself with exit code 0 (the damage was in
    bytes the schedule does not read) or 1 (a ParseError / plan mismatch) — never a signal, never a hang."""
    import random
    build_harness()
    raw = open(os.path.join(golden_dir, "qbench_example.bin"), "rb").read()
    rnd = random.Random(5)
    cases = []
    for cut in [0, 1, 7, 8, 9, 15, 16, 24, 31, 32, 40, 100, 1000, len(raw) // 2, len(raw) - 9, len(raw) - 8, len(raw) - 1]:
        cases.append(raw[:cut])
    for _ in range(40):                                   # a u64 length prefix somewhere in the first 4 KiB made huge / zero / off by one
        b = bytearray(raw)
        o = rnd.randrange(0, 4096) & ~7
        b[o:o + 8] = struct.pack("<Q", rnd.choice([0, 1, 2 ** 63, 2 ** 64 - 1, 2 ** 32, len(raw), len(raw) + 1,
                                                   struct.unpack("<Q", raw[o:o + 8])[0] + 1]))
        cases.append(bytes(b))
    for _ in range(60):                                   # byte flips: half in the head of the file, half anywhere
        b = bytearray(raw)
        for _ in range(rnd.randrange(1, 6)):
            o = rnd.randrange(0, 2048) if rnd.random() < 0.5 else rnd.randrange(0, len(raw))
            b[o] ^= 1 << rnd.randrange(8)
        cases.append(bytes(b))
    cases.append(bytes(rnd.randrange(256) for _ in range(5000)))
    cases.append(raw + b"\x00" * 8)
    p = str(tmp_path / "mutated.bin")
    seen = {0: 0, 1: 0}
    for c in cases:
        open(p, "wb").write(c)
        r = subprocess.run([EXE, "-i", p, "--dry-run", "--contexts", "2", "--batch", "4"], capture_output=True, text=True, timeout=60)
        assert r.returncode in (0, 1), (r.returncode, r.stderr[-300:])
        if r.returncode == 1:
            assert r.stderr.strip(), "an error exit must say why"
        seen[r.returncode] += 1
    assert seen[1] >= 20                                  # truncations and broken length prefixes are noticed; flips inside proof bytes are not read


def test_harness_fails_loudly_without_a_gpu(golden_dir, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    err = run(["-i", os.path.join(golden_dir, "qbench_example.bin"), "--pack", str(tmp_path)], ok=False)
    assert "no HIP device" in err and "no CPU fallback" in err


def test_scheduler_under_thread_sanitizer(golden_dir, tmp_path):
    """The harness's ready queue and proof stores with 32 worker threads, four blocks in flight, under ThreadSanitizer
    (--dry-run: the whole schedule, no proving): every block completes and the race detector stays silent."""
    build_harness()
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    exe = str(tmp_path / "qbench_tsan")
    r = subprocess.run([clang, "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "tools"), EXE + ".cpp", "-L" + os.path.join(ROOT, "city-rollup_amd"), "-lcityprover_hip",
                        "-Wl,-rpath," + os.path.join(ROOT, "city-rollup_amd"), "-lpthread", "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        probe = subprocess.run([clang, "-x", "c++", "-fsanitize=thread", "-o", exe, "-"], input="int main() { return 0; }",
                               capture_output=True, text=True)
        if probe.returncode != 0:
            pytest.skip("no ThreadSanitizer toolchain: " + probe.stderr[-300:])
        raise AssertionError(r.stderr[-3000:])
    dump = os.path.join(golden_dir, "qbench_example.bin")
    for args in (["--contexts", "32", "--batch", "1", "-n", "16", "--blocks-in-flight", "4"],
                 ["--contexts", "16", "--batch", "4", "-n", "8", "--blocks-in-flight", "8"],
                 # the stage machinery (a job of k proofs passes through the queue k times: requeue, longest chain first,
                 # shared short queues) and the sliding window, with simulated launch times so that the workers interleave
                 ["--contexts", "6", "--batch", "8", "-n", "12", "--blocks-in-flight", "3", "--dry-run-stages", "--dry-run-job-us", "200"],
                 ["--contexts", "6", "--batch", "8", "-n", "12", "--blocks-in-flight", "3", "--dry-run-stages", "--dry-run-job-us", "200", "--sliding"],
                 # the STARK of a sighash job as a queue stage of its own, taken by any worker / only by workers of its own
                 ["--contexts", "6", "--batch", "8", "-n", "12", "--blocks-in-flight", "3", "--dry-run-stages", "--dry-run-job-us", "200",
                  "--stark-log-rows", "10"],
                 ["--contexts", "4", "--batch", "8", "-n", "12", "--blocks-in-flight", "3", "--dry-run-stages", "--dry-run-job-us", "200",
                  "--stark-log-rows", "10", "--stark-contexts", "2"],
                 ["--contexts", "4", "--batch", "8", "-n", "12", "--blocks-in-flight", "3", "--dry-run-stages", "--dry-run-job-us", "100",
                  "--stark-log-rows", "10", "--stark-contexts", "1", "--sliding"]):
        r = subprocess.run([exe, "-i", dump, "--dry-run"] + args, capture_output=True, text=True, timeout=300)
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
        assert r.returncode == 0, r.stderr[-1000:]
        res = json.loads(r.stdout.strip().splitlines()[-1])
        assert res["blocks_complete"] == res["blocks"] == int(args[5])
        if "--stark-log-rows" in args:
            assert res["stark_proofs"] == 3 * res["blocks"]


@pytest.mark.gpu
def test_one_block_end_to_end_with_byte_parity(golden_dir, tmp_path):
    """One example block proved job by job on 2 worker contexts: 46 jobs = 64 proofs, every proof byte-identical to the
    CPU oracle's (recorded in the pack's witness files), output in the reference's format."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_circuit_pack
    pack = make_circuit_pack.make_pack(str(tmp_path / "pack"), n_circuits=3, db=7, small=True)
    dump = os.path.join(golden_dir, "qbench_example.bin")
    out = str(tmp_path / "out.json")
    res = run(["-i", dump, "-o", out, "--pack", pack, "--contexts", "2", "--batch", "8", "--check-plan"])
    assert res["blocks_complete"] == 1 and res["jobs"] == 46 and res["proofs"] == 64
    assert res["proofs_byte_checked"] >= 64 and res["workers"] >= 2
    bench = json.load(open(out))
    idx = json.load(open(os.path.join(golden_dir, "example_dump_index.json")))["entries"]
    assert sorted(b["job_id"] for b in bench) == sorted(e["key"] for e in idx if e["topic"] == 0 and e["data_type"] == 0)
    # the reference's single-threaded loop: one context, one job at a time — same jobs, same order as the dry run
    out1, trace1, trace0 = str(tmp_path / "o1.json"), str(tmp_path / "t1.jsonl"), str(tmp_path / "t0.jsonl")
    run(["-i", dump, "-o", out1, "--pack", pack, "--devices", "0", "--contexts", "1", "--batch", "1", "--trace", trace1])
    run(["-i", dump, "--dry-run", "--contexts", "1", "--batch", "1", "--trace", trace0])
    pops = lambda p: [json.loads(x)["popped"] for x in open(p) if "popped" in x]
    assert pops(trace1) == pops(trace0)
    # the Groth16 job with the prover kernels on a synthetic key (five MSMs + quotient + the 192-byte packing) instead of
    # the all-zero dev-mode proof: three per block
    res = run(["-i", dump, "--pack", pack, "--contexts", "2", "--batch", "8", "--groth16-log-size", "10"])
    assert res["blocks_complete"] == 1 and res["groth16_proofs"] == 3 and res["groth16_log_constraints"] == 10
    # the three SHA-256 STARKs of a block (tools/qbench/stark_stage.h: cp_stark_prove on a synthetic AIR of the reference's shape, once
    # per sighash job), alone and together with the Groth16 stage; the proofs of the block are byte-checked as before
    res = run(["-i", dump, "--pack", pack, "--contexts", "2", "--batch", "8", "--stark-log-rows", "7", "--check-plan"])
    assert res["blocks_complete"] == 1 and res["stark_proofs"] == 3 and res["stark_log_rows"] == 7 and res["stark_proof_bytes_mean"] > 100000
    assert res["proofs"] == 64 and res["proofs_byte_checked"] >= 64
    res = run(["-i", dump, "-n", "2", "--blocks-in-flight", "2", "--pack", pack, "--contexts", "2", "--batch", "8", "--stark-log-rows", "8",
               "--groth16-log-size", "10"])
    assert res["blocks_complete"] == 2 and res["stark_proofs"] == 6 and res["groth16_proofs"] == 6
    # ... and with contexts of their own for the STARK stages (they prove nothing else, the other contexts no STARK)
    res = run(["-i", dump, "-n", "2", "--blocks-in-flight", "2", "--pack", pack, "--contexts", "2", "--batch", "8", "--stark-log-rows", "7",
               "--stark-contexts", "2", "--check-plan"])
    assert res["blocks_complete"] == 2 and res["stark_proofs"] == 6 and res["workers"] == 4 and res["proofs_byte_checked"] >= 128
    assert "--stark-contexts must be" in run(["-i", dump, "--pack", pack, "--stark-contexts", "2"], ok=False)
    assert "--stark-log-rows must be" in run(["-i", dump, "--pack", pack, "--stark-log-rows", "3"], ok=False)
    assert "one caller per context" in run(["-i", dump, "--pack", pack, "--callers", "4", "--stark-log-rows", "8"], ok=False)
    # the worker pool over a device LIST (section 8(e)): the one GPU of the test box named twice gives two device entries,
    # each with its own contexts and resident circuits, all four workers on the one ready queue
    res = run(["-i", dump, "-n", "2", "--blocks-in-flight", "2", "--pack", pack, "--devices", "0,0", "--contexts", "2", "--batch", "8",
               "--check-plan"])
    assert res["blocks_complete"] == 2 and res["workers"] == 4 and res["devices"] == [0, 0] and res["proofs_byte_checked"] >= 128
    # reference-style loops as THREADS of one process: twelve of them on one context with two lanes, each popping one job and
    # proving one proof per call, merged by a cp_batcher — same jobs, same bytes; and the raw callers loop
    res = run(["-i", dump, "-n", "2", "--blocks-in-flight", "2", "--pack", pack, "--contexts", "1", "--lanes", "2", "--callers", "12",
               "--batch", "8", "--linger-us", "200"])
    assert res["blocks_complete"] == 2 and res["workers"] == 12 and res["callers_per_context"] == 12 and res["proofs_byte_checked"] >= 128
    res = run(["--mode", "callers", "--pack", pack, "--contexts", "1", "--lanes", "2", "--callers", "12", "--batch", "8", "--iters", "6"])
    assert res["proofs"] == 72 and res["proofs_byte_checked"] == 72 and res["batches"] < 72 and res["retried_singly"] == 0
    assert "one caller per context" in run(["-i", dump, "--pack", pack, "--callers", "4", "--groth16-log-size", "10"], ok=False)
    # a tampered witness file (recorded proof altered) is caught by the byte comparison
    wit = os.path.join(pack, "synthetic_0.cpwit")
    from cityprover import files
    w = files.read_witness_file(wit)
    files.write_witness_file(wit, w["digest"], w["wires"], w["public_inputs"], proof=w["proof"][:-1] + bytes([w["proof"][-1] ^ 1]))
    assert "differ" in run(["-i", dump, "--pack", pack, "--contexts", "1"], ok=False)


def test_dry_run_over_eight_device_slots(golden_dir):
    """SURVEY.md section 8(e) without hardware: `--devices 0,...,7 --contexts 3` = 24 worker slots on one ready queue (the
    in-process form of N worker processes on one Redis queue). Every block completes, every proving job is taken exactly once,
    every device's slots get work, and the schedule still equals the dump's own records (--check-plan)."""
    dump = os.path.join(golden_dir, "qbench_example.bin")
    res = run(["-i", dump, "--dry-run", "--dry-run-job-us", "300", "--devices", "0,1,2,3,4,5,6,7", "--contexts", "3", "-n", "16",
               "--blocks-in-flight", "16", "--batch", "4", "--check-plan"])
    assert res["blocks_complete"] == 16 and res["jobs"] == 16 * 46 and res["proofs"] == 16 * 64 and res["workers"] == 24
    per = res["dry_run_jobs_per_device"]
    assert sorted(per) == [str(d) for d in range(8)] and sum(per.values()) == 16 * 46
    assert min(per.values()) > 0, per
    # one block alone cannot use eight devices at once (its critical path is ~9 dependent proofs), but it must still finish
    one = run(["-i", dump, "--dry-run", "--dry-run-job-us", "300", "--devices", "0,1,2,3,4,5,6,7", "--contexts", "1", "--batch", "32"])
    assert one["blocks_complete"] == 1 and sum(one["dry_run_jobs_per_device"].values()) == 46


@pytest.mark.gpu
def test_the_section_8d_workload_one_witness_per_job(golden_dir, tmp_path):
    """one circuit per (job type, stage) the block schedules, one witness per job (seed = job
    index) — 64 DISTINCT proofs of 26 distinct circuits per block. Eight of them carry the CPU oracle's bytes and must equal
    them, the other 56 pass cp_verify before the clock starts, and every proof of the run equals the bytes that passed; ready
    jobs of different types share launches (all circuits of this pack are batch-compatible)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_circuit_pack
    pack = make_circuit_pack.make_pack(str(tmp_path / "pack"), db=7, small=True, n_checked=8, processes=1)
    dump = os.path.join(golden_dir, "qbench_example.bin")
    res = run(["-i", dump, "-n", "4", "--blocks-in-flight", "4", "--pack", pack, "--contexts", "2", "--batch", "16", "--check-plan"])
    assert res["blocks_complete"] == 4 and res["proofs"] == 256 and res["proofs_byte_checked"] == 256
    assert res["circuits"] == 26 and res["witnesses"] == 64 and res["batch_classes"] == 1
    assert res["distinct_proofs"] == 64 and res["distinct_proofs_equal_to_recorded_bytes"] == 8 and res["distinct_proofs_cp_verified"] == 56
    assert res["mean_batch"] > 2.0, res["mean_batch"]
    # the three proofs the sighash jobs (type 33) make at stage 0 are three different byte strings
    from cityprover import files
    ws = [files.read_witness_file(os.path.join(pack, "type33_stage0_job%d.cpwit" % k))["wires"] for k in range(3)]
    assert not (ws[0] == ws[1]).all() and not (ws[1] == ws[2]).all()
    res = run(["--mode", "throughput", "--pack", pack, "--contexts", "2", "--batch", "8", "--iters", "3"])
    assert res["distinct_proofs"] == 64 and res["proofs_byte_checked"] == 48


@pytest.mark.gpu
def test_product_shape_blocks_in_flight_and_throughput(golden_dir, tmp_path):
    """Product shape (n = 2^12, 135 wires, 28 queries, 16-bit PoW): 3 blocks in flight on 3 contexts, every one of the
    192 proofs equal to the oracle's 130 KB; then the raw throughput mode."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_circuit_pack
    pack = make_circuit_pack.make_pack(str(tmp_path / "pack"), n_circuits=2, db=12)
    dump = os.path.join(golden_dir, "qbench_example.bin")
    res = run(["-i", dump, "-n", "3", "--pack", pack, "--contexts", "3", "--batch", "32", "--blocks-in-flight", "3"])
    assert res["blocks_complete"] == 3 and res["proofs"] == 192 and res["proofs_byte_checked"] >= 192
    res = run(["--mode", "throughput", "--pack", pack, "--contexts", "2", "--batch", "8", "--iters", "2"])
    assert res["proofs"] >= 32 and res["proofs_byte_checked"] >= 32
