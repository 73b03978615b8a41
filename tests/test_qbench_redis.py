"""`cityprover_qbench --mode redis-worker` (the Redis half): the harness as one `l2-worker` of a live
deployment — RSMQ pop_message("JOB"), the Redis proof store (`proofs` / `proof_counters` hashes), counters / goals /
next jobs and the CoreJobCompleted notification exactly as actors/simple.rs:57-115 does them — against the in-process
server of tests/fake_redis.py holding the reference's own example block (tests/golden/qbench_example.bin). --dry-run: the
whole protocol, no proving, no GPU."""
import json
import os
import struct
import subprocess

import pytest

from fake_redis import FakeRedis
from test_qbench_harness import EXE, build_harness, expected_pop_order


def dump_entries(golden_dir):
    b = open(os.path.join(golden_dir, "qbench_example.bin"), "rb").read()
    off = 8 + 4 + 6 * 8
    (n,) = struct.unpack_from("<Q", b, off)
    off += 8
    out = []
    for _ in range(n):
        key = b[off:off + 24]
        (ln,) = struct.unpack_from("<Q", b, off + 24)
        out.append((key, b[off + 32:off + 32 + ln]))
        off += 32 + ln
    assert b[off:] == bytes(8)      # the `counters` map of the memory store: empty in this dump
    return out


def job_json(j, goal_id=4):
    topic, ct, group, sub, task = j
    return json.dumps({"topic": topic, "goal_id": goal_id, "circuit_type": ct, "group_id": group, "sub_group_id": sub, "task_index": task,
                       "data_type": 0, "data_index": 0}, separators=(",", ":"))


def job_tuple(body):
    d = json.loads(body)
    return (d["topic"], d["circuit_type"], d["group_id"], d["sub_group_id"], d["task_index"])


def leaves(golden_dir):
    """the jobs plan_jobs enqueues itself (job_planner.rs:141-151): everything the pop order holds before the first released job"""
    cfg = json.load(open(os.path.join(golden_dir, "example_dump_index.json")))["config"]
    return expected_pop_order(golden_dir)[:sum(cfg["job_config"]) + cfg["job_config"][5] + 1]   # the op leaves + one introspection per deposit + 1


def load_block(r, golden_dir):
    """what the orchestrator leaves in Redis before the workers start: every record of the dump in `proofs`, the queues
    created, the leaf jobs sent"""
    for key, val in dump_entries(golden_dir):
        r.store.hashes.setdefault(b"proofs", {})[key] = val
    r.store.create_queue("JOB")
    r.store.create_queue("NOTIFICATIONS")
    for j in leaves(golden_dir):
        r.store.send("JOB", job_json(j))


def worker(uri, *extra):
    build_harness()
    return subprocess.run([EXE, "--mode", "redis-worker", "--redis", uri, "--dry-run", "--drain", *extra], capture_output=True, text=True, timeout=120)


def key_of(j, data_type, data_index=0, goal_id=4, task=None):
    topic, ct, group, sub, t = j
    return struct.pack("<BQBIIIBB", topic, goal_id, ct, group, sub, t if task is None else task, data_type, data_index)


def test_one_worker_drains_the_example_block(golden_dir):
    want = expected_pop_order(golden_dir)
    with FakeRedis() as r:
        load_block(r, golden_dir)
        before = dict(r.store.hashes[b"proofs"])
        p = worker(r.uri)
        assert p.returncode == 0, p.stderr
        res = json.loads(p.stdout.strip().splitlines()[-1])
        assert res["mode"] == "redis-worker" and res["dry_run"] is True
        assert res["jobs"] == 60 and res["proving_jobs"] == 46 and res["proofs"] == 64 and res["notifications"] == 1 and res["queue_left"] == 0
        assert res["jobs_released"] == 60 - len(leaves(golden_dir))
        # (a) the jobs came off the queue in the order the reference's loop would pop them
        assert [job_tuple(b) for b in r.store.popped[b"rsmq:JOB"]] == want
        # (b) every proving job left an output under get_output_id(), written with HSETNX, and nothing else of `proofs` changed
        proofs = r.store.hashes[b"proofs"]
        outs = {key_of(j, 8) for j in want if j[0] == 0}
        assert set(proofs) - set(before) == outs and all(proofs[k] == before[k] for k in before)
        wrap = [k for k in outs if k[9] == 36]
        assert len(wrap) == 3 and all(len(proofs[k]) > 1 for k in wrap)   # GROTH16_DISABLED_DEV_MODE: the zero proof's bincode
        assert all(proofs[k] == b"\0" for k in outs if k[9] != 36)    # dry run: nothing was proved
        assert {a[1] for n, a in r.store.log if n == "HSETNX"} == outs
        # (c) every group counter stands at its goal, in `proof_counters` (the Redis store keeps them apart from `proofs`)
        dag = json.load(open(os.path.join(golden_dir, "example_job_dag.json")))
        counters = r.store.hashes[b"proof_counters"]
        assert len(counters) == len(dag)
        for g in dag:
            assert int(counters[key_of(tuple(g["group"]) + (0,), 16)]) == g["goal"]
        # (d) the orchestrator was told once, in serde_json of QueueNotification::CoreJobCompleted; the released jobs went out in
        # the reference's JSON field order
        assert r.store.queue_bodies("NOTIFICATIONS") == [b"0"] and r.store.queue_bodies("JOB") == []
        sent = [a[2] for n, a in r.store.log if n == "HSET" and a[0] == b"rsmq:JOB:Q"]
        assert len(sent) == res["jobs_released"] and all(list(json.loads(s)) == ["topic", "goal_id", "circuit_type", "group_id", "sub_group_id", "task_index", "data_type", "data_index"] for s in sent)
        assert int(r.store.hashes[b"rsmq:JOB:Q"][b"totalsent"]) == res["jobs_released"] and int(r.store.hashes[b"rsmq:JOB:Q"][b"totalrecv"]) == 60


def test_two_workers_share_the_queue(golden_dir):
    """the point of the Redis half: several workers, one queue; HINCRBY decides who releases a group's next jobs"""
    with FakeRedis() as r:
        load_block(r, golden_dir)
        build_harness()
        cmd = [EXE, "--mode", "redis-worker", "--redis", r.uri, "--dry-run"]
        # A polls until it has seen the notify job go by or B has; both leave after their share
        a = subprocess.Popen(cmd + ["--drain"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        b = subprocess.Popen(cmd + ["--drain"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        outs = [p.communicate(timeout=120) for p in (a, b)]
        assert a.returncode == 0 and b.returncode == 0, outs
        if r.store.queue_bodies("JOB"):   # a worker that found the queue momentarily empty has left; one more finishes the block
            assert worker(r.uri).returncode == 0
        assert sorted(job_tuple(x) for x in r.store.popped[b"rsmq:JOB"]) == sorted(expected_pop_order(golden_dir))
        dag = json.load(open(os.path.join(golden_dir, "example_job_dag.json")))
        for g in dag:
            assert int(r.store.hashes[b"proof_counters"][key_of(tuple(g["group"]) + (0,), 16)]) == g["goal"]
        assert r.store.queue_bodies("NOTIFICATIONS") == [b"0"] and r.store.queue_bodies("JOB") == []


def test_rounds_of_several_messages(golden_dir):
    """--redis-batch N: a round takes up to N messages that are in the queue and does the jobs together (on the GPU: shared
    launches). Same outputs, same counters, one notification; fewer rounds than messages; the order of service is a
    permutation of the reference's."""
    with FakeRedis() as r:
        load_block(r, golden_dir)
        p = worker(r.uri, "--redis-batch", "8")
        assert p.returncode == 0, p.stderr
        res = json.loads(p.stdout.strip().splitlines()[-1])
        assert res["jobs"] == 60 and res["proving_jobs"] == 46 and res["proofs"] == 64 and res["notifications"] == 1 and res["queue_left"] == 0
        assert res["redis_batch"] == 8 and 8 <= res["rounds"] < 60
        want = expected_pop_order(golden_dir)
        assert sorted(job_tuple(b) for b in r.store.popped[b"rsmq:JOB"]) == sorted(want)
        proofs = r.store.hashes[b"proofs"]
        assert {key_of(j, 8) for j in want if j[0] == 0} <= set(proofs)
        dag = json.load(open(os.path.join(golden_dir, "example_job_dag.json")))
        for g in dag:
            assert int(r.store.hashes[b"proof_counters"][key_of(tuple(g["group"]) + (0,), 16)]) == g["goal"]
        assert r.store.queue_bodies("NOTIFICATIONS") == [b"0"]
    with FakeRedis() as r:       # --max-jobs is honoured inside a round too
        load_block(r, golden_dir)
        p = worker(r.uri, "--redis-batch", "8", "--max-jobs", "13")
        assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["jobs"] == 13


def test_max_jobs_and_resume(golden_dir):
    """a worker that stops (here: --max-jobs) leaves a state another one picks up: nothing is held outside Redis"""
    with FakeRedis() as r:
        load_block(r, golden_dir)
        p = worker(r.uri, "--max-jobs", "17")
        assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["jobs"] == 17
        assert len(r.store.queue_bodies("JOB")) > 0
        p = worker(r.uri)
        assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["jobs"] == 43
        assert [job_tuple(b) for b in r.store.popped[b"rsmq:JOB"]] == expected_pop_order(golden_dir)


def test_failures_are_loud(golden_dir):
    with FakeRedis() as r:
        # (a) a job whose witness is not in the store (get_bytes_by_id fails in the reference: the worker's loop logs and goes on;
        # this harness stops, it is a measurement tool)
        load_block(r, golden_dir)
        first = leaves(golden_dir)[0]
        del r.store.hashes[b"proofs"][key_of(first, 0)]
        p = worker(r.uri)
        assert p.returncode != 0 and "not found" in p.stderr
    with FakeRedis() as r:
        # (b) a proof the witness names is missing: the aggregation job cannot run
        load_block(r, golden_dir)
        p = worker(r.uri, "--max-jobs", "6")      # the introspection leaves and some ops
        assert p.returncode == 0
        done = [k for k in r.store.hashes[b"proofs"] if k[22] == 8]
        assert done
        for k in done:
            del r.store.hashes[b"proofs"][k]
        p = worker(r.uri)
        assert p.returncode != 0 and ("not found" in p.stderr or "is empty" in p.stderr)
    with FakeRedis() as r:
        # (c) the NOTIFICATIONS queue was never created: rsmq's queueNotFound
        load_block(r, golden_dir)
        del r.store.hashes[b"rsmq:NOTIFICATIONS:Q"]
        p = worker(r.uri)
        assert p.returncode != 0 and "NOTIFICATIONS not found" in p.stderr
    with FakeRedis() as r:
        # (d) a message that is not a job id
        load_block(r, golden_dir)
        r.store.zsets[b"rsmq:JOB"].clear()
        r.store.send("JOB", '{"topic":9,"goal_id":4,"circuit_type":0,"group_id":1,"sub_group_id":0,"task_index":0,"data_type":0,"data_index":0}')
        p = worker(r.uri)
        assert p.returncode != 0 and "unknown topic" in p.stderr
    # (e) no server
    p = worker("127.0.0.1:1")
    assert p.returncode != 0 and "cannot connect" in p.stderr
    # (f) without --dry-run the mode needs the pack and a GPU: no CPU fallback
    build_harness()
    p = subprocess.run([EXE, "--mode", "redis-worker", "--redis", "127.0.0.1:1"], capture_output=True, text=True)
    assert p.returncode != 0


@pytest.mark.gpu
def test_redis_worker_proves_the_block_on_the_gpu(golden_dir, tmp_path):
    """the same protocol with the proving in: every job's stages proved on the device from the pack (each proof gated against
    the oracle's bytes for the same witness when the worker opens), the final stage's bytes stored as the job's output"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(EXE)))
    import make_circuit_pack
    pack = make_circuit_pack.make_pack(str(tmp_path / "pack"), n_circuits=3, db=7, small=True)
    with FakeRedis() as r:
        load_block(r, golden_dir)
        build_harness()
        p = subprocess.run([EXE, "--mode", "redis-worker", "--redis", r.uri, "--pack", pack, "--drain"], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr
        res = json.loads(p.stdout.strip().splitlines()[-1])
        assert res["jobs"] == 60 and res["proofs"] == 64 and res["dry_run"] is False and res["queue_left"] == 0
        proofs = r.store.hashes[b"proofs"]
        outs = [k for k in proofs if k[22] == 8]
        assert len(outs) == 46 and all(len(proofs[k]) > 1_000 for k in outs if k[9] != 36)
        assert r.store.queue_bodies("NOTIFICATIONS") == [b"0"]
        one_by_one = {k: proofs[k] for k in outs}
    # rounds of up to 16 messages: shared launches, the same bytes for every job
    with FakeRedis() as r:
        load_block(r, golden_dir)
        p = subprocess.run([EXE, "--mode", "redis-worker", "--redis", r.uri, "--pack", pack, "--drain", "--redis-batch", "16"],
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr
        res = json.loads(p.stdout.strip().splitlines()[-1])
        assert res["jobs"] == 60 and res["proofs"] == 64 and res["rounds"] < 40 and res["launches"] < 64
        proofs = r.store.hashes[b"proofs"]
        assert {k: proofs[k] for k in proofs if k[22] == 8} == one_by_one
