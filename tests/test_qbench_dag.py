"""Host logic of the qbench DAG replay (tools/qbench_replay.py): the proof-level DAG of one example block has the
shape SURVEY.md §8(d) M1 states (64 proofs: 20 op leaves + 14 op aggregations + 2 x (root-agg + minifier) +
(state transition + minifier) + 3 x (sighash inner + 3 minifiers + wrapper) + 3 x (final + minifier) + 3 wraps), and
the scheduler releases a proof only after all its dependencies."""
import os
import sys
import threading

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import qbench_replay as Q  # noqa: E402


def test_block_dag_shape():
    dag = Q.block_dag()
    names = [n for n, _ in dag]
    assert len(dag) == 64 and len(set(names)) == 64
    assert sum("/leaf" in n for n in names) == 20
    assert sum("/agg" in n for n in names) == 14
    assert sum(n.startswith("sighash") and "final" not in n for n in names) == 15
    seen = set()
    for n, deps in dag:           # topological order, dependencies exist
        assert all(d in seen for d in deps), n
        seen.add(n)
    assert Q.critical_path(dag) == 10   # leaf, agg, agg, part, min, transition, min, final, min, wrap
    assert len(Q.job_dag()) == 46
    final = dict(dag)["sighash_final0"]
    assert "state_transition/min" in final and sum(d.endswith("/wrapper") for d in final) == 3


def test_job_dag_equals_the_one_in_example_bin(golden_dir):
    """The job DAG the replay uses == the DAG `plan_jobs` left in qbench_data/example.bin (counter / goal / next-jobs
    triplets, extracted by tests/golden/make_golden.py): same jobs per (circuit type, sub group), and a job waits for
    exactly the jobs of the groups that release it (barrier groups of topic 4 resolved transitively)."""
    import json
    fx = json.load(open(os.path.join(golden_dir, "example_job_dag.json")))
    groups = {tuple(g["group"]): g for g in fx}
    # group -> the job groups whose completion releases it, looking through the AggregateJobs barriers (topic 4)
    released_by = {}
    for gk, g in groups.items():
        for nxt in g["next"]:
            released_by.setdefault(tuple(nxt[:4]), set()).add(gk)

    def real_sources(gk):
        out = set()
        for src in released_by.get(gk, ()):
            out |= real_sources(src) if src[0] == 4 else {src}
        return out

    want = {}   # (circuit_type, sub_group) -> (number of jobs, set of (circuit_type, sub_group) it waits for)
    for gk, g in groups.items():
        if gk[0] != 0:
            continue
        targets = {tuple(n[:4]) for n in g["next"]}
        want.setdefault((gk[1], gk[3]), [g["goal"], set()])
        for t in targets:
            if t[0] == 0:
                n_jobs = sum(1 for n in g["next"] if tuple(n[:4]) == t)
                want.setdefault((t[1], t[3]), [n_jobs, set()])
    for gk in groups:
        if gk[0] == 0:
            want[(gk[1], gk[3])][1] = {(s[1], s[3]) for s in real_sources(gk)}
    want = {k: (v[0], v[1]) for k, v in want.items()}

    got = {}
    key_of = {name: key for name, key, _ in Q.job_dag()}
    for name, key, deps in Q.job_dag():
        n, d = got.get(key[:2], (0, set()))
        got[key[:2]] = (n + 1, d | {key_of[x][:2] for x in deps})
    assert got == want
    assert sum(n for n, _ in got.values()) == 46   # the 46 GenerateStandardProof jobs of the block


def test_scheduler_respects_dependencies():
    rp = Q.Replay(3)
    deps = {(b, n): [(b, d) for d in ds] for b in range(3) for n, ds in Q.block_dag()}
    finished, lock, t = set(), threading.Lock(), [0.0]

    def worker():
        while True:
            batch = rp.take(5)
            if not batch:
                return
            with lock:
                for task in batch:
                    assert all(d in finished for d in deps[task]), task
                t[0] += 1.0
                now = t[0]
            with lock:
                finished.update(batch)
            rp.done(batch, now)

    ths = [threading.Thread(target=worker) for _ in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert len(finished) == 3 * 64 and all(x is not None for x in rp.block_done_at)
