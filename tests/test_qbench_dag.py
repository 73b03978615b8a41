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
    final = dict(dag)["sighash_final0"]
    assert "state_transition/min" in final and sum(d.endswith("/wrapper") for d in final) == 3


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
