"""cp_batcher's host logic (city-rollup_amd/csrc/batcher.inc, the product source) under ThreadSanitizer, without a GPU:
tests/batcher_sim/batcher_sim.cpp includes it over a stand-in for the proving call. Every caller must receive ITS result,
a failing request must fail alone, shapes must never share a batch, a slot must never run two batches at once — and the
race detector must stay silent. Built with the ROCm clang (its TSAN runtime knows pthread_cond_clockwait, which gcc 11's
does not: false "double lock" reports from condition_variable::wait_for)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "tests", "batcher_sim")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def sim():
    src, exe = os.path.join(SIM, "batcher_sim.cpp"), os.path.join(SIM, "batcher_sim.tsan")
    deps = [src, os.path.join(ROOT, "city-rollup_amd", "csrc", "batcher.inc"), os.path.join(ROOT, "include", "cityprover.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        tmp = "%s.%d.tmp" % (exe, os.getpid())
        probe = subprocess.run([CLANG, "-x", "c++", "-fsanitize=thread", "-o", tmp, "-"], input="int main() { return 0; }",
                               capture_output=True, text=True)
        if probe.returncode != 0:   # no clang / no TSAN runtime in this image: nothing to run (a compile error below is a failure)
            pytest.skip("no ThreadSanitizer toolchain at %s: %s" % (CLANG, probe.stderr[-300:]))
        r = subprocess.run([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include"), src, "-lpthread",
                            "-o", tmp], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        os.replace(tmp, exe)
    return exe


# lanes (slots), caller threads, calls per thread, max_batch, linger (us)
@pytest.mark.parametrize("cfg", [(1, 16, 100, 32, 0), (3, 32, 60, 8, 100), (4, 64, 40, 16, 300), (2, 48, 40, 4, 0), (1, 1, 50, 8, 1000)])
def test_batcher_logic_under_thread_sanitizer(sim, cfg):
    r = subprocess.run([sim] + [str(x) for x in cfg], capture_output=True, text=True, timeout=300)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, r.stdout + r.stderr[-1000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    calls = cfg[1] * cfg[2]
    assert res["calls"] == res["proofs"] == calls and res["wrong"] == 0 and res["violations"] == 0
    assert res["largest_batch"] <= cfg[3]
    expected_failures = sum(1 for t in range(cfg[1]) for k in range(cfg[2]) if (t * 131 + k * 17) % 97 == 0)
    assert res["failed_as_expected"] == expected_failures
    if cfg[1] > 1:
        assert res["batches"] < calls          # calls were merged
    else:
        assert res["batches"] == calls and res["largest_batch"] == 1
