"""tools/cityprover_qbench (the C++ measurement harness above the C ABI): without a GPU it must fail
loudly; on a GPU it proves the dumped qbench-shaped circuits, finds the oracle's bytes, and replays the block DAG."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "cityprover_qbench")


def build_harness():
    sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
    from cityprover import build
    build.build()
    src = EXE + ".cpp"
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(src):
        tmp = "%s.%d.tmp" % (EXE, os.getpid())
        subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-L" + os.path.join(ROOT, "city-rollup_amd"),
                        "-lcityprover_hip", "-Wl,-rpath,$ORIGIN/../city-rollup_amd", "-lpthread", "-o", tmp], check=True)
        os.replace(tmp, EXE)


def test_harness_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    build_harness()
    r = subprocess.run([EXE, "--case", str(tmp_path / "missing.bin")], capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device" in r.stderr and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_harness_parity_and_dag_replay(tmp_path):
    build_harness()
    case = str(tmp_path / "case.bin")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dump_qbench_case.py"), case, "2"], check=True)
    for args, proofs in ((["--mode", "throughput", "--contexts", "2", "--batch", "4", "--iters", "2"], 16),
                         (["--mode", "dag", "--contexts", "2", "--batch", "8", "--blocks", "2"], 128)):
        r = subprocess.run([EXE, "--case", case] + args, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        out = json.loads(r.stdout.strip().splitlines()[-1])
        assert out["proofs"] == proofs and out["proof_bytes"] == 130576
        assert out["parity"].startswith("proof bytes == oracle bytes for all 2 circuits")
