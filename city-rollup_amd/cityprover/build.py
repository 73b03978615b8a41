"""Build libcityprover_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU). The library is several
translation units (csrc/*.hip) compiled in parallel and linked into one shared object."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG, "csrc")
SO = os.path.join(PKG, "libcityprover_hip.so")
OBJ = os.path.join(PKG, "build")
# translation unit -> the headers it includes (core.h and its own includes are shared by all)
COMMON = ["core.h", "gl.h", "host_util.h", "dev_pool.h"]
UNITS = {
    "cityprover.hip": ["poseidon.h", "poseidon_coop.h", "poseidon_tables.h", "merkle.h", "ntt.h", "ntt16.h", "fri.h", "transcript.h", "zs.h", "quotient.h", "gates.h",
                       "prover_tail.inc", "fri_engine.inc", "fri_prove.inc", "batcher.inc", "verify.inc", "circuit_file.inc", "air.h", "ext3.h", "stark.inc"],
    "bls.hip": ["bls12_381.h", "bls12_381_tables.h", "msm.h", "msm.inc", "bls12_381_fr.h", "fr_ntt.h", "fr_ntt.inc", "groth16.inc",
                "groth16_pack.inc"],
}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-function"]


def _mtime(path):
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _unit_stale(unit):
    obj = os.path.join(OBJ, unit + ".o")
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, f) for f in [unit] + COMMON + UNITS[unit]]
    deps.append(os.path.join(os.path.dirname(PKG), "include", "cityprover.h"))
    return any(_mtime(d) > t for d in deps)


def stale():
    if not os.path.exists(SO):
        return True
    return any(_unit_stale(u) or _mtime(os.path.join(OBJ, u + ".o")) > os.path.getmtime(SO) for u in UNITS)


def build(force=False, verbose=False):
    if not os.path.exists(os.path.join(CSRC, "poseidon_tables.h")):
        subprocess.run([sys.executable, os.path.join(CSRC, "gen_tables.py")], check=True)
    if not force and not stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    pid = os.getpid()   # several ranks may build at once: private names, atomic renames

    def compile_unit(unit):
        obj = os.path.join(OBJ, unit + ".o")
        if not force and not _unit_stale(unit):
            return obj
        tmp = "%s.%d.tmp" % (obj, pid)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, unit), "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        os.replace(tmp, obj)
        return obj

    with ThreadPoolExecutor(len(UNITS)) as ex:
        objs = list(ex.map(compile_unit, UNITS))
    tmp = "%s.%d.tmp" % (SO, pid)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(tmp, SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
