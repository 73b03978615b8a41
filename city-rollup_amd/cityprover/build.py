"""Build libcityprover_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG, "csrc")
SO = os.path.join(PKG, "libcityprover_hip.so")
SOURCES = ["cityprover.hip"]
HEADERS = ["gl.h", "poseidon.h", "poseidon_tables.h", "merkle.h", "ntt.h", "ntt16.h", "fri.h", "zs.h", "quotient.h", "gates.h", "prover_tail.inc", "verify.inc", "bls12_381.h", "bls12_381_tables.h", "msm.h", "msm.inc", "bls12_381_fr.h", "fr_ntt.h", "fr_ntt.inc", "groth16.inc"]


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps.append(os.path.join(os.path.dirname(PKG), "include", "cityprover.h"))
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not os.path.exists(os.path.join(CSRC, "poseidon_tables.h")):
        subprocess.run([sys.executable, os.path.join(CSRC, "gen_tables.py")], check=True)
    if not force and not stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = "%s.%d.tmp" % (SO, os.getpid())   # several ranks may build at once: private name, atomic rename
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-value", "-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(tmp, SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
