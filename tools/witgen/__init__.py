"""ctypes loader for the bench-side witness helper (tools/witgen/witgen.hip)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libwitgen.so")
_lib = None


def build():
    """Compile libwitgen.so if missing or stale. Several ranks may get here at once: build to a private name, then
    rename atomically."""
    src = os.path.join(HERE, "witgen.hip")
    if not os.path.exists(SO) or os.path.getmtime(src) > os.path.getmtime(SO):
        tmp = "%s.%d.tmp" % (SO, os.getpid())
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC",
                        "-Wno-unused-value", "-o", tmp, src], check=True)
        os.replace(tmp, SO)
    return SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def poseidon_gate_rows(inputs, swaps):
    inputs = np.ascontiguousarray(inputs, dtype=np.uint64)
    swaps = np.ascontiguousarray(swaps, dtype=np.uint64)
    n = inputs.shape[0]
    out = np.zeros((n, 135), np.uint64)
    p = ctypes.POINTER(ctypes.c_uint64)
    lib().wg_poseidon_gate_rows(inputs.ctypes.data_as(p), swaps.ctypes.data_as(p), ctypes.c_size_t(n),
                                out.ctypes.data_as(p))
    return out
