"""CPU-side checks of the drop-in boundary: the HIP library builds, loads, and exports every
symbol include/cityprover.h declares. No compute calls (there is no GPU here)."""
import os
import re

import pytest

import cityprover

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "cityprover.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = cityprover.load_library()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/cityprover.h but not exported"
    # and the python binding table covers the header exactly
    assert sorted(cityprover.ABI.keys()) == syms


def test_abi_version_and_no_gpu_fails_loudly():
    lib = cityprover.load_library()
    assert lib.cp_abi_version() == 4
    if lib.cp_device_count() == 0:
        assert not lib.cp_ctx_create(0)
        assert b"no HIP device" in lib.cp_last_error(None)
        with pytest.raises(cityprover.CityProverError):
            cityprover.Prover()


def test_batcher_entry_points_reject_null_without_a_gpu():
    """cp_batcher_*: no context, no batcher — a status and a message, never a crash."""
    import ctypes
    lib = cityprover.load_library()
    assert not lib.cp_batcher_create(None, 8, 0)
    assert b"ctx is NULL" in lib.cp_last_error(None)
    out, n = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t()
    assert lib.cp_batcher_prove(None, None, None, None, 0, 0, 0, ctypes.byref(out), ctypes.byref(n)) != 0
    assert b"NULL" in lib.cp_last_error(None)
    assert lib.cp_batcher_get_stats(None, None) != 0
    lib.cp_batcher_destroy(None)


def test_every_entry_point_is_a_function_try_block():
    """Nothing may unwind through the C ABI into the host (Rust) process: every exported function that does more than read a
    field is a function-try-block (`int cp_x(...) try { ... } CP_CATCH(ctx)`). The exceptions are listed and trivial."""
    import glob
    src = "\n".join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "city-rollup_amd", "csrc", "*.hip")) +
                                                    glob.glob(os.path.join(ROOT, "city-rollup_amd", "csrc", "*.inc"))))
    trivial = {"cp_abi_version", "cp_device_count", "cp_last_error", "cp_free", "cp_fault_inject",
               "cp_ctx_destroy", "cp_circuit_destroy", "cp_batcher_destroy", "cp_batch_destroy", "cp_air_program_destroy"}   # getters, free(), destructors (void, try inside)
    without = set()
    for sym in header_symbols():
        m = re.search(r"^(?:extern \"C\" )?[A-Za-z_][\w \*]*\b" + sym + r"\s*\(([^;{]*?)\)\s*(try\s*)?\{", src, flags=re.M | re.S)
        assert m, f"{sym}: definition not found in csrc"
        if not m.group(2):
            without.add(sym)
    assert without == trivial, sorted(without ^ trivial)


def test_product_does_not_reference_oracle():
    # the product path must never import / link / call the oracle
    pkg = os.path.join(ROOT, "city-rollup_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "libcityoracle" not in txt and "oracle_lib" not in txt, os.path.join(dp, f)
                assert not re.search(r"#include\s+\"[^\"]*oracle", txt), os.path.join(dp, f)
