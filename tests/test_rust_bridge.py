"""The Rust side of the bridge (rust/) cannot be compiled here (no cargo); what can be checked is that it stays in step with
the C side: the generated `extern "C"` block is current and lists every exported symbol, the layout constants the Rust
writers use are the ones the library reads."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ffi_block_is_generated_from_the_header_and_complete():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_ffi
    text, names = gen_rust_ffi.generate()
    assert open(gen_rust_ffi.OUT).read() == text, "rust/cityprover-sys/src/ffi.rs is stale: run tools/gen_rust_ffi.py"
    sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
    import cityprover
    cityprover.load_library()
    assert sorted(names) == sorted(cityprover.ABI)          # == the ctypes table == the header == the library's exports
    so = os.path.join(ROOT, "city-rollup_amd", "libcityprover_hip.so")
    exported = {l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout.splitlines()
                if " T cp_" in l}
    assert exported == set(names)


def test_rust_writers_use_the_library_layout():
    rs = open(os.path.join(ROOT, "rust", "plonky2-hwa-patch", "cityprover.rs")).read()
    inc = open(os.path.join(ROOT, "city-rollup_amd", "csrc", "circuit_file.inc")).read()
    assert 'b"CPCIRCv1"' in rs and '"CPCIRCv1"' in inc
    assert 'b"CPWITNv1"' in rs and '"CPWITNv1"' in open(os.path.join(ROOT, "tools", "qbench", "pack.h")).read()
    assert "0xcbf2_9ce4_8422_2325" in rs and "0xcbf29ce484222325" in inc          # FNV-1a offset basis
    assert "0x0000_0100_0000_01b3" in rs and "0x100000001b3" in inc              # FNV-1a prime
    # every gate name the Rust mapper knows maps to the id the header assigns
    hdr = open(os.path.join(ROOT, "include", "cityprover.h")).read()
    ids = {m.group(1): int(m.group(2)) for m in re.finditer(r"(CP_GATE_\w+) = (\d+)", hdr)}
    rust = {m.group(1): int(m.group(2)) for m in re.finditer(r'"(\w+Gate)" => \((\d+),', rs)}
    want = {"NoopGate": "NOOP", "ConstantGate": "CONSTANT", "PublicInputGate": "PUBLIC_INPUT", "ArithmeticGate": "ARITHMETIC",
            "PoseidonGate": "POSEIDON", "ComparisonGate": "COMPARISON", "U32ArithmeticGate": "U32_ARITHMETIC",
            "U32RangeCheckGate": "U32_RANGE_CHECK", "U32AddManyGate": "U32_ADD_MANY", "U32SubtractionGate": "U32_SUBTRACTION",
            "U32InterleaveGate": "U32_INTERLEAVE", "UninterleaveToU32Gate": "UNINTERLEAVE_TO_U32",
            "UninterleaveToB32Gate": "UNINTERLEAVE_TO_B32", "ArithmeticExtensionGate": "ARITHMETIC_EXT", "MulExtensionGate": "MUL_EXT",
            "BaseSumGate": "BASE_SUM", "RandomAccessGate": "RANDOM_ACCESS", "ReducingGate": "REDUCING",
            "ReducingExtensionGate": "REDUCING_EXT", "PoseidonMdsGate": "POSEIDON_MDS", "CosetInterpolationGate": "COSET_INTERPOLATION",
            "ExponentiationGate": "EXPONENTIATION"}
    assert set(rust) == set(want)
    for name, suffix in want.items():
        assert rust[name] == ids["CP_GATE_" + suffix], name
