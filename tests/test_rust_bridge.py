"""The Rust side of the bridge (rust/) cannot be compiled here (no cargo); what can be checked is that it stays in step with
the C side: the generated `extern "C"` block is current and lists every exported symbol, the layout constants the Rust
writers use are the ones the library reads."""
import os
import re
import subprocess
import sys

import pytest

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


# ---- cp_gate_of: the parser of `Gate::id()` strings, mirrored in Python and fed the LITERAL ids --------------------------------
def _rust_gate_table():
    """name -> (type id, [field names of param, param2, param3]) read off rust/plonky2-hwa-patch/cityprover.rs"""
    rs = open(os.path.join(ROOT, "rust", "plonky2-hwa-patch", "cityprover.rs")).read()
    table = {}
    for name, ty, rest in re.findall(r'"(\w+Gate)" => \((\d+),([^\n]*)\),', rs):
        params = [p.strip() for p in rest.split(",")]
        assert len(params) == 3, (name, params)
        table[name] = (int(ty), [re.fullmatch(r'field\("(\w+)"\)', p).group(1) if p != "0" else None for p in params])
    return table


def cp_gate_of_py(gate_id, table):
    """the Rust parser, statement for statement: first token = the type name, `field(x)` = the digits after the first "x: " """
    name = re.split(r"[ <({]", gate_id, maxsplit=1)[0]

    def field(f):
        parts = gate_id.split(f + ": ")
        if len(parts) < 2:
            return 0
        m = re.match(r"\d+", parts[1])
        return int(m.group(0)) if m else 0
    ty, fields = table[name]
    return (ty,) + tuple(field(f) if f else 0 for f in fields)


def test_gate_id_strings_parse_to_the_right_type_and_parameters():
    """the id formats of the eight in-tree gates are reference-held — `fn id` = `format!("{self:?}")` (or with
    `<D={D}>` for ComparisonGate) over the struct definitions in city_common_circuit/src/u32/gates/*.rs (e.g. range_check_u32.rs:
    20-24,52-54; add_many_u32.rs:25-30,88-90; comparison.rs:28-33,97-99; interleave_u32.rs:32-35,85-87), i.e. Rust's derived Debug:
    `Name { field: value, .., _phantom: PhantomData<type> }`. The upstream ids follow plonky2 0.2.2's `id()` implementations
    (UPSTREAM-MEMORY). Each literal string must come out as (CP_GATE_* id, param, param2, param3)."""
    t = _rust_gate_table()
    hdr = open(os.path.join(ROOT, "include", "cityprover.h")).read()
    ids = {m.group(1): int(m.group(2)) for m in re.finditer(r"CP_GATE_(\w+) = (\d+)", hdr)}
    ph = "PhantomData<plonky2_field::goldilocks_field::GoldilocksField>"
    cases = [
        # in-tree (city_common_circuit/src/u32/gates): field order as declared
        ("U32AddManyGate { num_addends: 3, num_ops: 5, _phantom: %s }" % ph, ("U32_ADD_MANY", 5, 3, 0)),
        ("U32ArithmeticGate { num_ops: 3, _phantom: %s }" % ph, ("U32_ARITHMETIC", 3, 0, 0)),
        ("ComparisonGate { num_bits: 32, num_chunks: 16, _phantom: %s }<D=2>" % ph, ("COMPARISON", 32, 16, 0)),
        ("U32InterleaveGate { num_ops: 3 }", ("U32_INTERLEAVE", 3, 0, 0)),
        ("U32RangeCheckGate { num_input_limbs: 7, _phantom: %s }" % ph, ("U32_RANGE_CHECK", 7, 0, 0)),
        ("U32SubtractionGate { num_ops: 6, _phantom: %s }" % ph, ("U32_SUBTRACTION", 6, 0, 0)),
        ("UninterleaveToU32Gate { num_ops: 2 }", ("UNINTERLEAVE_TO_U32", 2, 0, 0)),
        ("UninterleaveToB32Gate { num_ops: 2 }", ("UNINTERLEAVE_TO_B32", 2, 0, 0)),
        # upstream plonky2 0.2.2 (the city-common set of builder/pad_circuit.rs:31-55, + Noop / PublicInput / Exponentiation)
        ("NoopGate", ("NOOP", 0, 0, 0)),
        ("ConstantGate { num_consts: 2 }", ("CONSTANT", 2, 0, 0)),
        ("PublicInputGate", ("PUBLIC_INPUT", 0, 0, 0)),
        ("ArithmeticGate { num_ops: 20 }", ("ARITHMETIC", 20, 0, 0)),
        ("PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>", ("POSEIDON", 0, 0, 0)),
        ("ArithmeticExtensionGate { num_ops: 10 }", ("ARITHMETIC_EXT", 10, 0, 0)),
        ("MulExtensionGate { num_ops: 13 }", ("MUL_EXT", 13, 0, 0)),
        ("BaseSumGate { num_limbs: 63 } + Base: 2", ("BASE_SUM", 63, 2, 0)),
        ("RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2, _phantom: %s }<D=2>" % ph, ("RANDOM_ACCESS", 4, 4, 2)),
        ("ReducingGate { num_coeffs: 43 }", ("REDUCING", 43, 0, 0)),
        ("ReducingExtensionGate { num_coeffs: 32 }", ("REDUCING_EXT", 32, 0, 0)),
        ("PoseidonMdsGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>", ("POSEIDON_MDS", 0, 0, 0)),
        ("CosetInterpolationGate { subgroup_bits: 4, degree: 6, barycentric_weights: [1, 2, 3], _phantom: %s }<D=2>" % ph,
         ("COSET_INTERPOLATION", 4, 6, 0)),
        ("ExponentiationGate { num_power_bits: 66, _phantom: %s }<D=2>" % ph, ("EXPONENTIATION", 66, 0, 0)),
    ]
    assert {c[1][0] for c in cases} == set(ids), "one literal id per gate type"
    for gate_id, (suffix, p1, p2, p3) in cases:
        assert cp_gate_of_py(gate_id, t) == (ids[suffix], p1, p2, p3), gate_id
    # traps the parser must not fall into: "bits: " also ends "num_bits: " / "subgroup_bits: " / "num_power_bits: " (the first
    # occurrence is taken — the gates that ask for "bits" have it first), "Base: " is not part of the type name
    assert cp_gate_of_py("ComparisonGate { num_bits: 10, num_chunks: 5, _phantom: x }<D=2>", t)[1:3] == (10, 5)
    assert cp_gate_of_py("BaseSumGate { num_limbs: 4 } + Base: 16", t)[1:3] == (4, 16)
    with pytest.raises(KeyError):
        cp_gate_of_py("LookupGate { num_slots: 40 }", t)     # the Rust side bails: the circuit stays on the CPU prover


def test_both_polynomial_batch_constructors_are_hooked_and_the_cpu_fallback_never_asserts():
    """a STARK prover commits traces with `from_values` and its quotient with `from_coeffs`; `prove_openings` runs on
    the device only when every oracle has a twin, so both constructors carry a hook, an oracle without a twin gets one on
    demand (`twin_on_demand`: coefficients + the salts of its host leaves — the C-ABI side of that is exercised on the GPU by
    tests/test_gpu_fri_generic.py::test_twin_on_demand_of_a_cpu_committed_oracle), and when the device path declines the hook
    rebuilds the host trees of device-committed oracles instead of asserting. Text-level: nothing here can be compiled."""
    patch = open(os.path.join(ROOT, "rust", "plonky2-hwa-patch", "hooks.patch")).read()
    rs = open(os.path.join(ROOT, "rust", "plonky2-hwa-patch", "cityprover.rs")).read()
    assert "batch_from_values_gpu(vals, rate_bits, blinding, cap_height)" in patch
    assert "batch_from_coeffs_gpu(polys, rate_bits, blinding, cap_height)" in patch
    body = patch[patch.index("pub fn prove_openings("):]
    assert "assert!(oracles.iter()" not in body and "with_host_tree" in body
    for fn in ("pub fn batch_from_coeffs_gpu", "fn twin_on_demand", "pub fn with_host_tree", "fn batch_commit_gpu"):
        assert fn in rs, fn
    # the mirror takes coefficients through cp_batch_coeffs: cp_batch_device_ptrs would take the handle out of buffer recycling
    assert "cp_batch_device_ptrs" not in rs
    lib_rs = open(os.path.join(ROOT, "rust", "cityprover-sys", "src", "lib.rs")).read()
    assert "ffi::cp_batch_coeffs" in lib_rs


def test_recording_parser_speaks_the_header_op_codes():
    """rust/starkyx-patch/recording_parser.rs (uncompiled): every parser call it records maps to a CP_AIR_* code of the header, the
    op / descriptor layouts it fills are the generated ones, and the safe wrapper it calls exists in cityprover-sys."""
    rs = open(os.path.join(ROOT, "rust", "starkyx-patch", "recording_parser.rs")).read()
    ffi = open(os.path.join(ROOT, "rust", "cityprover-sys", "src", "ffi.rs")).read()
    lib_rs = open(os.path.join(ROOT, "rust", "cityprover-sys", "src", "lib.rs")).read()
    used = set(re.findall(r"ffi::(CP_AIR_\w+)", rs))
    assert used >= {"CP_AIR_LOCAL", "CP_AIR_NEXT", "CP_AIR_PUBLIC", "CP_AIR_GLOBAL", "CP_AIR_CHALLENGE", "CP_AIR_CONST", "CP_AIR_ADD", "CP_AIR_SUB",
                    "CP_AIR_MUL", "CP_AIR_NEG", "CP_AIR_ASSERT_ZERO", "CP_AIR_ASSERT_ZERO_TRANSITION", "CP_AIR_ASSERT_ZERO_FIRST_ROW",
                    "CP_AIR_ASSERT_ZERO_LAST_ROW", "CP_AIR_CONSTRAINTS"}
    for name in used:
        assert re.search(r"pub const %s: c_int = \d+;" % name, ffi), name
    # the values are the header's (the Python recorder of tests/air_programs.py uses the same numbers and is run on the device)
    hdr = open(os.path.join(ROOT, "include", "cityprover.h")).read()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import air_programs as A
    for name, val in (("CP_AIR_LOCAL", A.LOCAL), ("CP_AIR_MUL", A.MUL), ("CP_AIR_NEG", A.NEG), ("CP_AIR_INV", A.INV), ("CP_AIR_ASSERT_ZERO_LAST_ROW", A.ASSERT_ZERO_LAST_ROW),
                      ("CP_AIR_STORE", A.STORE)):
        assert re.search(r"%s = %d\b" % (name, val), hdr), name
        assert re.search(r"pub const %s: c_int = %d;" % (name, val), ffi), name
    for field in ("kind", "ops", "n_ops", "consts", "n_consts", "n_columns", "n_public", "n_global", "n_challenge", "n_out_columns"):
        assert re.search(r"\b%s:" % field, rs) and re.search(r"pub %s:" % field, ffi), field
    assert "CpAirOp { op: op as u32, a, b, c: 0 }" in rs and "pub struct CpAirOp" in ffi
    for fn in ("pub fn air_quotient_commit", "impl AirProgram", "ffi::cp_air_program_create", "ffi::cp_air_quotient_commit"):
        assert fn in lib_rs, fn
