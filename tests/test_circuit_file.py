"""The circuit file format of the import bridge (layout in csrc/circuit_file.inc): what a patched
plonky2 dumps once per built circuit and cp_circuit_load_file turns into a resident circuit. CPU part: the host-side
parser accepts what the writer writes and refuses every damaged file; GPU part: save -> load round trip, coefficient
form, and proof bytes of a loaded circuit == the oracle's."""
import os
import struct
import sys

import numpy as np
import pytest

import oracle_lib as O
from synth_circuit import build

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "city-rollup_amd"))


def shape_of(cp, c):
    from test_gpu_prove_full import cp_shape_of
    return cp_shape_of(cp, c["shape"], num_public_inputs=len(c["public_inputs"]))


def pi_targets(c):
    """(row, wire) of each public input of the synthetic circuit: PublicInputGate rows hold them 4 wires... the synthetic
    builder does not route them, so the test plants targets of its own and checks the read-back."""
    n = c["wires"].shape[1]
    return [(j % n, (3 * j + 1) % c["wires"].shape[0]) for j in range(len(c["public_inputs"]))]


def write(cp, path, c, digest, **kw):
    from cityprover import files
    return files.write_circuit_file(path, shape_of(cp, c), digest, c["gate_list"], 1, c["cs_values"], **kw)


def test_file_info_and_damaged_files(tmp_path):
    import cityprover as cp
    c = build(db=6, num_routed=16, num_wires=20, chunk=8, rate_bits=3, arity_bits=(2, 2), seed=5)
    path = str(tmp_path / "c.cpcirc")
    k_is = [pow(7, j, O.P) for j in range(16)]
    total = write(cp, path, c, [9, 8, 7, 6], k_is=k_is, pi_targets=pi_targets(c))
    assert os.path.getsize(path) == total
    info = cp.circuit_file_info(path)
    sh = shape_of(cp, c)
    assert info["digest"] == [9, 8, 7, 6] and info["n_gates"] == len(c["gate_list"]) and info["num_selectors"] == 1
    assert info["flags"] == 2 | 4
    for f, _ in sh._fields_:
        a, b = getattr(info["shape"], f), getattr(sh, f)
        assert (list(a) == list(b)) if f == "arity_bits" else (a == b), f
    good = open(path, "rb").read()

    def expect_refused(data, match):
        bad = str(tmp_path / "bad.cpcirc")
        open(bad, "wb").write(data)
        with pytest.raises(cp.CityProverError, match=match):
            cp.circuit_file_info(bad)

    expect_refused(b"CPCIRCv2" + good[8:], "magic")
    expect_refused(good[:8] + struct.pack("<I", 2) + good[12:], "version")
    expect_refused(good[:100], "truncated")
    expect_refused(good[:-16], "size|bytes")
    flipped = bytearray(good)
    flipped[len(good) // 2] ^= 1
    expect_refused(bytes(flipped), "checksum")
    expect_refused(good + b"\0" * 8, "size")
    expect_refused(good[:12] + struct.pack("<I", 0x80) + good[16:], "flags")
    # a non-canonical polynomial element, checksum repaired: refused on content
    from cityprover import files
    bad = bytearray(good)
    bad[-16:-8] = struct.pack("<Q", O.P)
    bad[-8:] = struct.pack("<Q", files.fnv1a64(bytes(bad[:-8])))
    expect_refused(bytes(bad), "canonical")
    # header fields overwritten with hostile values and the checksum REPAIRED, so the content checks alone stand between the
    # file and the loader: every such file is refused with an error or read consistently — no crash, no giant allocation
    import random
    rnd = random.Random(9)
    hdr = min(len(good) - 8, 512)
    refused = 0
    for _ in range(300):
        m = bytearray(good)
        for _ in range(rnd.randrange(1, 4)):
            o = 12 + 4 * rnd.randrange((hdr - 12) // 4)
            m[o:o + 4] = struct.pack("<I", rnd.choice([0, 1, 0x7FFFFFFF, 0xFFFFFFFF, 0x80000000, 1 << 20, 1 << 30, 65, 4097,
                                                       struct.unpack("<I", good[o:o + 4])[0] ^ (1 << rnd.randrange(32))]))
        m[-8:] = struct.pack("<Q", files.fnv1a64(bytes(m[:-8])))
        badp = str(tmp_path / "hostile.cpcirc")
        open(badp, "wb").write(bytes(m))
        try:
            got = cp.circuit_file_info(badp)
            assert got["n_gates"] <= 4096
        except cp.CityProverError:
            refused += 1
    assert refused >= 50   # (digest words, gate parameters and reserved words are free-form: overwriting them is not an error)
    with pytest.raises(cp.CityProverError, match="cannot open"):
        cp.circuit_file_info(str(tmp_path / "missing.cpcirc"))
    assert cp.circuit_file_info(path)["n_gates"] == len(c["gate_list"])   # the good file still reads


def test_witness_file_round_trip(tmp_path):
    from cityprover import files
    c = build(db=5, num_routed=16, num_wires=20, chunk=8, rate_bits=3, arity_bits=(2,), seed=1)
    path = str(tmp_path / "w.cpwit")
    files.write_witness_file(path, [1, 2, 3, 4], c["wires"], c["public_inputs"], proof=b"abcde")
    w = files.read_witness_file(path)
    assert w["digest"] == [1, 2, 3, 4] and (w["wires"] == c["wires"]).all() and w["proof"] == b"abcde"
    assert [int(x) for x in w["public_inputs"]] == list(c["public_inputs"])
    files.write_witness_file(path, [1, 2, 3, 4], c["wires"], c["public_inputs"])
    assert files.read_witness_file(path)["proof"] is None


@pytest.mark.gpu
def test_save_load_prove_parity(tmp_path):
    import cityprover as cp
    from cityprover import files
    p = cp.Prover(0)
    c = build(db=7, num_routed=24, num_wires=30, chunk=8, rate_bits=3, arity_bits=(2, 2), seed=77)
    digest = [5, 6, 7, 8]
    want, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    # (a) written by the host-side writer (what the Rust dumper does), values form
    f1 = str(tmp_path / "values.cpcirc")
    write(cp, f1, c, digest, pi_targets=pi_targets(c))
    c1 = cp.load_circuit_file(p, f1)
    assert cp.prove(c1, c["wires"], c["public_inputs"]) == want
    got_pi = cp.public_inputs_from_wires(c1, c["wires"])
    assert [int(x) for x in got_pi] == [int(c["wires"][w, r]) for r, w in pi_targets(c)]
    # (b) coefficient form (PolynomialBatch::polynomials): same circuit, same cap, same proof bytes
    f2 = str(tmp_path / "coeffs.cpcirc")
    coeffs = np.stack([O.intt(row) for row in c["cs_values"]])
    files.write_circuit_file(f2, shape_of(cp, c), digest, c["gate_list"], 1, coeffs, coeffs=True)
    c2 = cp.load_circuit_file(p, f2)
    assert (c2.cs_cap() == c1.cs_cap()).all()
    assert cp.prove(c2, c["wires"], c["public_inputs"]) == want
    with pytest.raises(cp.CityProverError, match="targets"):
        cp.public_inputs_from_wires(c2, c["wires"])
    # (c) saved by the library from a circuit built through the plain API: byte-identical to the writer's file
    sh = shape_of(cp, c)
    c3 = cp.Circuit(p, sh, digest, c["cs_values"])
    with pytest.raises(cp.CityProverError, match="gate set"):
        cp.save_circuit_file(c3, str(tmp_path / "nogates.cpcirc"))
    cp.set_gates(c3, c["gate_list"], 1)
    cp.set_public_input_targets(c3, pi_targets(c))
    f3 = str(tmp_path / "saved.cpcirc")
    cp.save_circuit_file(c3, f3)
    k_is = [pow(7, j, O.P) for j in range(sh.num_routed_wires)]
    f4 = str(tmp_path / "writer.cpcirc")
    write(cp, f4, c, digest, k_is=k_is, pi_targets=pi_targets(c))
    assert open(f3, "rb").read() == open(f4, "rb").read()
    c4 = cp.load_circuit_file(p, f3)
    assert cp.prove(c4, c["wires"], c["public_inputs"]) == want
    cp.verify(c4, want)
    # a corrupt file is refused by the loader too, and the context stays usable
    bad = bytearray(open(f3, "rb").read())
    bad[200] ^= 4
    open(str(tmp_path / "bad.cpcirc"), "wb").write(bad)
    with pytest.raises(cp.CityProverError, match="checksum"):
        cp.load_circuit_file(p, str(tmp_path / "bad.cpcirc"))
    assert cp.prove(c4, c["wires"], c["public_inputs"]) == want
    for x in (c1, c2, c3, c4):
        x.close()
    p.close()
