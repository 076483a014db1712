"""CPU: circom `.r1cs` / `.wtns` readers against the reference's OWN known-answer vectors
(circom-compat/src/lib.rs:548-737 — the only golden vectors in the reference repository; fixtures extracted
by tests/golden/gen_circom_fixtures.py).  The assertions are the reference's assertions."""
import os

import pytest

from hekaton_system_amd import circom
from hekaton_system_amd.cp_groth16 import FrCodec

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _data(name, mode="rb"):
    with open(os.path.join(G, name), mode) as f:
        return f.read()


def test_sample():                       # lib.rs:548-645 `sample`
    file = circom.R1CSFile.new(_data("circom_sample.r1cs"))
    assert file.version == 1
    assert file.header.field_size == 32
    assert file.header.prime_size == bytes.fromhex("010000f093f5e1439170b97948e833285d588181b64550b829a031e1724e6430")
    assert file.header.n_wires == 7
    assert file.header.n_pub_out == 1
    assert file.header.n_pub_in == 2
    assert file.header.n_prv_in == 3
    assert file.header.n_labels == 0x03E8
    assert file.header.n_constraints == 3
    assert len(file.constraints) == 3
    assert len(file.constraints[0][0]) == 2
    assert file.constraints[0][0][0] == (5, 3)
    assert file.constraints[2][1][0] == (0, 6)
    assert len(file.constraints[1][2]) == 0
    assert len(file.wire_mapping) == 7           # (commented out in the reference's reader; the spec'd map)
    assert file.wire_mapping[1] == 3
    witness = circom.read_witness(_data("circom_sample_witness.txt", "r"))
    assert len(witness) == 5
    assert witness[0] == 1
    assert witness[4] == 0


def test_write():                        # lib.rs:647-721 `test_write`
    data = _data("circom_sample.r1cs")
    file = circom.R1CSFile.new(data)
    assert file.write() == data
    wit_text = _data("circom_sample_witness.txt", "r")
    assert circom.write_witness(circom.read_witness(wit_text)) == wit_text


def test_wtns_bin_file():                # lib.rs:723-737 `wtns_bin_file`
    assert circom.read_binary_wtns(_data("circom_sample.wtns")) == [1]


def test_rejects_what_the_reference_rejects():
    data = bytearray(_data("circom_sample.r1cs"))
    with pytest.raises(circom.InvalidData):
        circom.R1CSFile.new(b"r1cX" + bytes(data[4:]))                 # magic
    bad = bytearray(data); bad[4] = 2
    with pytest.raises(circom.InvalidData):
        circom.R1CSFile.new(bytes(bad))                                # version
    bad = bytearray(data); bad[28] ^= 0xFF                             # first byte of the prime
    with pytest.raises(circom.InvalidData):
        circom.R1CSFile.new(bytes(bad))                                # "only supports bn256"


def test_to_csr_column_mapping():
    """generate_constraints (lib.rs:374-420): wire i < n_pub_in + n_pub_out -> instance, else witness; with
    the constant-one instance variable in front every wire index shifts by one column."""
    file = circom.R1CSFile.new(_data("circom_sample.r1cs"))
    (A, B, C), n_inst, n_wit = file.to_csr(FrCodec("bn254"))
    assert (n_inst, n_wit) == (1 + 3, 7 - 3)
    assert list(A[0]) == [0, 2, 5, 6] and list(B[0]) == [0, 3, 5, 8] and list(C[0]) == [0, 2, 2, 3]
    assert list(A[1][:2]) == [5 + 1, 6 + 1]
    fc = FrCodec("bn254")
    assert fc.dec(A[2][:64]) == [3, 8]


def test_wire_index_beyond_header_is_invalid_data():
    """A constraint that names a wire >= n_wires (truncated / malformed file) must be InvalidData at to_csr time, not
    an out-of-bounds column handed to the GPU (hk_pk_upload would answer HK_ERR_ARG; the parser says it first)."""
    file = circom.R1CSFile.new(_data("circom_sample.r1cs"))
    a, b, c = file.constraints[0]
    file.constraints[0] = ([(file.header.n_wires, 1)] + list(a[1:]), b, c)
    with pytest.raises(circom.InvalidData):
        file.to_csr(FrCodec("bn254"))
