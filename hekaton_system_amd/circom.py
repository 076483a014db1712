"""circom `.r1cs` / `.wtns` readers — the input path for real constraint matrices.

Mirrors the reference's `circom-compat` crate (circom-compat/src/lib.rs):
    R1CSFile::{new, write}, Header            lib.rs:18-250   (iden3 r1cs binary format, BN254 only)
    read_witness / write_witness              lib.rs:331-372  (the JSON-ish text witness)
    read_binary_wtns                          lib.rs:455-537  (snarkjs .wtns)
    R1CSFile::generate_constraints            lib.rs:374-420  (wire -> instance/witness variable mapping)
and adds `to_csr()`, which turns the constraints into the three `hk_csr` matrices of include/hekaton.h
with exactly the column numbering `generate_constraints` + `cs.to_matrices()` produce.
Coefficients are ark `F::deserialize_uncompressed` values: 32 bytes little-endian canonical.
"""
import io
import struct

import numpy as np

BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BN254_R_LE = bytes.fromhex("010000f093f5e1439170b97948e833285d588181b64550b829a031e1724e6430")
assert int.from_bytes(BN254_R_LE, "little") == BN254_R


class InvalidData(ValueError):
    """ark `SerializationError::IoError(ErrorKind::InvalidData, ..)`."""


class Header:                                         # lib.rs:188-250
    def __init__(self, field_size, prime_size, n_wires, n_pub_out, n_pub_in, n_prv_in, n_labels, n_constraints):
        self.field_size, self.prime_size = field_size, prime_size
        self.n_wires, self.n_pub_out, self.n_pub_in, self.n_prv_in = n_wires, n_pub_out, n_pub_in, n_prv_in
        self.n_labels, self.n_constraints = n_labels, n_constraints

    @classmethod
    def read(cls, f, size):
        (field_size,) = struct.unpack("<I", f.read(4))
        if field_size != 32:
            raise InvalidData("This parser only supports 32-byte fields")
        if size != 32 + field_size:
            raise InvalidData("Invalid header section size")
        prime = f.read(field_size)
        if prime != BN254_R_LE:
            raise InvalidData("This parser only supports bn256")
        n_wires, n_pub_out, n_pub_in, n_prv_in = struct.unpack("<IIII", f.read(16))
        (n_labels,) = struct.unpack("<Q", f.read(8))
        (n_constraints,) = struct.unpack("<I", f.read(4))
        return cls(field_size, prime, n_wires, n_pub_out, n_pub_in, n_prv_in, n_labels, n_constraints)

    def write(self, f):
        f.write(struct.pack("<I", self.field_size))
        f.write(self.prime_size)
        f.write(struct.pack("<IIII", self.n_wires, self.n_pub_out, self.n_pub_in, self.n_prv_in))
        f.write(struct.pack("<Q", self.n_labels))
        f.write(struct.pack("<I", self.n_constraints))


def _read_fr(f):
    b = f.read(32)
    if len(b) != 32:
        raise InvalidData("unexpected end of file")
    v = int.from_bytes(b, "little")
    if v >= BN254_R:
        raise InvalidData("field element not canonical")     # ark deserialize_uncompressed validates
    return v


def _read_constraint_vec(f):                          # lib.rs:252-263
    (n,) = struct.unpack("<I", f.read(4))
    out = []
    for _ in range(n):
        (idx,) = struct.unpack("<I", f.read(4))
        out.append((idx, _read_fr(f)))
    return out


def _write_constraint_vec(vec, f):                    # lib.rs:265-275
    f.write(struct.pack("<I", len(vec)))
    for idx, coeff in vec:
        f.write(struct.pack("<I", idx))
        f.write(int(coeff).to_bytes(32, "little"))


class R1CSFile:
    def __init__(self, version, header, constraints, wire_mapping=None, witness=None):
        self.version, self.header, self.constraints = version, header, constraints
        self.wire_mapping = wire_mapping if wire_mapping is not None else []
        self.witness = witness if witness is not None else []

    @classmethod
    def new(cls, data, read_wire_map=True):
        """lib.rs:32-124.  `data`: bytes or a seekable binary file."""
        f = io.BytesIO(data) if isinstance(data, (bytes, bytearray, memoryview)) else data
        if f.read(4) != b"r1cs":
            raise InvalidData("Invalid magic number")
        (version,) = struct.unpack("<I", f.read(4))
        if version != 1:
            raise InvalidData("Unsupported version")
        (num_sections,) = struct.unpack("<I", f.read(4))
        offsets, sizes = {}, {}
        for _ in range(num_sections):
            sec_type, sec_size = struct.unpack("<IQ", f.read(12))
            offsets[sec_type], sizes[sec_type] = f.tell(), sec_size
            f.seek(sec_size, io.SEEK_CUR)
        for t, what in ((1, "header"), (2, "constraint"), (3, "wire2label")):
            if t not in offsets:
                raise InvalidData("No section offset for %s type found" % what)
        f.seek(offsets[1])
        header = Header.read(f, sizes[1])
        f.seek(offsets[2])
        constraints = [(_read_constraint_vec(f), _read_constraint_vec(f), _read_constraint_vec(f))
                       for _ in range(header.n_constraints)]
        wire_mapping = []
        if read_wire_map:                              # lib.rs:277-299 read_map (the reference skips it)
            f.seek(offsets[3])
            if sizes[3] != header.n_wires * 8:
                raise InvalidData("Invalid map section size")
            wire_mapping = list(struct.unpack("<%dQ" % header.n_wires, f.read(8 * header.n_wires)))
            if wire_mapping and wire_mapping[0] != 0:
                raise InvalidData("Wire 0 should always be mapped to 0")
        return cls(version, header, constraints, wire_mapping)

    def write(self):
        """lib.rs:144-157: magic, version, 3 sections (header, constraints, wire map)."""
        out = io.BytesIO()
        out.write(b"r1cs")
        out.write(struct.pack("<II", 1, 3))

        def section(t, body):
            out.write(struct.pack("<IQ", t, len(body)))
            out.write(body)
        hb = io.BytesIO(); self.header.write(hb); section(1, hb.getvalue())
        cb = io.BytesIO()
        for a, b, c in self.constraints:
            _write_constraint_vec(a, cb); _write_constraint_vec(b, cb); _write_constraint_vec(c, cb)
        section(2, cb.getvalue())
        section(3, b"".join(struct.pack("<Q", v) for v in self.wire_mapping))
        return out.getvalue()

    # ---- matrices for the GPU path ---------------------------------------------------------------
    def num_inputs(self):
        return self.header.n_pub_in + self.header.n_pub_out           # lib.rs:376

    def to_csr(self, fr_codec, offset_instance=1):
        """CSR matrices (row_ptr u64, col u32, val Montgomery bytes) with the columns that
        `generate_constraints` (lib.rs:374-420) followed by `cs.to_matrices()` yields when the file is
        loaded into a constraint system that already holds `offset_instance` instance variables (the
        constant one): wire i < num_inputs -> instance column offset_instance + i, otherwise witness column
        (offset_instance + num_inputs) + (i - num_inputs)."""
        n_in = self.num_inputs()

        def col(i):
            return offset_instance + i        # instance block is followed directly by the witness block

        mats = []
        for k in range(3):
            row_ptr = np.zeros(len(self.constraints) + 1, dtype=np.uint64)
            cols, vals = [], []
            for r, con in enumerate(self.constraints):
                for idx, coeff in con[k]:
                    if idx >= self.header.n_wires:       # a wire the header does not declare: out-of-bounds column
                        raise InvalidData("constraint %d references wire %d of %d" % (r, idx, self.header.n_wires))
                    cols.append(col(idx))
                    vals.append(coeff)
                row_ptr[r + 1] = len(cols)
            mats.append((row_ptr, np.array(cols, dtype=np.uint32), fr_codec.enc(vals)))
        return mats, offset_instance + n_in, self.header.n_wires - n_in   # (A,B,C), n_inst, n_wit


def read_witness(text):                               # lib.rs:331-347
    out = []
    for line in text.splitlines()[1:]:
        if len(line) <= 1:
            continue
        out.append(int(line[2:len(line) - 1]) % BN254_R)
    return out


def write_witness(witness):                           # lib.rs:349-372
    lines = ["[", ' "%d"' % witness[0]]
    lines += [',"%d"' % v for v in witness[1:]]
    lines.append("]")
    return "\n".join(lines) + "\n"


def read_binary_wtns(data):                           # lib.rs:455-537
    f = io.BytesIO(data)
    if f.read(4) != b"wtns":
        raise InvalidData("Invalid magic number")
    (version,) = struct.unpack("<I", f.read(4))
    if version != 2:
        raise InvalidData("Unsupported version")
    (num_sections,) = struct.unpack("<I", f.read(4))
    offsets, sizes = {}, {}
    for _ in range(num_sections):
        sec_type, sec_size = struct.unpack("<IQ", f.read(12))
        offsets[sec_type], sizes[sec_type] = f.tell(), sec_size
        f.seek(sec_size, io.SEEK_CUR)
    if 1 not in offsets or 2 not in offsets:
        raise InvalidData("No section offset found")
    f.seek(offsets[1])
    (field_size,) = struct.unpack("<I", f.read(4))
    if field_size != 32:
        raise InvalidData("This parser only supports 32-byte fields")
    if sizes[1] != 8 + field_size:
        raise InvalidData("Invalid header section size")
    if f.read(32) != BN254_R_LE:
        raise InvalidData("This parser only supports bn256")
    (n_witness,) = struct.unpack("<I", f.read(4))
    f.seek(offsets[2])
    return [_read_fr(f) for _ in range(n_witness)]
