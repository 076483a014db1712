"""Host mirror of the native (non-gadget) half of the reference's portal-transcript module - what the coordinator
computes between the two rounds of a job from the stage-0 commitments (distributed-prover/src/transcript/mod.rs,
rom_transcript.rs, ram_transcript.rs; coordinator.rs:92-160):

    RomTranscriptEntry / RamTranscriptEntry     entries of the portal trace, `padding()`, `to_field_elements()`, wire bytes
    RunningEvaluation.new(mem_type, super_com)  the challenges: SHA-256(context string || serialized commitment),
                                                little-endian mod r (rom_transcript.rs:42-75, ram_transcript.rs:50-98)
    update_time_ordered / update_addr_ordered   eval *= tr_chal - repr(entry)       (rom :78-107, ram :101-135)
    sort_subtraces_by_addr                      flatten, stable sort by addr (ROM) / (addr, timestamp) (RAM), re-chunk
    running_evaluations                         the evals after every subcircuit = the leaves of `generate_exec_tree`
                                                (coordinator.rs:125-160) without its Merkle tree (ark-crypto-primitives
                                                `TreeConfig`, third-party, out of scope)

Plain Python ints; field arithmetic is a handful of products per entry.  The ark-serialize layouts are those the derives
produce (field order of the structs; `RunningEvaluation` / `TranscriptEntry` enums carry a one-byte tag, mod.rs:47-66,
162-180); `Unsigned32` is `{ bits: Vec<bool> }` (uint32.rs:19-22): a u64 length (32) then one byte per bit, least
significant first."""
import hashlib
from dataclasses import dataclass

ROM, RAM = "rom", "ram"


def _chal(tag, com_bytes, r):
    return int.from_bytes(hashlib.sha256(tag + com_bytes).digest(), "little") % r


@dataclass(frozen=True)
class RomTranscriptEntry:                      # rom_transcript.rs:222-237
    addr: int
    val: int

    @staticmethod
    def padding():
        return RomTranscriptEntry(0, 0)

    def to_field_elements(self):               # rom_transcript.rs:110-114
        return [self.addr, self.val]

    def repr(self, chal):                      # rom_transcript.rs:84-86
        return self.val + chal[0] * self.addr

    def sort_key(self):
        return self.addr

    def to_wire(self, nb):
        return self.addr.to_bytes(8, "little") + self.val.to_bytes(nb, "little")

    @staticmethod
    def from_wire(buf, off, nb):
        return RomTranscriptEntry(int.from_bytes(buf[off:off + 8], "little"),
                                  int.from_bytes(buf[off + 8:off + 8 + nb], "little")), off + 8 + nb


@dataclass(frozen=True)
class RamTranscriptEntry:                      # ram_transcript.rs:261-278
    addr: int
    val: int
    i: int                                     # Unsigned32 timestamp
    read: bool

    @staticmethod
    def padding():
        return RamTranscriptEntry(0, 0, 0, False)

    def to_field_elements(self):               # ram_transcript.rs:280-289
        return [self.addr, self.val, self.i, int(self.read)]

    def repr(self, chal):                      # ram_transcript.rs:109-112
        return self.val + chal[0] * self.addr + chal[1] * self.i + chal[2] * int(self.read)

    def sort_key(self):                        # coordinator.rs:107 (addr, timestamp)
        return (self.addr, self.i)

    def to_wire(self, nb):
        bits = (32).to_bytes(8, "little") + bytes((self.i >> k) & 1 for k in range(32))
        return self.addr.to_bytes(8, "little") + self.val.to_bytes(nb, "little") + bits + bytes([int(self.read)])

    @staticmethod
    def from_wire(buf, off, nb):
        addr = int.from_bytes(buf[off:off + 8], "little")
        val = int.from_bytes(buf[off + 8:off + 8 + nb], "little")
        off += 8 + nb
        n_bits = int.from_bytes(buf[off:off + 8], "little")
        if n_bits != 32 or len(buf) < off + 8 + 33:
            raise ValueError("InvalidData: Unsigned32 is 32 bools")
        bits = buf[off + 8:off + 40]
        if any(b > 1 for b in bits) or buf[off + 40] > 1:
            raise ValueError("InvalidData: bool byte > 1")
        i = sum(int(b) << k for k, b in enumerate(bits))
        return RamTranscriptEntry(addr, val, i, bool(buf[off + 40])), off + 41


class RunningEvaluation:
    """mod.rs:69-160.  `challenges` = (entry_chal, tr_chal) for ROM, (entry_chal_1, entry_chal_2, entry_chal_3, tr_chal)
    for RAM - the reference's order (`challenges()`, mod.rs:148-159)."""

    def __init__(self, mem_type, r, challenges=None, time_ordered_eval=1, addr_ordered_eval=1):
        assert mem_type in (ROM, RAM)
        self.mem_type, self.r = mem_type, r
        self.challenges = None if challenges is None else tuple(c % r for c in challenges)
        self.time_ordered_eval, self.addr_ordered_eval = time_ordered_eval % r, addr_ordered_eval % r

    @staticmethod
    def new(mem_type, super_com, r):
        """Hash the trace commitment to the challenges.  super_com: aggregation.IppCom (or its uncompressed bytes)."""
        com_bytes = super_com if isinstance(super_com, (bytes, bytearray)) else super_com.serialize_uncompressed()
        tags = (b"entry_chal", b"tr_chal") if mem_type == ROM else (b"entry_chal_1", b"entry_chal_2", b"entry_chal_3", b"tr_chal")
        return RunningEvaluation(mem_type, r, [_chal(t, bytes(com_bytes), r) for t in tags])

    def copy(self):
        return RunningEvaluation(self.mem_type, self.r, self.challenges, self.time_ordered_eval, self.addr_ordered_eval)

    def copy_challenges_from(self, other):     # mod.rs:134-145
        if other.mem_type != self.mem_type:
            raise TypeError("Invalid entry type")
        self.challenges = other.challenges

    def _factor(self, entry):
        if self.challenges is None:
            raise RuntimeError("RunningEvals.challenges needs to be set in order to run update")
        want = RomTranscriptEntry if self.mem_type == ROM else RamTranscriptEntry
        if not isinstance(entry, want):
            raise TypeError("Invalid entry type")                          # mod.rs:97,101
        return (self.challenges[-1] - entry.repr(self.challenges)) % self.r

    def update_time_ordered(self, entry):
        self.time_ordered_eval = self.time_ordered_eval * self._factor(entry) % self.r

    def update_addr_ordered(self, entry):
        self.addr_ordered_eval = self.addr_ordered_eval * self._factor(entry) % self.r

    def to_wire(self, nb):
        """tag byte, the two evals, Option<challenges> (one presence byte)."""
        out = bytes([0 if self.mem_type == ROM else 1])
        out += self.time_ordered_eval.to_bytes(nb, "little") + self.addr_ordered_eval.to_bytes(nb, "little")
        if self.challenges is None:
            return out + b"\x00"
        return out + b"\x01" + b"".join(c.to_bytes(nb, "little") for c in self.challenges)

    @staticmethod
    def from_wire(buf, off, nb, r):
        tag = buf[off]
        if tag > 1:
            raise ValueError("InvalidData: RunningEvaluation tag")
        mem = ROM if tag == 0 else RAM
        off += 1
        t = int.from_bytes(buf[off:off + nb], "little")
        a = int.from_bytes(buf[off + nb:off + 2 * nb], "little")
        off += 2 * nb
        some = buf[off]
        off += 1
        ch = None
        if some > 1:
            raise ValueError("InvalidData: Option tag")
        if some:
            k = 2 if mem == ROM else 4
            ch = [int.from_bytes(buf[off + j * nb:off + (j + 1) * nb], "little") for j in range(k)]
            off += k * nb
        if t >= r or a >= r or (ch and any(c >= r for c in ch)):
            raise ValueError("InvalidData: field element not reduced")
        return RunningEvaluation(mem, r, ch, t, a), off


def sort_subtraces_by_addr(time_ordered_subtraces):
    """coordinator.rs:92-123: Rust's `sort_by_key` is stable, so is `sorted`."""
    flat = [e for st in time_ordered_subtraces for e in st]
    flat = sorted(flat, key=lambda e: e.sort_key())
    out, pos = [], 0
    for st in time_ordered_subtraces:
        out.append(flat[pos:pos + len(st)])
        pos += len(st)
    return out


def running_evaluations(mem_type, super_com, r, time_ordered_subtraces, addr_ordered_subtraces):
    """coordinator.rs:137-160: [(RunningEvaluation after subcircuit i, last entry of its address-ordered subtrace)]."""
    evals = RunningEvaluation.new(mem_type, super_com, r)
    last = (RomTranscriptEntry if mem_type == ROM else RamTranscriptEntry).padding()
    leaves = []
    for time_st, addr_st in zip(time_ordered_subtraces, addr_ordered_subtraces):
        for te, ae in zip(time_st, addr_st):
            evals.update_time_ordered(te)
            evals.update_addr_ordered(ae)
            last = ae
        leaves.append((evals.copy(), last))
    return leaves
