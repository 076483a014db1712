"""merlin 3.0 transcripts (the reference's `ProtoTranscript`, distributed-prover/src/util.rs:22) and the reference's
`TranscriptProtocol` on top of them (util.rs:41-75): `append_serializable` = append_message(label, uncompressed
ark-serialize bytes), `challenge_scalar` = 32 challenge bytes -> ChaCha12Rng seed -> Fr::rand.

merlin (crate `merlin`, pinned "3.0.0" in distributed-prover/Cargo.toml:27) is not under /root/reference.  This is a
restatement of its published construction: STROBE-128 (v1.0.2, rate 166) over Keccak-f[1600], operations meta-AD / AD /
PRF, protocol label "Merlin v1.0", messages framed as meta-AD(label) meta-AD(le32 length) AD(message).  Pinned by
merlin's own known-answer test ("test protocol" / "some label" / "some data" -> 32 challenge bytes,
tests/test_merlin.py) and the Keccak permutation by hashlib's SHA3-256."""
import struct

from .chacha import ChaCha12Rng

_MASK = (1 << 64) - 1
_RC = []
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]


def _init_rc():
    r = 1
    for _ in range(24):
        rc = 0
        for j in range(7):
            r = ((r << 1) ^ ((r >> 7) * 0x71)) & 0xFF
            if r & 2:
                rc ^= 1 << ((1 << j) - 1)
        _RC.append(rc)


_init_rc()


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _MASK if n else x


_NATIVE = None


def _native():
    """libhekaton's host-side hk_keccak_f1600 (plain C; 100 x the speed of the Python permutation below), or False."""
    global _NATIVE
    if _NATIVE is None:
        try:
            import ctypes
            from . import capi
            fn = capi.load().hk_keccak_f1600
            fn.argtypes = [ctypes.c_void_p]
            fn.restype = None
            _NATIVE = (fn, ctypes)
        except Exception:       # noqa: BLE001  (library not built: the Python permutation is the same function)
            _NATIVE = False
    return _NATIVE


def keccak_f1600(state):
    """In-place Keccak-f[1600] on a 200-byte bytearray (lanes little-endian, lane (x, y) at 8 * (x + 5 y))."""
    nat = _native()
    if nat:
        fn, ctypes = nat
        buf = (ctypes.c_ubyte * 200).from_buffer(state)
        fn(ctypes.addressof(buf))
        return
    keccak_f1600_py(state)


def keccak_f1600_py(state):
    """The permutation in plain Python (reference form of the above; used when the library is not built)."""
    a = list(struct.unpack("<25Q", state))
    for rnd in range(24):
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        b = [0] * 25
        for x in range(5):
            for y in range(5):
                b[y + 5 * ((2 * x + 3 * y) % 5)] = _rol(a[x + 5 * y], _ROT[x][y])
        a = [b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & _MASK & b[(x + 2) % 5 + 5 * y]) for y in range(5) for x in range(5)]
        a[0] ^= _RC[rnd]
    state[:] = struct.pack("<25Q", *a)


STROBE_R = 166
FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32


class Strobe128:
    def __init__(self, protocol_label):
        st = bytearray(200)
        st[0:6] = bytes([1, STROBE_R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.state, self.pos, self.pos_begin, self.cur_flags = st, 0, 0, 0
        self.meta_ad(protocol_label, False)

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[STROBE_R + 1] ^= 0x80
        keccak_f1600(self.state)
        self.pos = self.pos_begin = 0

    def _absorb(self, data):
        data = bytes(data)
        off = 0
        while off < len(data):
            k = min(STROBE_R - self.pos, len(data) - off)
            chunk = int.from_bytes(data[off:off + k], "little") ^ int.from_bytes(self.state[self.pos:self.pos + k], "little")
            self.state[self.pos:self.pos + k] = chunk.to_bytes(k, "little")
            self.pos += k
            off += k
            if self.pos == STROBE_R:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray(n)
        for i in range(n):
            out[i] = self.state[self.pos]
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        assert not flags & FLAG_T
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if flags & (FLAG_C | FLAG_K) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(FLAG_M | FLAG_A, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(FLAG_A, more)
        self._absorb(data)

    def prf(self, n, more=False):
        self._begin_op(FLAG_I | FLAG_A | FLAG_C, more)
        return self._squeeze(n)


class Transcript:
    """merlin::Transcript: new / append_message / challenge_bytes, plus util.rs's TranscriptProtocol."""

    def __init__(self, label):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label, message):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(struct.pack("<I", len(message)), True)
        self.strobe.ad(bytes(message), False)

    def challenge_bytes(self, label, n):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(struct.pack("<I", n), True)
        return self.strobe.prf(n, False)

    def append_serializable(self, label, ser_bytes):
        """util.rs:56-65; the caller passes the value's uncompressed ark-serialize bytes."""
        self.append_message(label, ser_bytes)

    def challenge_scalar(self, label, r_mod):
        """util.rs:68-75: 32 challenge bytes seed a ChaCha12Rng whose first `Fr::rand` draw is the challenge (a
        canonical int mod `r_mod`, the scalar field's order)."""
        return ChaCha12Rng(self.challenge_bytes(label, 32)).fr(r_mod)
