"""Wire / disk formats either side of the hot path: ark-serialize 0.4 framing of the
records the reference moves between coordinator, workers and the key file, over the ABI's packed-affine
Montgomery arrays.

What the reference serialises (everything `serialize_uncompressed` / `deserialize_uncompressed_unchecked`):
    Stage0Response / Stage1Response         distributed-prover/src/worker.rs:20-52 (104 B / 336 B on BN254)
    Proof, VerifyingKey, ProvingKey, CommitterKey   cp-groth16/src/data_structures.rs:6-16,32-46,65-83,107-114
    ProvingKeys (the key file)              mpi-snark/src/data_structures.rs:41-51 (derive order) and :112-128
    Packed (256-byte MPI framing)           mpi-snark/src/lib.rs:68-111
    final proof size, compressed            mpi-snark/src/bin/node.rs:611-616
and the commitment randomness the worker re-derives from `com_seed` (distributed-prover/src/worker.rs:129-137,
cp-groth16/src/committer.rs:85): `Fr::rand(ChaCha12Rng::from_seed(com_seed))`.

The formats live in third-party crates that are absent from /root/reference (ark-serialize / ark-ec / ark-ff
^0.4, ark-bls12-381 ^0.4, rand_chacha 0.3.1); they are restated here from their published definitions:
  * usize -> u64 LE; Vec<T> / String / BTreeMap -> u64 LE length then the items; [u8; N] -> raw bytes;
  * Fp -> canonical (non-Montgomery) little-endian bytes;
  * short-Weierstrass affine (BN254): uncompressed x || y, compressed x; flags in the two top bits of the LAST
    byte (bit 7: y > -y, bit 6: infinity; infinity has x = y = 0); Fp2 = c0 || c1, flags on c1; Fp2 order
    compares c1 first;
  * ark-bls12-381 overrides points with the zcash format: big-endian, Fp2 = c1 || c0, flags in the three top
    bits of the FIRST byte (bit 7 compressed, bit 6 infinity, bit 5 y lexicographically largest — compressed only).
PARITY UNPINNED by reference bytes: the reference holds no serialised fixture. What pins this file instead:
published generator encodings (tests/test_ark_serialize.py), the record sizes SURVEY.md §8(a10) derives from the
reference's types, RFC 7539 / eSTREAM ChaCha keystream vectors, and round trips.

Bulk Montgomery <-> canonical conversion runs on the device (`hk_field_convert`) when a `capi.Context` is given;
without one the (slow) Python big-int path is used — fine for responses and proofs, not for key files.
"""
import struct

import numpy as np

from .cp_groth16 import CURVE_PARAMS, CommitterKey, Proof, ProvingKey, VerifyingKey


class SerializationError(Exception):
    """ark_serialize::SerializationError (InvalidData, UnexpectedFlags, IoError)."""


# --------------------------------------------------------------------------------------- byte cursor
class Reader:
    def __init__(self, buf, off=0):
        self.buf = memoryview(buf).cast("B") if not isinstance(buf, memoryview) else buf
        self.off = off

    def take(self, n):
        if self.off + n > len(self.buf):
            raise SerializationError("IoError: unexpected end of input")
        v = self.buf[self.off:self.off + n]
        self.off += n
        return v

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def remaining(self):
        return len(self.buf) - self.off


class Writer:
    def __init__(self):
        self.parts = []

    def put(self, b):
        self.parts.append(bytes(b) if not isinstance(b, (bytes, bytearray)) else b)

    def u64(self, v):
        self.parts.append(struct.pack("<Q", v))

    def getvalue(self):
        return b"".join(self.parts)


def write_bytes_vec(w, b):            # Vec<u8> / String
    w.u64(len(b))
    w.put(b)


def read_bytes_vec(r):
    return bytes(r.take(r.u64()))


def write_usize_map(w, d):            # BTreeMap<usize, usize>: sorted by key
    w.u64(len(d))
    for k in sorted(d):
        w.u64(k)
        w.u64(d[k])


def read_usize_map(r):
    return {r.u64(): r.u64() for _ in range(r.u64())}


PACKED_BYTE_SIZE = 256                # mpi-snark/src/lib.rs:82-84


def to_packed(b):
    """serialize_to_packed_vec (mpi-snark/src/lib.rs:74-79): zero-pad to a whole number of 256-byte `Packed`s."""
    pad = (-len(b)) % PACKED_BYTE_SIZE
    return bytes(b) + b"\0" * pad


# --------------------------------------------------------------------------------------- the codec
class ArkCodec:
    """Converts between the C ABI's arrays (Montgomery, x || y, infinity = zeros) and ark-serialize bytes."""

    def __init__(self, curve, ctx=None):
        p = CURVE_PARAMS[curve]
        self.curve = curve
        self.ctx = ctx
        self.r, self.q = p["r"], p["q"]
        self.frb, self.fqb = p["fr_bytes"], p["fq_bytes"]
        self.g1b, self.g2b = 2 * self.fqb, 4 * self.fqb
        self.zcash = curve == "bls12_381"
        self.Rq = 1 << (8 * self.fqb)
        self.Rr = 1 << (8 * self.frb)
        self.half = (self.q - 1) // 2
        self._half_limbs = np.array([(self.half >> (64 * i)) & (2 ** 64 - 1) for i in range(self.fqb // 8)], dtype=np.uint64)

    # ---- Montgomery <-> canonical, packed little-endian arrays of field elements ----
    def _convert(self, which, arr, to_mont):
        arr = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1)
        if arr.size == 0:
            return arr.copy()
        if self.ctx is not None:
            return self.ctx.field_convert(which, arr, to_mont)
        nb, p = (self.frb, self.r) if which == 0 else (self.fqb, self.q)
        R = 1 << (8 * nb)
        f = R % p if to_mont else pow(R, -1, p)
        b = arr.tobytes()
        out = bytearray(len(b))
        for i in range(0, len(b), nb):
            x = int.from_bytes(b[i:i + nb], "little")
            if x >= p:
                raise SerializationError("InvalidData: field element not reduced")
            out[i:i + nb] = (x * f % p).to_bytes(nb, "little")
        return np.frombuffer(bytes(out), dtype=np.uint8).copy()

    # ---- scalars ----
    def fr_to_wire(self, mont):
        """Fr elements, ABI (Montgomery) -> canonical LE bytes (ark-ff `serialize_with_flags`, no flags)."""
        return self._convert(0, mont, False).tobytes()

    def fr_from_wire(self, buf, n=None):
        a = np.frombuffer(bytes(buf), dtype=np.uint8)
        if n is not None and a.size != n * self.frb:
            raise SerializationError("IoError: wrong length")
        self._check_reduced(a, self.frb, self.r)
        return self._convert(0, a, True)

    def _check_reduced(self, canon, nb, p):
        """deserialize rejects a non-canonical field element (InvalidData)."""
        a = canon.reshape(-1, nb // 8, 8).view(np.uint64).reshape(-1, nb // 8) if canon.size else None
        if a is None:
            return
        pl = [(p >> (64 * i)) & (2 ** 64 - 1) for i in range(nb // 8)]
        lt = np.zeros(a.shape[0], dtype=bool)
        eq = np.ones(a.shape[0], dtype=bool)
        for i in range(nb // 8 - 1, -1, -1):
            lt |= eq & (a[:, i] < np.uint64(pl[i]))
            eq &= a[:, i] == np.uint64(pl[i])
        if not lt.all():
            raise SerializationError("InvalidData: field element not reduced")

    # ---- y sign:  y > -y  <=>  y > (q-1)/2  (y != 0) ----
    def _gt_half(self, canon_fq):
        """canon_fq: (n, fqb) uint8 canonical LE -> bool (n,)"""
        L = self.fqb // 8
        a = np.ascontiguousarray(canon_fq).reshape(-1, L, 8).view(np.uint64).reshape(-1, L)
        gt = np.zeros(a.shape[0], dtype=bool)
        eq = np.ones(a.shape[0], dtype=bool)
        for i in range(L - 1, -1, -1):
            gt |= eq & (a[:, i] > self._half_limbs[i])
            eq &= a[:, i] == self._half_limbs[i]
        return gt

    def _y_negative(self, canon, group):
        """canon: (n, coords, fqb) canonical; SWFlags::from_y_coordinate / zcash 'lexicographically largest'."""
        if group == 1:
            return self._gt_half(canon[:, 1])
        y0, y1 = canon[:, 2], canon[:, 3]
        c1_zero = ~y1.any(axis=1)
        return np.where(c1_zero, self._gt_half(y0), self._gt_half(y1))

    # ---- points ----
    def points_to_wire(self, group, abi, compress=False):
        """ABI array of n points -> concatenated ark-serialize encodings (n * point_size bytes)."""
        coords = 2 if group == 1 else 4
        fqb = self.fqb
        abi = np.ascontiguousarray(abi, dtype=np.uint8).reshape(-1, coords * fqb)
        n = abi.shape[0]
        if n == 0:
            return b""
        inf = ~abi.any(axis=1)                          # Montgomery zero == canonical zero == the ABI's infinity
        canon = self._convert(1, abi, False).reshape(n, coords, fqb)
        neg = self._y_negative(canon, group) & ~inf
        half = coords // 2
        if not self.zcash:
            out = canon[:, :half].reshape(n, -1).copy() if compress else canon.reshape(n, -1).copy()
            out[:, -1] |= np.where(inf, 0x40, np.where(neg, 0x80, 0)).astype(np.uint8)
            return out.tobytes()
        be = canon[:, :, ::-1]                          # big-endian coordinates
        if group == 2:
            be = be[:, [1, 0, 3, 2]]                    # c1 || c0
        out = (be[:, :half] if compress else be).reshape(n, -1).copy()
        flags = np.where(inf, 0x40, 0).astype(np.uint8)
        if compress:
            flags |= 0x80
            flags |= np.where(neg, 0x20, 0).astype(np.uint8)
        out[:, 0] |= flags
        return out.tobytes()

    def point_size(self, group, compress=False):
        full = self.g1b if group == 1 else self.g2b
        return full // 2 if compress else full

    def points_from_wire(self, group, buf, n, compress=False):
        """n encoded points -> ABI array.  Unchecked like the reference's `deserialize_*_unchecked` (no curve or
        subgroup check); flags and canonical-range are validated as ark-serialize always does."""
        coords = 2 if group == 1 else 4
        fqb = self.fqb
        half = coords // 2
        size = self.point_size(group, compress)
        a = np.frombuffer(bytes(buf), dtype=np.uint8)
        if a.size != n * size:
            raise SerializationError("IoError: wrong length")
        if n == 0:
            return np.zeros(0, dtype=np.uint8)
        a = a.reshape(n, size).copy()
        if not self.zcash:
            fl = a[:, -1] & 0xC0
            if (fl == 0xC0).any():
                raise SerializationError("UnexpectedFlags")
            a[:, -1] &= 0x3F
            inf, neg = fl == 0x40, fl == 0x80
            canon = a.reshape(n, -1, fqb)
        else:
            fl = a[:, 0] & 0xE0
            if ((fl & 0x80 != 0) != compress).any():
                raise SerializationError("UnexpectedFlags: compression bit")
            if not compress and (fl & 0x20 != 0).any():
                raise SerializationError("UnexpectedFlags: sort bit on an uncompressed point")
            a[:, 0] &= 0x1F
            inf, neg = fl & 0x40 != 0, fl & 0x20 != 0
            if (inf & neg).any():
                raise SerializationError("UnexpectedFlags")
            be = a.reshape(n, -1, fqb)
            if group == 2:
                be = be[:, [1, 0]] if compress else be[:, [1, 0, 3, 2]]
            canon = be[:, :, ::-1]
        canon = np.ascontiguousarray(canon)
        self._check_reduced(canon.reshape(-1), fqb, self.q)
        if inf.any() and canon[inf].any():
            raise SerializationError("InvalidData: infinity with non-zero coordinates")
        if compress:
            canon = self._decompress(group, canon, inf, neg)
        mont = self._convert(1, canon, True).reshape(n, coords * fqb)
        mont[inf] = 0
        return mont.reshape(-1)

    # ---- decompression (host big-int; the reference only ever *writes* compressed points, node.rs:611) ----
    def _fq_sqrt(self, a):
        q = self.q                                       # q = 3 mod 4 on both curves
        s = pow(a, (q + 1) // 4, q)
        return s if s * s % q == a else None

    def _fq2_sqrt(self, a0, a1):
        q = self.q
        if a1 == 0:
            s = self._fq_sqrt(a0)
            if s is not None:
                return s, 0
            s = self._fq_sqrt(-a0 % q)                   # sqrt(a0) = s*u since u^2 = -1
            return (0, s) if s is not None else None
        alpha = self._fq_sqrt((a0 * a0 + a1 * a1) % q)   # norm
        if alpha is None:
            return None
        inv2 = pow(2, -1, q)
        delta = (a0 + alpha) * inv2 % q
        c0 = self._fq_sqrt(delta)
        if c0 is None:
            delta = (a0 - alpha) * inv2 % q
            c0 = self._fq_sqrt(delta)
            if c0 is None:
                return None
        c1 = a1 * pow(2 * c0, -1, q) % q
        return c0, c1

    def _decompress(self, group, xs, inf, neg):
        q, fqb = self.q, self.fqb
        n = xs.shape[0]
        out = np.zeros((n, 2 if group == 1 else 4, fqb), dtype=np.uint8)
        b1 = 3 if self.curve == "bn254" else 4
        if group == 2:
            if self.curve == "bn254":                    # b' = 3 / (9 + u)
                d = pow(82, -1, q)
                b2 = (27 * d % q, (-3 * d) % q)
            else:                                        # b' = 4 (1 + u)
                b2 = (4, 4)
        for i in range(n):
            if inf[i]:
                continue
            if group == 1:
                x = int.from_bytes(xs[i, 0].tobytes(), "little")
                y = self._fq_sqrt((x * x * x + b1) % q)
                if y is None:
                    raise SerializationError("InvalidData: x is not on the curve")
                if (y > self.half) != bool(neg[i]):
                    y = (-y) % q
                vals = (x, y)
            else:
                x0 = int.from_bytes(xs[i, 0].tobytes(), "little")
                x1 = int.from_bytes(xs[i, 1].tobytes(), "little")
                s0, s1 = (x0 * x0 - x1 * x1) % q, 2 * x0 * x1 % q
                c0, c1 = (s0 * x0 - s1 * x1 + b2[0]) % q, (s0 * x1 + s1 * x0 + b2[1]) % q
                y = self._fq2_sqrt(c0, c1)
                if y is None:
                    raise SerializationError("InvalidData: x is not on the curve")
                y0, y1 = y
                is_neg = (y1 > self.half) if y1 else (y0 > self.half)
                if is_neg != bool(neg[i]):
                    y0, y1 = (-y0) % q, (-y1) % q
                vals = (x0, x1, y0, y1)
            for k, v in enumerate(vals):
                out[i, k] = np.frombuffer(v.to_bytes(fqb, "little"), dtype=np.uint8)
        return out

    # ---- Vec<Affine> ----
    def write_points_vec(self, w, group, abi, compress=False):
        pb = self.g1b if group == 1 else self.g2b
        abi = _host(abi, self.ctx)
        w.u64(abi.size // pb)
        w.put(self.points_to_wire(group, abi, compress))

    def read_points_vec(self, r, group, compress=False):
        n = r.u64()
        return self.points_from_wire(group, r.take(n * self.point_size(group, compress)), n, compress)

    def write_point(self, w, group, abi, compress=False):
        w.put(self.points_to_wire(group, _host(abi, self.ctx), compress))

    def read_point(self, r, group, compress=False):
        return self.points_from_wire(group, r.take(self.point_size(group, compress)), 1, compress)

    # ---- cp-groth16 records (field order = derive order, data_structures.rs) ----
    def write_proof(self, w, proof, compress=False):                       # :6-16
        self.write_point(w, 1, proof.a, compress)
        self.write_point(w, 2, proof.b, compress)
        self.write_point(w, 1, proof.c, compress)
        ds = np.concatenate([np.asarray(d, dtype=np.uint8).reshape(-1) for d in proof.ds]) if proof.ds else np.zeros(0, np.uint8)
        self.write_points_vec(w, 1, ds, compress)

    def read_proof(self, r, compress=False):
        a = self.read_point(r, 1, compress)
        b = self.read_point(r, 2, compress)
        c = self.read_point(r, 1, compress)
        ds = self.read_points_vec(r, 1, compress)
        return Proof(a, b, c, [ds[i:i + self.g1b].copy() for i in range(0, ds.size, self.g1b)])

    def write_vk(self, w, vk):                                              # :32-46
        self.write_point(w, 1, vk.alpha_g)
        self.write_point(w, 2, vk.beta_h)
        self.write_point(w, 2, vk.gamma_h)
        self.write_point(w, 2, vk.last_delta_h)
        self.write_points_vec(w, 1, vk.gamma_abc_g)
        self.write_points_vec(w, 2, vk.deltas_h)

    def read_vk(self, r):
        return VerifyingKey(self.read_point(r, 1), self.read_point(r, 2), self.read_point(r, 2),
                            self.read_point(r, 2), self.read_points_vec(r, 1), self.read_points_vec(r, 2))

    def write_ck(self, w, ck):                                              # :107-114
        self.write_point(w, 1, ck.last_delta_g)
        w.u64(len(ck.deltas_abc_g))
        for v in ck.deltas_abc_g:
            self.write_points_vec(w, 1, v)

    def read_ck(self, r):
        last = self.read_point(r, 1)
        return CommitterKey(last, [self.read_points_vec(r, 1) for _ in range(r.u64())])

    def write_pk(self, w, pk):                                              # :65-83
        self.write_vk(w, pk.vk)
        self.write_point(w, 1, pk.beta_g)
        self.write_points_vec(w, 1, pk.a_g)
        self.write_points_vec(w, 1, pk.b_g)
        self.write_points_vec(w, 2, pk.b_h)
        self.write_points_vec(w, 1, pk.h_g)
        self.write_ck(w, pk.ck)
        self.write_points_vec(w, 1, pk.deltas_g)

    def read_pk(self, r):
        vk = self.read_vk(r)
        beta_g = self.read_point(r, 1)
        a_g = self.read_points_vec(r, 1)
        b_g = self.read_points_vec(r, 1)
        b_h = self.read_points_vec(r, 2)
        h_g = self.read_points_vec(r, 1)
        ck = self.read_ck(r)
        deltas_g = self.read_points_vec(r, 1)
        return ProvingKey(vk, beta_g, a_g, b_g, b_h, h_g, ck, deltas_g)

    # ---- worker responses (distributed-prover/src/worker.rs:20-52) ----
    def stage0_response_to_wire(self, resp):
        w = Writer()
        w.u64(resp.subcircuit_idx)
        self.write_point(w, 1, resp.com)
        if len(resp.com_seed) != 32:
            raise SerializationError("InvalidData: com_seed is [u8; 32]")
        w.put(resp.com_seed)
        return w.getvalue()

    def stage0_response_from_wire(self, buf):
        from .worker import Stage0Response
        r = Reader(buf)
        idx = r.u64()
        com = self.read_point(r, 1)
        return Stage0Response(idx, com, bytes(r.take(32)))

    def stage1_response_to_wire(self, resp):
        w = Writer()
        w.u64(resp.subcircuit_idx)
        self.write_proof(w, resp.proof)
        return w.getvalue()

    def stage1_response_from_wire(self, buf):
        from .worker import Stage1Response
        r = Reader(buf)
        idx = r.u64()
        return Stage1Response(idx, self.read_proof(r))

    # ---- coordinator requests (distributed-prover/src/coordinator.rs:195-261, 520-622), ROM circuits ----
    # TranscriptEntry (transcript/mod.rs:216-233): tag byte 0 (Rom) | addr u64 | val Fr canonical.
    def _fr_wire(self, v):
        return int(v % self.r).to_bytes(self.frb, "little")

    def _fr_read(self, r):
        v = int.from_bytes(bytes(r.take(self.frb)), "little")
        if v >= self.r:
            raise SerializationError("InvalidData: field element not reduced")
        return v

    def _write_entry(self, w, e):
        """`TranscriptEntry` (transcript/mod.rs:216-233): tag 0 = Rom (addr u64, val), tag 1 = Ram (addr, val, i as
        Vec<bool>, read).  ROM entries may be plain (addr, val) tuples."""
        from .transcript import RamTranscriptEntry, RomTranscriptEntry
        if isinstance(e, RamTranscriptEntry):
            w.put(b"\x01" + e.to_wire(self.frb))
            return
        if isinstance(e, RomTranscriptEntry):
            e = (e.addr, e.val)
        w.put(b"\x00")
        w.u64(e[0])
        w.put(self._fr_wire(e[1]))

    def _read_entry(self, r):
        tag = bytes(r.take(1))[0]
        if tag == 0:
            return (r.u64(), self._fr_read(r))
        if tag == 1:
            from .transcript import RamTranscriptEntry
            addr, val = r.u64(), self._fr_read(r)
            if r.u64() != 32:
                raise SerializationError("InvalidData: Unsigned32 is 32 bools")
            raw = bytes(r.take(33))
            if any(b > 1 for b in raw):
                raise SerializationError("InvalidData: bool byte > 1")
            return RamTranscriptEntry(addr, val, sum(int(b) << k for k, b in enumerate(raw[:32])), bool(raw[32]))
        raise SerializationError("InvalidData: TranscriptEntry tag %d" % tag)

    def stage0_request_to_wire(self, req):
        """Stage0RequestRef (coordinator.rs:229-249): subcircuit_idx | Vec<TranscriptEntry> time | Vec<..> addr."""
        w = Writer()
        w.u64(req.subcircuit_idx)
        for trace in (req.time_ordered_subtrace, req.addr_ordered_subtrace):
            w.u64(len(trace))
            for e in trace:
                self._write_entry(w, e)
        return w.getvalue()

    def stage0_request_from_wire(self, buf):
        from .worker import Stage0Request
        r = Reader(buf)
        idx = r.u64()
        traces = []
        for _ in range(2):
            n = r.u64()
            traces.append([self._read_entry(r) for _ in range(n)])
        return Stage0Request(idx, traces[0], traces[1])

    def stage1_request_to_wire(self, req):
        """Stage1RequestRef (coordinator.rs:569-593): subcircuit_idx | cur_leaf | next_leaf_membership | root |
        serialized_witnesses | circ_params.  cur_leaf = ExecTreeLeaf { evals: RunningEvaluation (tag 0 = Rom:
        time_ordered_eval, addr_ordered_eval, Option<(entry_chal, tr_chal)>), last_subtrace_entry }.
        next_leaf_membership = ark-crypto-primitives 0.4 `merkle_tree::Path { leaf_sibling_hash, auth_path: Vec<_>,
        leaf_index: usize }` with Fr digests (Poseidon) - third-party layout, restated from memory."""
        w = Writer()
        w.u64(req.subcircuit_idx)
        w.put(b"\x00")
        w.put(self._fr_wire(req.time_ordered_eval))
        w.put(self._fr_wire(req.addr_ordered_eval))
        if req.challenges is None:
            w.put(b"\x00")
        else:
            w.put(b"\x01")
            w.put(self._fr_wire(req.challenges[0]))
            w.put(self._fr_wire(req.challenges[1]))
        self._write_entry(w, req.last_subtrace_entry)
        w.put(self._fr_wire(req.leaf_sibling_hash))
        w.u64(len(req.auth_path))
        for h in req.auth_path:
            w.put(self._fr_wire(h))
        w.u64(req.leaf_index)
        w.put(self._fr_wire(req.root))
        write_bytes_vec(w, req.serialized_witnesses)
        for v in req.circ_params:
            w.u64(v)
        return w.getvalue()

    def stage1_request_from_wire(self, buf):
        from .worker import Stage1Request
        r = Reader(buf)
        idx = r.u64()
        if bytes(r.take(1))[0] != 0:
            raise SerializationError("InvalidData: only ROM running evaluations (tag 0) are supported")
        te, ae = self._fr_read(r), self._fr_read(r)
        opt = bytes(r.take(1))[0]
        if opt not in (0, 1):
            raise SerializationError("InvalidData: Option tag")
        chal = (self._fr_read(r), self._fr_read(r)) if opt else None
        last = self._read_entry(r)
        sib = self._fr_read(r)
        path = [self._fr_read(r) for _ in range(r.u64())]
        leaf_index = r.u64()
        root = self._fr_read(r)
        wit = bytes(read_bytes_vec(r))
        params = (r.u64(), r.u64(), r.u64())
        return Stage1Request(idx, 0, te, ae, chal, last, sib, path, leaf_index, root, wit, params)

    def stage0_response_size(self):
        return 8 + self.g1b + 32

    def stage1_response_size(self, n_ds=1):
        return 8 + 2 * self.g1b + self.g2b + 8 + n_ds * self.g1b

    def split_flattened(self, flat, item_size):
        """deserialize_flattened_bytes! (mpi-snark/src/lib.rs:55-65): the gathered buffer is cut into
        `chunks_exact(item_size)`; a trailing partial chunk is ignored, as chunks_exact does."""
        flat = bytes(flat)
        return [flat[i:i + item_size] for i in range(0, len(flat) - item_size + 1, item_size)]


def _host(x, ctx):
    """numpy view of an ABI array that may live on the device (capi.DeviceBuffer)."""
    if isinstance(x, np.ndarray):
        return x.reshape(-1)
    if hasattr(x, "to_host"):
        return np.asarray(x.to_host(), dtype=np.uint8).reshape(-1)
    return np.asarray(x, dtype=np.uint8).reshape(-1)


# --------------------------------------------------------------------------------------- the key file
class ProvingKeys:
    """mpi-snark/src/data_structures.rs:41-51: the file `setup-*` writes and `node work` reads
    (node.rs:231-237, 315).  `matrices` are NOT in the file — the reference re-synthesises them per proof; a
    drop-in supplies them per class (from the circuit or a circom .r1cs) before `ProvingKey.upload`."""

    def __init__(self, circuit_id, serialized_circ_params, minimal_proving_keys, subcircuit_representative_map):
        self.circuit_id = circuit_id
        self.serialized_circ_params = bytes(serialized_circ_params)
        self.minimal_proving_keys = dict(minimal_proving_keys)
        self.subcircuit_representative_map = dict(subcircuit_representative_map)

    def get_pk(self, subcircuit_idx):                                       # :93-101
        if subcircuit_idx not in self.subcircuit_representative_map:
            raise KeyError("subcircuit index out of range")
        rep = self.subcircuit_representative_map[subcircuit_idx]
        if rep not in self.minimal_proving_keys:
            raise KeyError("missing proving key")
        return self.minimal_proving_keys[rep]

    def get_id_str(self):
        return self.circuit_id

    def num_subcircuits(self):
        return len(self.subcircuit_representative_map)

    def serialize(self, codec, with_id=True):
        """with_id=True: the derived impl on `ProvingKeys` (what `pks.serialize_uncompressed` writes to disk);
        with_id=False: the hand-written impl on `&ProvingKeys` (:112-135), which omits `circuit_id`."""
        w = Writer()
        if with_id:
            write_bytes_vec(w, self.circuit_id.encode("utf-8"))
        write_bytes_vec(w, self.serialized_circ_params)
        w.u64(len(self.minimal_proving_keys))
        for k in sorted(self.minimal_proving_keys):
            w.u64(k)
            codec.write_pk(w, self.minimal_proving_keys[k])
        write_usize_map(w, self.subcircuit_representative_map)
        return w.getvalue()

    @classmethod
    def deserialize(cls, codec, buf, with_id=True):
        r = Reader(buf)
        cid = read_bytes_vec(r).decode("utf-8") if with_id else ""
        params = read_bytes_vec(r)
        pks = {}
        for _ in range(r.u64()):
            k = r.u64()
            pks[k] = codec.read_pk(r)
        rep = read_usize_map(r)
        return cls(cid, params, pks, rep)


from .chacha import ChaChaRng, ChaCha12Rng, _chacha_block, fr_rand_mont, commitment_randomness  # noqa: E402,F401
