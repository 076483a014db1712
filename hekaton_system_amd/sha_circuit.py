"""A re-implemented gadget set for the big-merkle subcircuits, with a trace -> assignment witness generator
(SURVEY.md section 8f row 2).

The reference synthesises `MerkleTreeCircuit` subcircuits with ark-r1cs-std / ark-crypto-primitives gadgets
(`Sha256Gadget::digest`, `UInt8`, `Boolean::le_bits_to_fp_var`; distributed-prover/src/tree_hash_circuit.rs:98-111,
313-398) inside the timed region (`prover.rs:70-75`, `node.rs:589-596`).  Those gadget crates are third-party and
absent, so their constraint LAYOUT cannot be reproduced: PARITY UNPINNED for the matrices.  What is reproduced is the
FUNCTION of a subcircuit and the shape of its two-stage R1CS:

    stage 0   the subcircuit's portal subtraces: (addr, val) of its time-ordered ROM operations and of its slice of the
              address-ordered trace (subcircuit_circuit.rs:139-160; rom_transcript.rs:286-306: 2 witnesses per entry)
    stage 1   instance (1, entry_chal, tr_chal, root); leaf: 64 witnessed bytes -> `ns` iterations of SHA-256 -> digest
              truncated to 27 bytes, packed little-endian into a field element (vkd/util.rs:17-28 `digest_to_fpvar`)
              = the value it `set`s; parent: the two `get` values unpacked (`fpvar_to_digest`) -> 54 bytes -> the same
              chain; root: digest field == root; padding: the chain over 64 zero bytes; every subcircuit: running
              products of (tr_chal - (val + entry_chal * addr)) over both subtraces (rom_transcript.rs:77-107), equal at
              the last subcircuit, and value-consistency of equal addresses in the address-ordered slice.
              Also (round 3): the witnessed previous leaf (evals + last address-ordered entry of the previous
              subcircuit), the address-step check of every consecutive pair of the address-ordered slice - next address
              equal or larger by exactly one, equal addresses carry equal values (rom_portal_manager.rs:151-165) - and the
              Poseidon Merkle membership of the subcircuit's OWN execution leaf (time eval, addr eval, last entry) under
              the public root (subcircuit_circuit.rs:233-260; poseidon.py restates the hash).  The instance is the
              reference's: (1, entry_chal, tr_chal, exec-tree root); the SHA tree's root hash is a witness, as in
              tree_hash_circuit.rs:29-36,369-372 (`input_digest`: "TODO: Make this an actual public input").

One program, two interpreters (`Tape`): BUILD records the R1CS rows (vectorised, 32 rows per word operation);
EVAL runs the same program over a BATCH of subcircuits with numpy word arithmetic and emits the assignment directly
- no constraint-system objects, no per-variable closures: the reference's `generate_constraints` replaced by a
bit-sliced trace.  Bits leave as one byte each (`assignment_bits`), so a proof's assignment crosses PCIe as ~1 MB
instead of 32 bytes per variable; `hk`'s Montgomery expansion is a table lookup (0 / 1) plus the few full-width values.

SHA-256 gadget cost here: 26.8 k constraints per compression (XOR 1 constraint / bit, Ch 1, Maj 2, modular additions
1 + booleanity of result and carry bits).
"""
import contextlib
import hashlib
import os

import numpy as np

from .cp_groth16 import CURVE_PARAMS, FrCodec, MultiStageConstraintSynthesizer

K256 = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
IV256 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
INNER_HASH_SIZE = 27          # bytes of the digest that become the node hash (vkd/sparse_tree.rs:42)
ONE = 0                       # column of the constant 1


class Word:
    """32 bit positions, least significant first: `cols[i]` = R1CS column of bit i, or -1 for the constant 0
    (shifts).  `val`: the word's value over the batch (EVAL mode), else None.  (`vid`, `rot`, `shr`): the word as a view
    of value `vid` of the WORD PROGRAM the tape records (rotated right by `rot`, or shifted right by `shr`)."""
    __slots__ = ("cols", "val", "vid", "rot", "shr")

    def __init__(self, cols, val, vid, rot=0, shr=0):
        self.cols, self.val, self.vid, self.rot, self.shr = cols, val, vid, rot, shr

    def ref(self):
        return self.vid | (self.rot << 20) | (self.shr << 25)


def _rotr(w, k):
    assert w.shr == 0
    return Word(np.roll(w.cols, -k), None if w.val is None else ((w.val >> np.uint32(k)) | (w.val << np.uint32(32 - k))),
                w.vid, (w.rot + k) % 32, 0)


def _shr(w, k):
    assert w.rot == 0 and w.shr == 0
    cols = np.concatenate([w.cols[k:], np.full(k, -1, np.int64)])
    return Word(cols, None if w.val is None else (w.val >> np.uint32(k)), w.vid, 0, k)


# word-program opcodes (interpreted by k_word_program on the GPU and by `run_word_program` on the host)
OP_INPUT, OP_CONST, OP_XOR, OP_CH, OP_AND, OP_MAJ, OP_ADD, OP_PACK4 = range(8)
# Macro entries: one SHA-256 round / one message-schedule step as ONE program entry that produces the same values, in the
# same order, as the entries of its gadgets would (the interpreter pays its per-entry cost - decode, operand round trips
# through the lane's ring - once per round instead of nine times):
#   OP_SHA_ROUND  refs[a .. a+9) = a, b, c, d, e, f, g, h, w; imm = K[t]; 11 values: r6^r11 (e), S1, ch, r2^r13 (a), S0,
#                 a & b, maj, (d + h + S1 + ch + w + K) low / carry, (h + S1 + ch + w + S0 + maj + K) low / carry
#   OP_SHA_SCHED  refs[a .. a+4) = w[t-15], w[t-2], w[t-7], w[t-16]; 6 values: r7^r18, s0, r17^r19, s1, the sum's low / carry
OP_SHA_ROUND, OP_SHA_SCHED = 8, 9
OP_VALUES = {OP_ADD: 2, OP_SHA_ROUND: 11, OP_SHA_SCHED: 6}          # values per entry (1 otherwise)


class Tape:
    """BUILD: rows accumulate as (row, col, coeff) triplets per matrix.  EVAL (batch > 0): allocation records."""

    def __init__(self, n_inst, batch=0):
        self.n_inst = n_inst
        self.batch = batch
        self.build = batch == 0
        self.macro_ops = not os.environ.get("HK_WPROG_NO_MACRO")       # one entry per SHA-256 round / schedule step
        self.n_wit = 0
        self.n_rows = 0
        self.trip = {m: [] for m in "ABC"}      # lists of (rows, cols, coefs) int64 arrays
        self.big = []                           # rows with coefficients beyond int64: (A, B, C) as [(coef int, col)]
        self.consts = {}                        # pinned constant words by value (allocated at first use, both modes)
        self.bit_records = []                   # EVAL: (first column, bit positions, word values)
        self.full_records = []                  # EVAL: (column, [python int per batch element])
        # the word program (recorded identically in both modes): one entry per VALUE
        self.prog = []                          # (opcode, a, b, c, imm)
        self.refs = []                          # operand references of ADD / PACK4 entries
        self.n_values = 0
        self.n_inputs = 0
        self.alloc_map = []                     # (first column, bit positions, value id)
        self._in_macro = False

    def _value(self, op, a=0, b=0, c=0, imm=0, count=1):
        vid = self.n_values
        if not self._in_macro:
            self.prog.append((op, a, b, c, imm))
        self.n_values += count
        return vid

    @contextlib.contextmanager
    def _macro(self, op, words, imm=0):
        """The gadget calls inside the block allocate their values and rows as always, but the program gets ONE entry."""
        assert not self._in_macro
        first_ref, start = len(self.refs), self.n_values
        self.refs += [w.ref() for w in words]
        self._in_macro = True
        try:
            yield
        finally:
            self._in_macro = False
        assert self.n_values - start == OP_VALUES[op]
        self.prog.append((op, first_ref, len(words), 0, imm))

    def input_value(self):
        """The next 32-bit input of the subcircuit (a leaf word, a byte of a child hash) as a program value."""
        k = self.n_inputs
        self.n_inputs += 1
        return self._value(OP_INPUT, imm=k)

    # ---- allocation -----------------------------------------------------------------------------------
    def _alloc(self, k):
        first = self.n_inst + self.n_wit
        self.n_wit += k
        return first

    def _rows(self, k):
        first = self.n_rows
        self.n_rows += k
        return np.arange(first, first + k, dtype=np.int64)

    def _emit(self, m, rows, cols, coefs):
        self.trip[m].append((np.asarray(rows, np.int64), np.asarray(cols, np.int64), np.asarray(coefs, np.int64)))

    def alloc_bits(self, pos, val, boolean, vid):
        """New witness bits for the bit positions `pos` of the word value `val` (= program value `vid`); booleanity
        rows b * (1 - b) = 0 when the bits are not boolean by construction.  Returns their columns."""
        k = len(pos)
        first = self._alloc(k)
        cols = np.arange(first, first + k, dtype=np.int64)
        if k:
            self.alloc_map.append((first, np.asarray(pos, np.uint32), vid))
        if self.build:
            if boolean:
                r = self._rows(k)
                self._emit("A", r, cols, np.ones(k, np.int64))
                self._emit("B", np.concatenate([r, r]), np.concatenate([np.full(k, ONE), cols]),
                           np.concatenate([np.ones(k, np.int64), -np.ones(k, np.int64)]))
        else:
            self.bit_records.append((first, np.asarray(pos, np.uint32), val))
        return cols

    def alloc_word(self, val, boolean, vid):
        return Word(self.alloc_bits(np.arange(32), val, boolean, vid), val, vid)

    def input_word(self, val):
        return self.alloc_word(val, True, self.input_value())

    def alloc_full(self, vals):
        """One full-width witness; vals: list of Python ints (EVAL) or None."""
        col = self._alloc(1)
        if not self.build:
            self.full_records.append((col, vals))
        return col

    # ---- bit gadgets (vectorised over the 32 positions of a word) -----------------------------------------
    def xor(self, x, y):
        """x ^ y; positions where one side is the constant 0 pass the other side through without a constraint."""
        both = (x.cols >= 0) & (y.cols >= 0)
        val = None if self.build else x.val ^ y.val
        pos = np.nonzero(both)[0]
        vid = self._value(OP_XOR, x.ref(), y.ref())
        new = self.alloc_bits(pos, val, False, vid)
        cols = np.where(x.cols >= 0, x.cols, y.cols).copy()
        cols[pos] = new
        if self.build and len(pos):
            k = len(pos)
            r = self._rows(k)
            xa, ya = x.cols[pos], y.cols[pos]
            self._emit("A", r, xa, np.full(k, 2))                                  # (2 x) * (y) = x + y - t
            self._emit("B", r, ya, np.ones(k, np.int64))
            self._emit("C", np.concatenate([r, r, r]), np.concatenate([xa, ya, new]),
                       np.concatenate([np.ones(k, np.int64), np.ones(k, np.int64), -np.ones(k, np.int64)]))
        return Word(cols, val, vid)

    def xor3(self, x, y, z):
        return self.xor(self.xor(x, y), z)

    def ch(self, e, f, g):
        """(e & f) ^ (~e & g):  e * (f - g) = ch - g."""
        val = None if self.build else (e.val & f.val) ^ (~e.val & g.val)
        w = self.alloc_word(val, False, self._value(OP_CH, e.ref(), f.ref(), g.ref()))
        if self.build:
            r = self._rows(32)
            self._emit("A", r, e.cols, np.ones(32, np.int64))
            self._emit("B", np.concatenate([r, r]), np.concatenate([f.cols, g.cols]),
                       np.concatenate([np.ones(32, np.int64), -np.ones(32, np.int64)]))
            self._emit("C", np.concatenate([r, r]), np.concatenate([w.cols, g.cols]),
                       np.concatenate([np.ones(32, np.int64), -np.ones(32, np.int64)]))
        return w

    def maj(self, a, b, c):
        """t = a * b;  c * (a + b - 2 t) = maj - t."""
        tv = None if self.build else a.val & b.val
        t = self.alloc_word(tv, False, self._value(OP_AND, a.ref(), b.ref()))
        val = None if self.build else (a.val & b.val) ^ (a.val & c.val) ^ (b.val & c.val)
        w = self.alloc_word(val, False, self._value(OP_MAJ, a.ref(), b.ref(), c.ref()))
        if self.build:
            o = np.ones(32, np.int64)
            r = self._rows(32)
            self._emit("A", r, a.cols, o); self._emit("B", r, b.cols, o); self._emit("C", r, t.cols, o)
            r = self._rows(32)
            self._emit("A", r, c.cols, o)
            self._emit("B", np.concatenate([r, r, r]), np.concatenate([a.cols, b.cols, t.cols]), np.concatenate([o, o, -2 * o]))
            self._emit("C", np.concatenate([r, r]), np.concatenate([w.cols, t.cols]), np.concatenate([o, -o]))
        return w

    def add(self, words, const=0):
        """Sum of the words (and a constant) mod 2^32: result bits and carry bits are witnessed (boolean) and one
        linear row ties them: (sum_j sum_i 2^i w_j[i] + const) * 1 = sum_i 2^i r[i] + 2^32 * carry."""
        k = len(words) + (1 if const else 0)
        cbits = max(1, (k - 1).bit_length())
        if self.build:
            tot = None
        else:
            tot = np.full(self.batch, const, np.uint64)
            for w in words:
                tot = tot + w.val.astype(np.uint64)
        first_ref = len(self.refs)
        self.refs += [w.ref() for w in words]
        vid = self._value(OP_ADD, first_ref, len(words), 0, const, count=2)          # two values: low word, carry
        r = self.alloc_word(None if self.build else (tot & np.uint64(0xffffffff)).astype(np.uint32), True, vid)
        carry = self.alloc_bits(np.arange(cbits), None if self.build else (tot >> np.uint64(32)).astype(np.uint32), True,
                                vid + 1)
        if self.build:
            row = self._rows(1)[0]
            pw = (1 << np.arange(32, dtype=np.int64))
            cols, coefs = [], []
            for w in words:
                nz = w.cols >= 0
                cols.append(w.cols[nz]); coefs.append(pw[nz])
            if const:
                cols.append(np.array([ONE])); coefs.append(np.array([const], np.int64))
            cols, coefs = np.concatenate(cols), np.concatenate(coefs)
            self._emit("A", np.full(len(cols), row), cols, coefs)
            self._emit("B", [row], [ONE], [1])
            self._emit("C", np.full(32 + cbits, row), np.concatenate([r.cols, carry]),
                       np.concatenate([pw, (1 << (32 + np.arange(cbits, dtype=np.int64)))]))
        return r

    def big_row(self, a, b, c):
        """One row with arbitrary (Python int) coefficients: lists of (coef, col)."""
        if self.build:
            row = self._rows(1)[0]
            self.big.append((row, a, b, c))

    # ---- SHA-256 ---------------------------------------------------------------------------------------
    def sha256_compress(self, state, block):
        """state: 8 Words or ints (the IV); block: 16 Words (big-endian message words).  Returns 8 Words."""
        w = list(block)
        macro = self.macro_ops
        nothing = contextlib.nullcontext()
        for t in range(16, 64):
            with (self._macro(OP_SHA_SCHED, [w[t - 15], w[t - 2], w[t - 7], w[t - 16]]) if macro else nothing):
                s0 = self.xor3(_rotr(w[t - 15], 7), _rotr(w[t - 15], 18), _shr(w[t - 15], 3))
                s1 = self.xor3(_rotr(w[t - 2], 17), _rotr(w[t - 2], 19), _shr(w[t - 2], 10))
                w.append(self.add([s1, w[t - 7], s0, w[t - 16]]))
        first = isinstance(state[0], int)
        if first:                          # IV: witnessed constant words (booleanity rows pin nothing; equality rows do)
            st = [self.const_word(v) for v in state]
        else:
            st = list(state)
        a, b, c, d, e, f, g, h = st
        for t in range(64):
            with (self._macro(OP_SHA_ROUND, [a, b, c, d, e, f, g, h, w[t]], K256[t]) if macro else nothing):
                S1 = self.xor3(_rotr(e, 6), _rotr(e, 11), _rotr(e, 25))
                chv = self.ch(e, f, g)
                S0 = self.xor3(_rotr(a, 2), _rotr(a, 13), _rotr(a, 22))
                mj = self.maj(a, b, c)
                new_e = self.add([d, h, S1, chv, w[t]], K256[t])
                new_a = self.add([h, S1, chv, w[t], S0, mj], K256[t])
            a, b, c, d, e, f, g, h = new_a, a, b, c, new_e, e, f, g
        return [self.add([x, y]) for x, y in zip(st, [a, b, c, d, e, f, g, h])]

    def const_word(self, v):
        """A word whose 32 bits are pinned to the constant v: 32 rows b * 1 = bit (one allocation per distinct value)."""
        if v in self.consts:
            return self.consts[v]
        val = None if self.build else np.full(self.batch, v, np.uint32)
        w = self.alloc_word(val, False, self._value(OP_CONST, imm=v))
        if self.build:
            r = self._rows(32)
            o = np.ones(32, np.int64)
            self._emit("A", r, w.cols, o)
            self._emit("B", r, np.full(32, ONE), o)
            bits = (v >> np.arange(32)) & 1
            nz = np.nonzero(bits)[0]
            self._emit("C", r[nz], np.full(len(nz), ONE), np.ones(len(nz), np.int64))
        self.consts[v] = w
        return w

    def sha256_bytes(self, byte_words_be, n_bytes):
        """SHA-256 of an n_bytes message given as big-endian 32-bit Words (n_bytes % 4 may be non-zero only through
        pre-packed words): pads (0x80, zeros, 64-bit length) and compresses.  Returns the 8 digest Words."""
        total = ((n_bytes + 9 + 63) // 64) * 64
        n_words = total // 4
        msg = list(byte_words_be)
        assert len(msg) * 4 >= n_bytes
        # the padding words are constants: allocate them as pinned constant words
        pad_bytes = bytearray(total)
        pad_bytes[n_bytes] = 0x80
        pad_bytes[-8:] = (8 * n_bytes).to_bytes(8, "big")
        full_words = n_bytes // 4
        assert n_bytes % 4 == 0 or len(msg) == full_words + 1
        words = msg[:full_words]
        if n_bytes % 4:
            words.append(msg[full_words])       # caller already merged the 0x80 marker into the partial word
        for k in range(len(words), n_words):
            words.append(self.const_word(int.from_bytes(pad_bytes[4 * k:4 * k + 4], "big")))
        state = IV256
        for blk in range(total // 64):
            state = self.sha256_compress(state, words[16 * blk:16 * blk + 16])
        return state

    # ---- finishing -------------------------------------------------------------------------------------
    def csr(self, fc):
        """BUILD: the three matrices as (row_ptr u64, col u32, val Montgomery bytes)."""
        r_mod = fc.r
        out = []
        for m in "ABC":
            rows = np.concatenate([t[0] for t in self.trip[m]]) if self.trip[m] else np.zeros(0, np.int64)
            cols = np.concatenate([t[1] for t in self.trip[m]]) if self.trip[m] else np.zeros(0, np.int64)
            coefs = np.concatenate([t[2] for t in self.trip[m]]) if self.trip[m] else np.zeros(0, np.int64)
            big_rows, big_cols, big_vals = [], [], []
            idx = "ABC".index(m)
            for entry in self.big:
                for coef, col in entry[1 + idx]:
                    big_rows.append(entry[0]); big_cols.append(col); big_vals.append(coef % r_mod)
            uniq, inv = np.unique(coefs, return_inverse=True)
            table = [int(u) % r_mod for u in uniq.tolist()] + big_vals
            vidx = np.concatenate([inv, len(uniq) + np.arange(len(big_vals))]).astype(np.int64)
            rows = np.concatenate([rows, np.array(big_rows, np.int64)])
            cols = np.concatenate([cols, np.array(big_cols, np.int64)])
            order = np.argsort(rows, kind="stable")
            rows, cols, vidx = rows[order], cols[order], vidx[order]
            row_ptr = np.zeros(self.n_rows + 1, np.uint64)
            row_ptr[1:] = np.cumsum(np.bincount(rows, minlength=self.n_rows))
            tab = fc.enc(table).reshape(len(table), fc.nb)
            out.append((row_ptr, cols.astype(np.uint32), np.ascontiguousarray(tab[vidx]).ravel(), vidx, table))
        return out

    def word_program(self, n_v):
        """The recorded program in the arrays hk_wprog_upload takes: ops uint32 (n, 8), refs uint32, map uint32 (n_v):
        map[col] = value id << 5 | bit position for bit-valued columns, 0xffffffff otherwise (instance, full-width)."""
        ops = np.zeros((len(self.prog), 8), np.uint32)
        for k, (op, a, b, c, imm) in enumerate(self.prog):
            ops[k, :5] = (op, a, b, c, imm)
        vmap = np.full(n_v, 0xffffffff, np.uint32)
        for first, pos, vid in self.alloc_map:
            vmap[first:first + len(pos)] = (np.uint32(vid) << np.uint32(5)) | pos
        return ops, np.array(self.refs, np.uint32), vmap

    def assignment_bits(self, n_v):
        """EVAL: (batch, n_v) uint8 with every bit variable's value (column 0 = 1; full-width columns left 0)."""
        out = np.zeros((self.batch, n_v), np.uint8)
        out[:, ONE] = 1
        for first, pos, val in self.bit_records:
            out[:, first:first + len(pos)] = ((val[:, None] >> pos[None, :]) & np.uint32(1)).astype(np.uint8)
        return out


def poseidon_path_trace(leaf_cfg, node_cfg, leaf, siblings, index):
    """Every witness of the membership block of one subcircuit, in allocation order (= what `k_poseidon_path` writes):
    the leaf hash's permutation traces, then per level (bit, sibling, left, the two-to-one hash's trace).  The last
    value is the root the path leads to."""
    out = []
    cur = leaf_cfg.crh(leaf, out)
    for lvl, sib in enumerate(siblings):
        bit = (index >> lvl) & 1
        left, right = (sib, cur) if bit else (cur, sib)
        out += [bit, sib % leaf_cfg.p, left]
        cur = node_cfg.crh([left, right], out)
    return out


def poseidon_path_root(leaf_cfg, node_cfg, leaf, siblings, index):
    cur = leaf_cfg.crh(leaf)
    for lvl, sib in enumerate(siblings):
        cur = node_cfg.crh([sib, cur] if (index >> lvl) & 1 else [cur, sib])
    return cur


# ---------------------------------------------------------------------------------------------------------------------
class ShaMerkleSubcircuit(MultiStageConstraintSynthesizer):
    """One proving-key class of the re-implemented big-merkle circuit.  kind: "leaf" | "parent" | "root" | "padding";
    `first` marks subcircuit 0 (evals pinned to 1), `last` the final subcircuit (time eval == addr eval)."""
    N_INST = 4

    def __init__(self, curve, kind, ns, n_portals, first=False, last=False, depth=3):
        """depth = log2(number of subcircuits): the length of the execution tree's membership path."""
        assert kind in ("leaf", "parent", "root", "padding")
        self.curve, self.kind, self.ns, self.np_, self.first, self.last = curve, kind, ns, n_portals, first, last
        self.depth = depth
        from .poseidon import merkle_params
        self.leaf_cfg, self.node_cfg = merkle_params(curve)
        self.r = CURVE_PARAMS[curve]["r"]
        self.fc = FrCodec(curve)
        self.n_time = self.n_addr = n_portals
        self.n0 = 2 * (self.n_time + self.n_addr)
        t = Tape(self.N_INST)
        self._program(t, None)
        self.tape = t
        self.n_c = t.n_rows
        self.n_wit = t.n_wit
        self.n_v = self.N_INST + t.n_wit
        self._csr = None
        self._batch = None

    # ---- the program: identical in BUILD and EVAL --------------------------------------------------------
    def _program(self, t, inp):
        """inp (EVAL): dict with numpy / int inputs for the batch, see `witness_batch`."""
        ev = not t.build
        B = t.batch
        ni = self.N_INST
        ENTRY, TR, ROOT = 1, 2, 3
        # ---- stage 0: (addr, val) of every time-ordered and address-ordered entry
        time_e = [(t.alloc_full(inp["time"][k][0] if ev else None), t.alloc_full(inp["time"][k][1] if ev else None))
                  for k in range(self.n_time)]
        addr_e = [(t.alloc_full(inp["addr"][k][0] if ev else None), t.alloc_full(inp["addr"][k][1] if ev else None))
                  for k in range(self.n_addr)]
        assert t.n_wit == self.n0
        r_mod = self.r
        # ---- stage 1
        # running evaluations (rom_transcript.rs:77-107): eval' = eval * (tr_chal - (val + entry_chal * addr))
        def running(entries, start_vals, key):
            ev_col = t.alloc_full(start_vals if ev else None)
            cur = start_vals
            if self.first:
                t.big_row([(1, ev_col)], [(1, ONE)], [(1, ONE)])                    # subcircuit 0: eval = 1
            for k, (a_col, v_col) in enumerate(entries):
                if ev:
                    ech, tr = inp["entry_chal"], inp["tr_chal"]
                    e_vals = [(int(v) + ech * int(a)) % r_mod for a, v in zip(inp[key][k][0], inp[key][k][1])]
                    nxt = [c * ((tr - e) % r_mod) % r_mod for c, e in zip(cur, e_vals)]
                else:
                    e_vals = nxt = None
                e_col = t.alloc_full(e_vals)
                n_col = t.alloc_full(nxt)
                t.big_row([(1, ENTRY)], [(1, a_col)], [(1, e_col), (r_mod - 1, v_col)])
                t.big_row([(1, ev_col)], [(1, TR), (r_mod - 1, e_col)], [(1, n_col)])
                ev_col, cur = n_col, nxt
            return ev_col, cur
        t_final, t_vals = running(time_e, inp["time_eval0"] if ev else None, "time")
        a_final, a_vals = running(addr_e, inp["addr_eval0"] if ev else None, "addr")
        if self.last:
            t.big_row([(1, t_final), (r_mod - 1, a_final)], [(1, ONE)], [])
        # the previous leaf's last address-ordered entry (subcircuit_circuit.rs:167, 209-216): witnessed; padding - address
        # 0 - in front of subcircuit 0 (:199-203; `is_padding` compares the address only, rom_transcript.rs:249-253)
        prev = (t.alloc_full(inp["prev"][0] if ev else None), t.alloc_full(inp["prev"][1] if ev else None))
        if self.first:
            t.big_row([(1, prev[0])], [(1, ONE)], [])
        # address-step check of every consecutive pair of [previous entry] + slice (rom_portal_manager.rs:151-165):
        #   d = addr' - addr;  d * inv = 1 - same;  same * d = 0          (same = [d == 0], `is_eq`)
        #   (1 - same) * (d - 1) = 0                                     (not the same address -> exactly one larger)
        #   same * (val' - val) = 0                                      (`conditional_enforce_equal`)
        chain = [prev] + addr_e
        for k in range(1, len(chain)):
            (a0, v0), (a1, v1) = chain[k - 1], chain[k]
            if ev:
                prev_a = inp["prev"][0] if k == 1 else inp["addr"][k - 2][0]
                d = [(int(x) - int(y)) % r_mod for x, y in zip(inp["addr"][k - 1][0], prev_a)]
                inv = [pow(x, -1, r_mod) if x else 0 for x in d]
                same = [0 if x else 1 for x in d]
            else:
                inv = same = None
            inv_c, same_c = t.alloc_full(inv), t.alloc_full(same)
            t.big_row([(1, a1), (r_mod - 1, a0)], [(1, inv_c)], [(1, ONE), (r_mod - 1, same_c)])
            t.big_row([(1, same_c)], [(1, a1), (r_mod - 1, a0)], [])
            t.big_row([(1, ONE), (r_mod - 1, same_c)], [(1, a1), (r_mod - 1, a0), (r_mod - 1, ONE)], [])
            t.big_row([(1, same_c)], [(1, v1), (r_mod - 1, v0)], [])
        # ---- the subcircuit's own execution leaf is in the tree (subcircuit_circuit.rs:233-252)
        self.pos_col0 = ni + t.n_wit
        leaf_lcs = [[(1, t_final)], [(1, a_final)], [(1, addr_e[-1][0])], [(1, addr_e[-1][1])]]
        if ev:
            leaf_vals = [[t_vals[b], a_vals[b], int(inp["addr"][-1][0][b]) % r_mod, int(inp["addr"][-1][1][b]) % r_mod]
                         for b in range(B)]
            traces = [poseidon_path_trace(self.leaf_cfg, self.node_cfg, leaf_vals[b], inp["path_sib"][b], inp["path_idx"][b])
                      for b in range(B)]
            cols_vals = list(zip(*traces))                       # per witness of the block: its value per batch element
            self._pos_iter = iter(cols_vals)
        nxt = (lambda: t.alloc_full(list(next(self._pos_iter)))) if ev else (lambda: t.alloc_full(None))
        cur = self._poseidon_crh(t, self.leaf_cfg, leaf_lcs, nxt)
        for _lvl in range(self.depth):
            bit, sib, left = nxt(), nxt(), nxt()
            t.big_row([(1, bit)], [(1, ONE), (r_mod - 1, bit)], [])                               # boolean
            t.big_row([(1, bit)], [(1, sib), (r_mod - 1, cur)], [(1, left), (r_mod - 1, cur)])    # left = bit ? sib : cur
            right = [(1, sib), (1, cur), (r_mod - 1, left)]                                        # the other one
            cur = self._poseidon_crh(t, self.node_cfg, [[(1, left)], right], nxt)
        t.big_row([(1, cur), (r_mod - 1, ROOT)], [(1, ONE)], [])                                   # the public root
        self.pos_cols = ni + t.n_wit - self.pos_col0
        # ---- the hash chain
        if self.kind in ("leaf", "padding"):
            # 64 witnessed bytes as 16 big-endian words (bits boolean)
            words = [t.input_word(inp["leaf_words"][:, k] if ev else None) for k in range(16)]
            if self.kind == "padding":
                zero = t.const_word(0)       # EMPTY_LEAF: pin the input to zero through one pinned word
                for w in words:
                    self._enforce_word_eq(t, w, zero)
            digest = t.sha256_bytes(words, 64)
        else:
            # two `get`s: unpack each value (216 bits, little-endian per byte as `fpvar_to_digest`) into 27 bytes
            words = self._unpack_children(t, time_e, inp)
            digest = t.sha256_bytes(words, 54)
        for _ in range(self.ns - 1):
            digest = t.sha256_bytes(digest, 32)
        # digest -> field: first 27 bytes, each byte's bits little-endian, bytes in order (digest_to_fpvar)
        terms = self._digest_field_terms(digest)
        if self.kind == "leaf" or self.kind == "parent":
            out_col = time_e[-1][1]                        # the `set` is the subcircuit's last time-ordered operation
            t.big_row(terms, [(1, ONE)], [(1, out_col)])
        elif self.kind == "root":
            # the SHA tree's root hash: a witness in the reference too (tree_hash_circuit.rs:29-36 `input_digest`)
            sha_root = t.alloc_full(inp["sha_root"] if ev else None)
            self.sha_root_col = sha_root
            t.big_row(terms, [(1, ONE)], [(1, sha_root)])
        return digest

    def _poseidon_crh(self, t, cfg, inputs, nxt):
        """`poseidon::constraints::CRHGadget::evaluate` with an own layout: inputs = linear combinations [(coef, col)];
        per round the S-box chain of every S-boxed element and the new state are witnesses (`nxt()` allocates the next
        one, in the order poseidon.PoseidonConfig.permute traces them); returns the digest's column."""
        p, tt = cfg.p, cfg.t
        state = [[] for _ in range(tt)]                      # linear combinations; [] = 0
        k = 0
        while True:
            blk = inputs[k:k + cfg.rate]
            for i, lc in enumerate(blk):
                state[1 + i] = state[1 + i] + lc
            k += len(blk)
            if k >= len(inputs):
                break
            state = self._poseidon_permute(t, cfg, state, nxt)
        state = self._poseidon_permute(t, cfg, state, nxt)
        return state[1][0][1]

    def _poseidon_permute(self, t, cfg, state, nxt):
        half = cfg.rf // 2
        for r in range(cfg.rf + cfg.rp):
            full = r < half or r >= half + cfg.rp
            y = [state[i] + [(cfg.ark[r][i], ONE)] for i in range(cfg.t)]
            for i in range(cfg.t if full else 1):
                u = y[i]
                prev_col = None
                n_chain = 3 if cfg.alpha == 5 else 5
                for step in range(n_chain):
                    c = nxt()
                    if step == 0:
                        t.big_row(u, u, [(1, c)])                                  # u^2
                    elif step < n_chain - 1:
                        t.big_row([(1, prev_col)], [(1, prev_col)], [(1, c)])      # squarings
                    else:
                        t.big_row([(1, prev_col)], u, [(1, c)])                    # x^(alpha-1) * u
                    prev_col = c
                y[i] = [(1, prev_col)]
            new = []
            for i in range(cfg.t):
                c = nxt()
                lc = [(cfg.mds[i][j] * coef % cfg.p, col) for j in range(cfg.t) for coef, col in y[j]]
                t.big_row(lc, [(1, ONE)], [(1, c)])
                new.append([(1, c)])
            state = new
        return state

    @staticmethod
    def _enforce_word_eq(t, w, z):
        if t.build:
            r = t._rows(32)
            o = np.ones(32, np.int64)
            t._emit("A", np.concatenate([r, r]), np.concatenate([w.cols, z.cols]), np.concatenate([o, -o]))
            t._emit("B", r, np.full(32, ONE), o)

    @staticmethod
    def _digest_field_terms(digest):
        """[(2^k, col)] with k the position `digest_to_fpvar` gives the bit: byte j (big-endian byte j of the digest),
        bit b (little-endian within the byte) -> 8 j + b."""
        terms = []
        for j in range(INNER_HASH_SIZE):
            w = digest[j // 4]
            shift = 8 * (3 - j % 4)                        # byte j%4 of a big-endian word
            for b in range(8):
                terms.append((1 << (8 * j + b), int(w.cols[shift + b])))
        return terms

    def _unpack_children(self, t, time_e, inp):
        """The two children hashes (values of the first two time-ordered entries) as 54 message bytes in 14 big-endian
        words; the last word carries the 0x80 padding marker in its free bytes."""
        ev = not t.build
        byte_cols, byte_vals, byte_vids = [], [], []
        for child in range(2):
            v_col = time_e[child][1]
            if ev:
                vals = inp["time"][child][1]
                raw = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), np.uint8).reshape(t.batch, 32)
            terms = []
            for j in range(INNER_HASH_SIZE):
                bv = raw[:, j].astype(np.uint32) if ev else None
                vid = t.input_value()
                cols = t.alloc_bits(np.arange(8), bv, True, vid)
                byte_cols.append(cols); byte_vals.append(bv); byte_vids.append(vid)
                terms += [(1 << (8 * j + b), int(cols[b])) for b in range(8)]
            t.big_row(terms, [(1, ONE)], [(1, v_col)])      # the unpacked bits re-pack to the `get` value
        words = []
        zero_vid = t._value(OP_CONST, imm=0)
        for k in range(14):
            cols = np.full(32, -1, np.int64)
            val = np.zeros(t.batch, np.uint32) if ev else None
            parts = []
            for bi in range(4):
                j = 4 * k + bi
                if j < 54:
                    cols[8 * (3 - bi):8 * (3 - bi) + 8] = byte_cols[j]
                    parts.append(byte_vids[j])
                    if ev:
                        val |= byte_vals[j] << np.uint32(8 * (3 - bi))
                else:
                    parts.append(zero_vid)
            marker = 0
            if k == 13:                                      # bytes 52, 53 then 0x80, 0x00: the marker bit is a pinned 1
                one_bit = t.alloc_bits(np.arange(1), np.ones(t.batch, np.uint32) if ev else None, False,
                                       t._value(OP_CONST, imm=1))
                t.big_row([(1, int(one_bit[0]))], [(1, ONE)], [(1, ONE)])
                cols[8 * 1 + 7] = one_bit[0]                 # byte 2 of the word = 0x80: its bit 7
                marker = 0x8000
                if ev:
                    val |= np.uint32(0x8000)
            first_ref = len(t.refs)
            t.refs += parts
            words.append(Word(cols, val, t._value(OP_PACK4, first_ref, 4, 0, marker)))
        return words

    # ---- MultiStageConstraintSynthesizer -------------------------------------------------------------------
    def total_num_stages(self):
        return 2

    def generate_constraints(self, stage, cs):
        z = self._setup_assignment()
        ni = self.N_INST
        cs.initialize_stage()
        if stage == 0:
            cs.witness_assignment.extend(z[ni:ni + self.n0])
        else:
            cs.instance_assignment.extend(z[1:ni])
            cs.witness_assignment.extend(z[ni + self.n0:])
            cs._n_constraints += self.n_c
        cs.finalize_stage()

    def _setup_assignment(self):
        if getattr(self, "_setup_z", None) is None:
            w = example_witness(self, seed=0)
            self._setup_z = self.assignment_ints(w)[0]
        return self._setup_z

    def csr(self, fc):
        if self._csr is None:
            self._csr = self.tape.csr(fc)
        return tuple((rp, col, val) for rp, col, val, _vi, _tab in self._csr)

    def qap_evaluate(self, t_pt):
        """instance_map_with_evaluation over the tape's rows (generator.rs:75-76)."""
        p = CURVE_PARAMS[self.curve]
        r, ni, n_c = self.r, self.N_INST, self.n_c
        m, log_m = 1, 0
        while m < n_c + ni:
            m *= 2
            log_m += 1
        w = pow(pow(p["gen"], (r - 1) >> p["two_adicity"], r), 1 << (p["two_adicity"] - log_m), r)
        zt = (pow(t_pt, m, r) - 1) % r
        from .cp_groth16 import _batch_inverse
        wi = [1] * m
        for i in range(1, m):
            wi[i] = wi[i - 1] * w % r
        den = _batch_inverse([m * (t_pt - x) % r for x in wi], r)
        u = [zt * x % r * d % r for x, d in zip(wi, den)]
        self.csr(self.fc)
        outs = []
        for (rp, col, _val, vidx, table) in self._csr:
            acc = [0] * self.n_v
            rp_l, col_l, vi_l = rp.tolist(), col.tolist(), vidx.tolist()
            for i in range(n_c):
                ui = u[i]
                for k in range(rp_l[i], rp_l[i + 1]):
                    acc[col_l[k]] += ui * table[vi_l[k]]
            outs.append([x % r for x in acc])
        a, b, c = outs
        for j in range(ni):
            a[j] = (a[j] + u[n_c + j]) % r
        return a, b, c, zt, m

    # ---- the workload interface bench.py / tests use for synthetic classes (workload.SyntheticSubcircuit) ----------
    def set_witness_seed(self, seed):
        self._cur = example_witness(self, seed=seed, entry_chal=0x1234567, tr_chal=0x7654321)
        self._cur_z = None

    def assignment_ints_current(self):
        if self._cur_z is None:
            self._cur_z = self.assignment_ints(self._cur)[0]
        return self._cur_z

    def full_assignment_bytes(self, cs=None):
        return self.assignment_bytes(self._cur)[0]

    def stage0_witness_bytes(self):
        z = self.assignment_ints_current()
        return self.fc.enc(z[self.N_INST:self.N_INST + self.n0])

    # ---- witness generation --------------------------------------------------------------------------------
    def witness_batch(self, inputs):
        """inputs: list of per-subcircuit dicts (see `example_witness`).  Runs the program in EVAL mode over the whole
        batch.  Returns (bits uint8 (batch, n_v), full-width {column: [ints]}, digests list of 32-byte strings)."""
        B = len(inputs)
        inp = dict(entry_chal=None, tr_chal=None)
        inp["entry_chal"], inp["tr_chal"] = inputs[0]["entry_chal"], inputs[0]["tr_chal"]
        assert all(i["entry_chal"] == inp["entry_chal"] and i["tr_chal"] == inp["tr_chal"] for i in inputs)
        for key, n in (("time", self.n_time), ("addr", self.n_addr)):
            inp[key] = [([i[key][k][0] for i in inputs], [i[key][k][1] for i in inputs]) for k in range(n)]
        inp["time_eval0"] = [i["time_eval0"] for i in inputs]
        inp["addr_eval0"] = [i["addr_eval0"] for i in inputs]
        inp["prev"] = ([i["prev"][0] for i in inputs], [i["prev"][1] for i in inputs])
        inp["path_sib"] = [i["path"][0] for i in inputs]
        inp["path_idx"] = [i["path"][1] for i in inputs]
        inp["sha_root"] = [i.get("sha_root", 0) for i in inputs]
        if self.kind in ("leaf", "padding"):
            leaves = np.frombuffer(b"".join(i["leaf"] for i in inputs), np.uint8).reshape(B, 64)
            inp["leaf_words"] = leaves.reshape(B, 16, 4).astype(np.uint32) @ np.array([1 << 24, 1 << 16, 1 << 8, 1], np.uint32)
        t = Tape(self.N_INST, batch=B)
        digest = self._program(t, inp)
        assert t.n_wit == self.n_wit
        bits = t.assignment_bits(self.n_v)
        full = {col: vals for col, vals in t.full_records}
        dig = np.stack([w.val for w in digest], axis=1)            # (B, 8) uint32 big-endian words
        digests = [b"".join(int(x).to_bytes(4, "big") for x in row) for row in dig]
        return bits, full, digests

    def assignment_ints(self, inputs):
        """Full assignments as Python ints (tests / setup)."""
        inputs = inputs if isinstance(inputs, list) else [inputs]
        bits, full, _ = self.witness_batch(inputs)
        out = []
        for b in range(len(inputs)):
            z = bits[b].astype(np.int64).tolist()
            z[1], z[2], z[3] = inputs[b]["entry_chal"], inputs[b]["tr_chal"], inputs[b]["root"]
            for col, vals in full.items():
                z[col] = int(vals[b]) % self.r
            out.append(z)
        return out

    def assignment_bytes(self, inputs):
        """Montgomery bytes of the full assignments, (batch, n_v * 32): bits through a two-entry table."""
        inputs = inputs if isinstance(inputs, list) else [inputs]
        bits, full, _ = self.witness_batch(inputs)
        fc = self.fc
        tab = fc.enc([0, 1]).reshape(2, fc.nb)
        out = tab[bits]                                           # (B, n_v, nb)
        for b in range(len(inputs)):
            cols = [1, 2, 3] + list(full.keys())
            vals = [inputs[b]["entry_chal"], inputs[b]["tr_chal"], inputs[b]["root"]] + [int(v[b]) for v in full.values()]
            out[b, cols] = fc.enc(vals).reshape(len(cols), fc.nb)
        return out.reshape(len(inputs), self.n_v * fc.nb)


def packed_assignments(circ, inputs):
    """For hk_assignment_from_bits: per subcircuit (bits uint8[n_v], full_cols uint32[k], full_vals Montgomery bytes)."""
    inputs = inputs if isinstance(inputs, list) else [inputs]
    bits, full, _ = circ.witness_batch(inputs)
    cols = np.array([1, 2, 3] + list(full.keys()), np.uint32)
    out = []
    for b, w in enumerate(inputs):
        vals = [w["entry_chal"], w["tr_chal"], w["root"]] + [int(v[b]) for v in full.values()]
        out.append((bits[b], cols, circ.fc.enc(vals)))
    return out


def node_hash_field(digest):
    """`digest_to_fpvar` on the host: first 27 bytes, byte j bit b -> 2^(8j + b): the little-endian integer."""
    return int.from_bytes(digest[:INNER_HASH_SIZE], "little")


def iterated_sha256(data, ns):
    d = data
    for _ in range(ns):
        d = hashlib.sha256(d).digest()
    return d


def example_witness(circ, seed=0, entry_chal=None, tr_chal=None):
    """A consistent input for one subcircuit of class `circ` (tests, setup): random leaf / children, the portal entries
    the program expects (parents: two gets then the set; leaves: placeholders then the set), random starting evals
    (1 for the first subcircuit), and the root the chain produces."""
    import random
    rnd = random.Random(seed)
    r = circ.r
    ech = entry_chal if entry_chal is not None else rnd.randrange(r)
    tr = tr_chal if tr_chal is not None else rnd.randrange(r)
    w = dict(entry_chal=ech, tr_chal=tr)
    n = circ.n_time
    time = [(rnd.randrange(1 << 20), 0) for _ in range(n)]          # placeholder gets of address 0's value 0
    if circ.kind in ("leaf", "padding"):
        w["leaf"] = bytes(64) if circ.kind == "padding" else bytes(rnd.randrange(256) for _ in range(64))
        out = node_hash_field(iterated_sha256(w["leaf"], circ.ns))
    else:
        kids = [bytes(rnd.randrange(256) for _ in range(INNER_HASH_SIZE)) for _ in range(2)]
        for k in range(2):
            time[k] = (rnd.randrange(1 << 20), int.from_bytes(kids[k], "little"))
        out = node_hash_field(iterated_sha256(kids[0] + kids[1], circ.ns))
    if circ.kind in ("leaf", "parent"):
        time[-1] = (rnd.randrange(1 << 20), out)
    w["time"] = time
    # an address-ordered slice that passes the step check: consecutive addresses equal or one apart, equal addresses carry
    # equal values; the previous subcircuit's last entry in front of it (padding (0, 0) before subcircuit 0)
    a0 = 0 if circ.first else rnd.randrange(1 << 20)
    prev = (a0, 0 if circ.first else rnd.randrange(r))
    addr, cur = [], prev
    for k in range(circ.n_addr):
        if k == 1 or rnd.random() < 0.4:
            nxt = cur                                            # same address, same value
        else:
            nxt = (cur[0] + 1, rnd.randrange(r))
        addr.append(nxt)
        cur = nxt
    w["addr"], w["prev"] = addr, prev
    w["time_eval0"] = 1 if circ.first else rnd.randrange(1, r)
    w["addr_eval0"] = 1 if circ.first else rnd.randrange(1, r)
    if circ.kind == "root":
        w["sha_root"] = out
    if circ.last:
        # the permutation check needs equal final evals: the address-ordered slice is the time-ordered one sorted - with
        # addresses that pass the step check - and the starts are equal
        time = [(prev[0] + 1 + k, v) for k, (_a, v) in enumerate(time)]      # strictly increasing: no value constraints
        w["time"] = time
        w["addr"] = list(time)
        w["addr_eval0"] = w["time_eval0"]
    # the execution leaf this subcircuit produces and a random membership path for it; the root follows
    ech, tr = w["entry_chal"], w["tr_chal"]
    step = lambda c, e: c * ((tr - (e[1] + ech * e[0])) % r) % r
    te, ae = w["time_eval0"], w["addr_eval0"]
    for e in w["time"]:
        te = step(te, e)
    for e in w["addr"]:
        ae = step(ae, e)
    leaf = [te, ae, w["addr"][-1][0] % r, w["addr"][-1][1] % r]
    sib = [rnd.randrange(r) for _ in range(circ.depth)]
    idx = rnd.randrange(1 << circ.depth)
    w["path"] = (sib, idx)
    w["root"] = poseidon_path_root(circ.leaf_cfg, circ.node_cfg, leaf, sib, idx)
    return w


# ---------------------------------------------------------------------------------------------------------------------
class ShaMerkleJob:
    """Witness generation for a WHOLE big-merkle job (what `MerkleTreeCircuit::get_portal_subtraces` +
    `generate_constraints(0..n)` produce in the reference, tree_hash_circuit.rs:313-398,400-470): the tree of iterated
    SHA-256 hashes over `n/2` leaves, the ROM trace of every `set` / `get`, its address-sorted copy, the slice of both
    each subcircuit commits to in stage 0, and the running evaluations that thread through the subcircuits.

    Subcircuit order = the reference's (`subcircuit_idx_to_node_idx`): leaves, then parents level by level, the root at
    n - 2, the padding subcircuit at n - 1.  Time-ordered operations of a subcircuit (n_portals of them):
        leaf     placeholder gets ..., set(node hash)            parent   get(left), get(right), placeholders ..., set
        root     get(left), get(right), placeholders ...         padding  placeholders ...
    address 0 = the placeholder portal (value 0), address 1 + j = the hash of the node proved by subcircuit j."""

    def __init__(self, curve, n_subcircuits, ns, n_portals, leaves, entry_chal=None, tr_chal=None):
        """Without challenges only the stage-0 side exists (traces, `stage0_ints`); `set_challenges` - called by the
        coordinator once the stage-0 commitments are in (coordinator.rs:315-352) - adds the running evaluations."""
        n = n_subcircuits
        assert n >= 4 and n & (n - 1) == 0 and len(leaves) == n // 2 and n_portals >= 3
        self.curve, self.n, self.ns, self.np_ = curve, n, ns, n_portals
        self.r = CURVE_PARAMS[curve]["r"]
        nl = n // 2
        # node j (subcircuit order) -> children; hashes
        self.kind = ["leaf"] * nl + ["parent"] * (n - 2 - nl) + ["root", "padding"]
        self.children = {}
        level_start, width, j = 0, nl, nl
        while width > 1:
            for k in range(width // 2):
                self.children[j] = (level_start + 2 * k, level_start + 2 * k + 1)
                j += 1
            level_start += width
            width //= 2
        assert j == n - 1
        self.leaves = list(leaves)
        self.digest = [None] * n
        for i in range(nl):
            self.digest[i] = iterated_sha256(self.leaves[i], ns)
        for jj in range(nl, n - 1):
            l, rr = self.children[jj]
            self.digest[jj] = iterated_sha256(self.digest[l][:INNER_HASH_SIZE] + self.digest[rr][:INNER_HASH_SIZE], ns)
        self.digest[n - 1] = iterated_sha256(bytes(64), ns)
        self.sha_root = node_hash_field(self.digest[n - 2])            # the data tree's root hash (a witness of the root class)
        self.root = None                                               # the EXECUTION tree's root: known after `set_challenges`
        self.depth = n.bit_length() - 1
        val = lambda jj: node_hash_field(self.digest[jj])
        # time-ordered trace
        self.time = []
        for idx in range(n):
            ops = []
            if self.kind[idx] in ("parent", "root"):
                l, rr = self.children[idx]
                ops += [(1 + l, val(l)), (1 + rr, val(rr))]
            n_set = 1 if self.kind[idx] in ("leaf", "parent") else 0
            ops += [(0, 0)] * (n_portals - len(ops) - n_set)
            if n_set:
                ops.append((1 + idx, val(idx)))
            assert len(ops) == n_portals
            self.time.append(ops)
        flat = [e for ops in self.time for e in ops]
        order = sorted(range(len(flat)), key=lambda k: (flat[k][0], k))
        srt = [flat[k] for k in order]
        self.addr = [srt[idx * n_portals:(idx + 1) * n_portals] for idx in range(n)]
        self.entry_chal = self.tr_chal = None
        if entry_chal is not None:
            self.set_challenges(entry_chal, tr_chal)

    def stage0_ints(self, idx):
        """The subcircuit's stage-0 witness (what `process_stage0_request` commits to, worker.rs:91-146): (addr, val) of
        its time-ordered then its address-ordered entries, the variable order of `ShaMerkleSubcircuit._program`."""
        return [x for e in self.time[idx] for x in e] + [x for e in self.addr[idx] for x in e]

    def set_challenges(self, entry_chal, tr_chal):
        """Running evaluations entering every subcircuit (coordinator.rs:125-160 `generate_exec_tree`)."""
        n = self.n
        self.entry_chal, self.tr_chal = entry_chal % self.r, tr_chal % self.r
        r, ech, tr = self.r, self.entry_chal, self.tr_chal
        step = lambda cur, e: cur * ((tr - (e[1] + ech * e[0])) % r) % r
        self.time_eval0, self.addr_eval0 = [1], [1]
        for idx in range(n):
            t, a = self.time_eval0[-1], self.addr_eval0[-1]
            for e in self.time[idx]:
                t = step(t, e)
            for e in self.addr[idx]:
                a = step(a, e)
            self.time_eval0.append(t)
            self.addr_eval0.append(a)
        assert self.time_eval0[-1] == self.addr_eval0[-1]          # same multiset: the permutation check will hold
        # the execution tree (coordinator.rs:125-174): leaf i = (evals after subcircuit i, last entry of its address-ordered
        # slice); every subcircuit gets the membership path of its own leaf (coordinator.rs:446-452)
        from .poseidon import ExecTree
        leaves = [[self.time_eval0[i + 1], self.addr_eval0[i + 1], self.addr[i][-1][0] % r, self.addr[i][-1][1] % r]
                  for i in range(n)]
        self.tree = ExecTree(self.curve, leaves)
        self.root = self.tree.root

    def class_of(self, idx):
        """(kind, first, last) - the proving-key class a subcircuit needs (5 classes, tree_hash_circuit.rs:192-216)."""
        return self.kind[idx], idx == 0, idx == self.n - 1

    def make_class(self, idx):
        kind, first, last = self.class_of(idx)
        return ShaMerkleSubcircuit(self.curve, kind, self.ns, self.np_, first=first, last=last, depth=self.depth)

    def inputs(self, idx):
        """What the subcircuit's Stage1Request carries (coordinator.rs:569-604): challenges, the previous leaf (evals and
        last entry; padding before subcircuit 0), the membership path of its own leaf, the root."""
        w = dict(entry_chal=self.entry_chal, tr_chal=self.tr_chal, root=self.root, time=self.time[idx], addr=self.addr[idx],
                 time_eval0=self.time_eval0[idx], addr_eval0=self.addr_eval0[idx],
                 prev=(self.addr[idx - 1][-1] if idx else (0, 0)), path=self.tree.path(idx), sha_root=self.sha_root)
        if self.kind[idx] == "leaf":
            w["leaf"] = self.leaves[idx]
        elif self.kind[idx] == "padding":
            w["leaf"] = bytes(64)
        return w


# ---------------------------------------------------------------------------------------------------------------------
# The word program: what the GPU interprets (csrc/witness.cuh `k_word_program`), restated in numpy for the CPU tests.
def _ref_val(values, ref):
    v = values[ref & 0xfffff]
    rot, shr = (ref >> 20) & 31, (ref >> 25) & 31
    if shr:
        return v >> np.uint32(shr)
    if rot:
        return (v >> np.uint32(rot)) | (v << np.uint32(32 - rot))
    return v


def run_word_program(ops, refs, n_values, inputs):
    """inputs: uint32 (batch, n_inputs).  Returns the value table uint32 (n_values, batch)."""
    B = inputs.shape[0]
    values = np.zeros((n_values, B), np.uint32)
    vid = 0
    rot = lambda v, k: (v >> np.uint32(k)) | (v << np.uint32(32 - k))
    for op, a, b, c, imm in ops[:, :5].tolist():
        if op == OP_INPUT:
            values[vid] = inputs[:, imm]
        elif op == OP_CONST:
            values[vid] = imm
        elif op == OP_XOR:
            values[vid] = _ref_val(values, a) ^ _ref_val(values, b)
        elif op == OP_CH:
            e, f, g = _ref_val(values, a), _ref_val(values, b), _ref_val(values, c)
            values[vid] = (e & f) ^ (~e & g)
        elif op == OP_AND:
            values[vid] = _ref_val(values, a) & _ref_val(values, b)
        elif op == OP_MAJ:
            x, y, z = _ref_val(values, a), _ref_val(values, b), _ref_val(values, c)
            values[vid] = (x & y) ^ (x & z) ^ (y & z)
        elif op == OP_ADD:
            tot = np.full(B, imm, np.uint64)
            for k in range(b):
                tot = tot + _ref_val(values, int(refs[a + k])).astype(np.uint64)
            values[vid] = (tot & np.uint64(0xffffffff)).astype(np.uint32)
            values[vid + 1] = (tot >> np.uint64(32)).astype(np.uint32)
            vid += 1
        elif op == OP_PACK4:
            p = [values[int(refs[a + k]) & 0xfffff] for k in range(4)]
            values[vid] = (p[0] << np.uint32(24)) | (p[1] << np.uint32(16)) | (p[2] << np.uint32(8)) | p[3] | np.uint32(imm)
        elif op == OP_SHA_ROUND:
            ra, rb, rc, rd, re, rf, rg, rh, rw = (_ref_val(values, int(refs[a + k])) for k in range(9))
            x = rot(re, 6) ^ rot(re, 11)
            s1 = x ^ rot(re, 25)
            chv = (re & rf) ^ (~re & rg)
            y = rot(ra, 2) ^ rot(ra, 13)
            s0 = y ^ rot(ra, 22)
            t_ab = ra & rb
            mj = (ra & rb) ^ (ra & rc) ^ (rb & rc)
            u64 = lambda *ws: sum((w_.astype(np.uint64) for w_ in ws), np.full(B, imm, np.uint64))
            te, ta = u64(rd, rh, s1, chv, rw), u64(rh, s1, chv, rw, s0, mj)
            lo, hi = (lambda v: (v & np.uint64(0xffffffff)).astype(np.uint32)), (lambda v: (v >> np.uint64(32)).astype(np.uint32))
            for j, v in enumerate((x, s1, chv, y, s0, t_ab, mj, lo(te), hi(te), lo(ta), hi(ta))):
                values[vid + j] = v
            vid += 10
        elif op == OP_SHA_SCHED:
            w15, w2, w7, w16 = (_ref_val(values, int(refs[a + k])) for k in range(4))
            x0 = rot(w15, 7) ^ rot(w15, 18)
            s0 = x0 ^ (w15 >> np.uint32(3))
            x1 = rot(w2, 17) ^ rot(w2, 19)
            s1 = x1 ^ (w2 >> np.uint32(10))
            tot = s1.astype(np.uint64) + w7.astype(np.uint64) + s0.astype(np.uint64) + w16.astype(np.uint64)
            for j, v in enumerate((x0, s0, x1, s1, (tot & np.uint64(0xffffffff)).astype(np.uint32), (tot >> np.uint64(32)).astype(np.uint32))):
                values[vid + j] = v
            vid += 5
        vid += 1
    assert vid == n_values
    return values


def program_inputs(circ, inputs):
    """The per-subcircuit uint32 inputs of the class's word program: 16 big-endian leaf words, or the 54 bytes of the
    two children hashes."""
    if circ.kind in ("leaf", "padding"):
        leaves = np.frombuffer(b"".join(i["leaf"] for i in inputs), np.uint8).reshape(len(inputs), 64)
        return (leaves.reshape(len(inputs), 16, 4).astype(np.uint32) @ np.array([1 << 24, 1 << 16, 1 << 8, 1], np.uint32)).astype(np.uint32)
    out = np.zeros((len(inputs), 2 * INNER_HASH_SIZE), np.uint32)
    for b, w in enumerate(inputs):
        raw = b"".join(int(w["time"][k][1]).to_bytes(32, "little")[:INNER_HASH_SIZE] for k in range(2))
        out[b] = np.frombuffer(raw, np.uint8)
    return out


def full_values(circ, inputs):
    """(columns uint32[k], Montgomery bytes (batch, k * 32)) of the full-width variables: the three instance values,
    the portal entries and the running-evaluation chain - ~40 field values per subcircuit, computed on the host."""
    r = circ.r
    ni = circ.N_INST
    cols = [1, 2, 3]
    rows = []
    for w in inputs:
        ech, tr = w["entry_chal"], w["tr_chal"]
        vals = [ech, tr, w["root"]]
        col = ni
        c_local = []
        for key in ("time", "addr"):
            for a, v in w[key]:
                vals += [a, v]; c_local += [col, col + 1]; col += 2
        for key, e0 in (("time", w["time_eval0"]), ("addr", w["addr_eval0"])):
            vals.append(e0); c_local.append(col); col += 1
            cur = e0
            for a, v in w[key]:
                e = (v + ech * a) % r
                cur = cur * ((tr - e) % r) % r
                vals += [e, cur]; c_local += [col, col + 1]; col += 2
        vals += [w["prev"][0], w["prev"][1]]; c_local += [col, col + 1]; col += 2
        chain = [w["prev"]] + list(w["addr"])
        for k in range(1, len(chain)):
            d = (chain[k][0] - chain[k - 1][0]) % r
            vals += [pow(d, -1, r) if d else 0, 0 if d else 1]; c_local += [col, col + 1]; col += 2
        assert col == circ.pos_col0                       # the membership block follows: k_poseidon_path fills it
        if circ.kind == "root":
            vals.append(w["sha_root"]); c_local.append(circ.sha_root_col)
        rows.append(circ.fc.enc(vals))
    return np.array(cols + c_local, np.uint32), np.stack(rows)


def poseidon_inputs(circ, inputs):
    """Per subcircuit, what `hk_poseidon_path` needs to fill the membership block on the device: the execution leaf
    (4 field values: the final evals and the last address-ordered entry), the path's siblings and the leaf index."""
    r = circ.r
    leaves, sibs, idx = [], [], []
    for w in inputs:
        ech, tr = w["entry_chal"], w["tr_chal"]
        te, ae = w["time_eval0"], w["addr_eval0"]
        for a, v in w["time"]:
            te = te * ((tr - (v + ech * a)) % r) % r
        for a, v in w["addr"]:
            ae = ae * ((tr - (v + ech * a)) % r) % r
        leaves.append(circ.fc.enc([te, ae, w["addr"][-1][0] % r, w["addr"][-1][1] % r]))
        sibs.append(circ.fc.enc(list(w["path"][0])))
        idx.append(w["path"][1])
    return np.stack(leaves), np.stack(sibs), np.array(idx, np.uint32)
