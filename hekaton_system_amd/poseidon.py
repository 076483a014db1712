"""Poseidon over the scalar field, as the reference's execution tree uses it.

What the reference does (distributed-prover/src/poseidon_util.rs:26-107, eval_tree.rs:53-101, coordinator.rs:125-174,
subcircuit_circuit.rs:233-260): leaf i of a Merkle tree is `(time_eval, addr_eval, last_addr, last_val)` after subcircuit
i; leaves are hashed with `poseidon::CRH` at rate 3 (alpha 5, 8 full + 56 partial rounds), inner nodes with
`poseidon::TwoToOneCRH` at rate 2 (alpha 17, 8 + 31 rounds); every subcircuit proves the membership of ITS leaf under the
public root.  The parameters come from `find_poseidon_ark_and_mds` (Grain LFSR), the sponge from `PoseidonSponge`.

ark-crypto-primitives (0.4, git-pinned by the reference) is not on this machine, so `find_poseidon_ark_and_mds`,
`PoseidonSponge::permute` and the sponge's absorb / squeeze discipline are restated.  Pinned by published known answers
(tests/test_poseidon.py): the Grain LFSR and the permutation reproduce circomlib's BN254 `x^5`, t = 3, (8 full, 57
partial) instance - first round constants, first MDS entry and poseidon([1, 2]) =
7853200120776062878684798364095072458815029376092732009249414926327459813530 - which the same procedure generates.
PARITY UNPINNED: the reference's own instances (alpha 17 at rate 2; 56 partial rounds at rate 3) have no published
vectors, and the sponge's state layout (capacity element at index 0, output = state[1], one permutation per full block
and one before the squeeze) is remembered from ark's source, not checked against it.  The circuit gadget
(sha_circuit.py) and the GPU witness kernel (csrc/witness.cuh `k_poseidon_path`) are checked against THIS module bit for
bit.
"""
from functools import lru_cache


class GrainLFSR:
    """`PoseidonGrainLFSR` of ark-crypto-primitives (sponge/poseidon/grain_lfsr.rs), i.e. the Poseidon paper's
    `generate_parameters_grain`: 80-bit state seeded with the field / S-box / size description, 160 warm-up updates,
    then bits taken through the "keep the second bit of a pair when the first is 1" filter."""

    def __init__(self, is_sbox_inverse, prime_bits, state_len, full_rounds, partial_rounds):
        st = [0] * 80
        st[1] = 1                                           # b0, b1 = (0, 1): prime field
        st[5] = 1 if is_sbox_inverse else 0                 # b2..b5: the S-box
        def put(lo, hi, v):
            for i in range(hi, lo - 1, -1):
                st[i] = v & 1
                v >>= 1
        put(6, 17, prime_bits)
        put(18, 29, state_len)
        put(30, 39, full_rounds)
        put(40, 49, partial_rounds)
        for i in range(50, 80):
            st[i] = 1
        self.st, self.head, self.prime_bits = st, 0, prime_bits
        for _ in range(160):
            self._update()

    def _update(self):
        s, h = self.st, self.head
        b = s[(h + 62) % 80] ^ s[(h + 51) % 80] ^ s[(h + 38) % 80] ^ s[(h + 23) % 80] ^ s[(h + 13) % 80] ^ s[h]
        s[h] = b
        self.head = (h + 1) % 80
        return b

    def bits(self, n):
        out = []
        for _ in range(n):
            b = self._update()
            while b == 0:
                self._update()                              # discard the partner of a 0
                b = self._update()
            out.append(self._update())
        return out

    def _int(self):
        v = 0
        for b in self.bits(self.prime_bits):                # first bit = most significant
            v = (v << 1) | b
        return v

    def field_elements_rejection_sampling(self, n, p):
        out = []
        while len(out) < n:
            v = self._int()
            if v < p:
                out.append(v)
        return out

    def field_elements_mod_p(self, n, p):
        return [self._int() % p for _ in range(n)]


def find_poseidon_ark_and_mds(p, prime_bits, rate, full_rounds, partial_rounds, skip_matrices=0):
    """(ark[rounds][rate + 1], mds[rate + 1][rate + 1]) - ark-crypto-primitives sponge/poseidon/mod.rs."""
    t = rate + 1
    lfsr = GrainLFSR(False, prime_bits, t, full_rounds, partial_rounds)
    ark = [lfsr.field_elements_rejection_sampling(t, p) for _ in range(full_rounds + partial_rounds)]
    for _ in range(skip_matrices):
        lfsr.field_elements_mod_p(2 * t, p)
    xs = lfsr.field_elements_mod_p(t, p)
    ys = lfsr.field_elements_mod_p(t, p)
    mds = [[pow((xs[i] + ys[j]) % p, -1, p) for j in range(t)] for i in range(t)]
    return ark, mds


class PoseidonConfig:
    """`PoseidonConfig` of one rate: full_rounds, partial_rounds, alpha, ark, mds, rate, capacity = 1."""

    def __init__(self, p, prime_bits, rate, alpha, full_rounds, partial_rounds):
        self.p, self.rate, self.alpha, self.rf, self.rp = p, rate, alpha, full_rounds, partial_rounds
        self.t = rate + 1
        self.ark, self.mds = find_poseidon_ark_and_mds(p, prime_bits, rate, full_rounds, partial_rounds)

    def sbox_chain(self):
        """Exponents of the multiplication chain the gadget and the witness kernel use for x^alpha: 5 -> (2, 4, 5),
        17 -> (2, 4, 8, 16, 17): squarings, then one product with x."""
        assert self.alpha in (5, 17)
        return (2, 4, 5) if self.alpha == 5 else (2, 4, 8, 16, 17)

    def permute(self, state, trace=None):
        """`PoseidonSponge::permute`: rf / 2 full rounds, rp partial rounds (S-box on element 0 only), rf / 2 full rounds;
        a round = add round constants, S-box, MDS.  trace (list): receives, per round, the S-box chain values of every
        S-boxed element followed by the new state - the witness order of the circuit gadget."""
        p, t = self.p, self.t
        s = list(state)
        half = self.rf // 2
        for r in range(self.rf + self.rp):
            full = r < half or r >= half + self.rp
            y = [(s[i] + self.ark[r][i]) % p for i in range(t)]
            for i in range(t if full else 1):
                u = y[i]
                x2 = u * u % p
                x4 = x2 * x2 % p
                if self.alpha == 5:
                    chain = [x2, x4, x4 * u % p]
                else:
                    x8 = x4 * x4 % p
                    x16 = x8 * x8 % p
                    chain = [x2, x4, x8, x16, x16 * u % p]
                if trace is not None:
                    trace.extend(chain)
                y[i] = chain[-1]
            s = [sum(self.mds[i][j] * y[j] for j in range(t)) % p for i in range(t)]
            if trace is not None:
                trace.extend(s)
        return s

    def crh(self, inputs, trace=None):
        """`poseidon::CRH::evaluate`: a fresh sponge (state 0, capacity element at index 0), absorb, squeeze one element.
        Absorbing adds up to `rate` inputs to state[1..], permuting between full blocks; the squeeze permutes once more and
        returns state[1]."""
        s = [0] * self.t
        k = 0
        inputs = list(inputs)
        while True:
            blk = inputs[k:k + self.rate]
            for i, v in enumerate(blk):
                s[1 + i] = (s[1 + i] + v) % self.p
            k += len(blk)
            if k >= len(inputs):
                break
            s = self.permute(s, trace)
        s = self.permute(s, trace)
        return s[1]


@lru_cache(maxsize=None)
def merkle_params(curve):
    """`gen_merkle_params()` (poseidon_util.rs:102-107): leaf hash = rate 3, two-to-one hash = rate 2, from the
    `optimized_for_weights = false` table (poseidon_util.rs:53-62: (2, 17, 8, 31), (3, 5, 8, 56))."""
    from .cp_groth16 import CURVE_PARAMS
    p = CURVE_PARAMS[curve]["r"]
    bits = p.bit_length()
    return PoseidonConfig(p, bits, 3, 5, 8, 56), PoseidonConfig(p, bits, 2, 17, 8, 31)


def device_params(curve, fc):
    """(Montgomery bytes of both instances' constants, their count, leaf descriptor, node descriptor) for
    hk_poseidon_path: per instance ark[(rf + rp)][t] then mds[t][t]; descriptor = (t, alpha, rf, rp, offset)."""
    leaf, node = merkle_params(curve)
    vals, descs = [], []
    for c in (leaf, node):
        descs.append((c.t, c.alpha, c.rf, c.rp, len(vals)))
        vals += [v for row in c.ark for v in row] + [v for row in c.mds for v in row]
    return fc.enc(vals), len(vals), descs[0], descs[1]


class ExecTree:
    """The Poseidon Merkle tree over the execution leaves (coordinator.rs:125-174 `generate_exec_tree`; ark
    `MerkleTree::new` + `generate_proof`): leaf digest = CRH(leaf fields), inner = TwoToOneCRH(left, right).
    `path(i)` = (siblings bottom-up: the leaf's sibling digest first, then one inner digest per level; index i) - what a
    Stage1Request carries as `next_leaf_membership` (coordinator.rs:446-452)."""

    def __init__(self, curve, leaves):
        n = len(leaves)
        assert n >= 2 and n & (n - 1) == 0
        self.leaf_cfg, self.node_cfg = merkle_params(curve)
        self.leaves = [list(l) for l in leaves]
        level = [self.leaf_cfg.crh(l) for l in self.leaves]
        self.levels = [level]
        while len(level) > 1:
            level = [self.node_cfg.crh([level[2 * k], level[2 * k + 1]]) for k in range(len(level) // 2)]
            self.levels.append(level)
        self.root = level[0]
        self.depth = len(self.levels) - 1

    def path(self, i):
        sib, k = [], i
        for lvl in self.levels[:-1]:
            sib.append(lvl[k ^ 1])
            k >>= 1
        return sib, i

    def verify(self, leaf, sib, index):
        cur = self.leaf_cfg.crh(leaf)
        for l, s in enumerate(sib):
            bit = (index >> l) & 1
            cur = self.node_cfg.crh([s, cur] if bit else [cur, s])
        return cur == self.root
