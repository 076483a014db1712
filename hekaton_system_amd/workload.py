"""Synthetic subcircuit workloads shaped like the reference's circuits (SURVEY.md §8 size table, §8d).

The reference synthesises its subcircuits with ark-r1cs-std gadgets (SHA-256, Poseidon; third-party,
absent) inside `SubcircuitWithPortalsProver` (distributed-prover/src/subcircuit_circuit.rs:134-277).
That synthesis is outside the hot path; what the hot path consumes is its OUTPUT: a two-stage R1CS
(stage 0 = the two portal subtraces, stage 1 = everything else), static per proving-key class, plus
one assignment per subcircuit.  `SyntheticSubcircuit` produces exactly that shape:

  * n_inst = 4 instance variables (1, entry_chal, tr_chal, root — subcircuit_circuit.rs:174-179)
  * stage 0: n0 full-width witnesses (ROM entry = 2 witnesses, rom_transcript.rs:286-306)
  * stage 1: free witnesses (85 % bits / 15 % full-width) and one gate-output witness per constraint.
    85 % of the constraints are boolean gates, as in a SHA-256 circuit — XOR `(2a)·(b) = a + b − p` and AND / NAND
    `(a | 1−a)·(b | 1−b) = p | 1−p` — whose operands are drawn uniformly from ALL earlier boolean variables (free bits
    and earlier gate outputs), so gate outputs stay bits (p(1) ≈ ½ through the XORs) at any depth; 15 % are
    full-width rows `(3 terms)·(2 terms) = p` over ALL earlier variables.  Every assignment satisfies every
    row, so (ab − c)/Z is a polynomial and real proofs verify.
  * consequence for the proving key: the A / B queries are dense (≈ 70 % / ≈ 65 % non-infinity bases at the
    2^21 size; printed by `query_density()`), and the assignment is the SURVEY §8d mixture: ≈ 85 % of the
    scalars in {0, 1}, ≈ 15 % uniform in [0, r).
"""
import numpy as np

from .cp_groth16 import (CURVE_PARAMS, FrCodec, MultiStageConstraintSynthesizer, _batch_inverse,
                         SynthesisError)

# BASELINE.json configs -> (n_constraints, n_free witnesses incl. stage 0, n0)   [estimates, SURVEY §8]
CONFIGS = {
    "big-merkle-4x1": dict(n_c=61_000, n_free=4_000, n0=16),          # config 0: m = 2^16
    "big-merkle-64x32": dict(n_c=1_250_000, n_free=50_000, n0=16),    # config 1: m = 2^21, n_v ~ 1.3e6
    "big-merkle-512x64": dict(n_c=2_500_000, n_free=100_000, n0=16),  # config 2: m = 2^22, n_v ~ 2.6e6
    # config 3: the per-subcircuit size of vkd is unknown offline ("placeholder 2^17; must be measured");
    # flagged placeholder m = 2^17, n0 = 8 192 ("portal-heavy"), 7 proving-key classes
    "vkd-256": dict(n_c=120_000, n_free=12_000, n0=8_192),
    "vm-1024x1024": dict(n_c=1_000_000, n_free=700_000, n0=217_280),  # config 4: m = 2^20, n_v ~ 1.7e6
    "tiny": dict(n_c=1_000, n_free=100, n0=16),
}

# circuit family and number of subcircuits of every BASELINE config: which proving-key class a subcircuit index
# uses is the family's `representative_subcircuit` (below)
FAMILIES = {
    "big-merkle-4x1": ("big-merkle", 4), "big-merkle-64x32": ("big-merkle", 64),
    "big-merkle-512x64": ("big-merkle", 512), "vkd-256": ("vkd", 256), "vm-1024x1024": ("vm", 1024),
    "tiny": ("big-merkle", 8),
}


def unique_subcircuits(family, n):
    """`CircuitWithPortals::get_unique_subcircuits`: one representative index per proving-key class.
    big-merkle: tree_hash_circuit.rs:192-197; vkd: vkd/vkd_constraints.rs:199-201; vm: vm/vm_constraints.rs:91-93."""
    if family == "big-merkle":
        out = [0, 1, n - 1, n - 2, n - 3]
    elif family == "vkd":
        out = [0, 6, 7, 8, 10, 19, n - 1]
    elif family == "vm":
        out = [0, 1]
    else:
        raise KeyError(family)
    seen = []
    for x in out:                      # tiny trees (n = 4) name the same index twice
        if x not in seen:
            seen.append(x)
    return seen


def representative_subcircuit(family, n, idx):
    """`CircuitWithPortals::representative_subcircuit`: subcircuit index -> index of its class representative.
    big-merkle: tree_hash_circuit.rs:200-216 (first leaf, other leaves, parents, root, padding);
    vkd: vkd/vkd_constraints.rs:203-214 over the layout `vkd_update_to_subcircuit` builds (vkd/vkd.rs:362-612:
    6 paddings, write-pp, one Append of 8 subcircuits, then Updates of 8 subcircuits each, final equality);
    vm: vm/vm_constraints.rs:95-97."""
    if not 0 <= idx < n:
        raise IndexError("subcircuit index out of range: %d" % idx)
    if family == "big-merkle":
        if idx == 0:
            return 0
        if 1 <= idx < n // 2:
            return 1
        if n // 2 <= idx <= n - 3:
            return n - 3
        return idx                                   # n - 1 (padding), n - 2 (root)
    if family == "vm":
        return 0 if idx == 0 else 1
    if family == "vkd":
        if idx < 6:
            return 0                                 # "padding"
        if idx == 6:
            return 6                                 # "write pp"
        if idx == n - 1:
            return n - 1                             # "equality"
        u, k = divmod(idx - 7, 8)
        if u == 0:                                   # the Append: 7 = hash leaf + get index + compute path,
            return {0: 7, 3: 10}.get(k, 8)           # 10 = compute path + equality, the rest plain compute path
        return 19 if k == 4 else 8                   # Updates: 5th = equality + hash leaf + compute path
    raise KeyError(family)


class SyntheticSubcircuit(MultiStageConstraintSynthesizer):
    N_INST = 4
    N_LAYERS = 16                     # rows of layer l only reference variables created before layer l
    KIND_WIDE, KIND_XOR, KIND_AND = 0, 1, 2

    def __init__(self, curve, n_c, n_free, n0, class_seed=0x48454B41544F4E31, bit_fraction=0.85):
        self.curve = curve
        self.r = CURVE_PARAMS[curve]["r"]
        self.n_c, self.n_free, self.n0 = n_c, n_free, n0
        self.n_wit = n_free + n_c
        self.n_v = self.N_INST + self.n_wit
        r = self.r
        self.wide_coeffs = [1, 1, r - 1, 2, r - 2, 1 << 7, 1 << 31, 1]
        rng = np.random.default_rng(class_seed)
        ni = self.N_INST
        self.is_bit_var = rng.random(n_free) < bit_fraction
        self.is_bit_var[:n0] = False
        if not self.is_bit_var.any():
            self.is_bit_var[min(n0, n_free - 1)] = True
        u = rng.random(n_c)
        self.kind = np.where(u < 1.0 - bit_fraction, self.KIND_WIDE,
                             np.where(u < 1.0 - bit_fraction + 0.6 * bit_fraction, self.KIND_XOR, self.KIND_AND)).astype(np.int8)
        self.neg = (rng.random((n_c, 2)) < 0.25) & (self.kind == self.KIND_AND)[:, None]   # AND operands may be 1 - x
        self.neg_out = (rng.random(n_c) < 0.5) & (self.kind == self.KIND_AND)               # NAND: a * b = 1 - p
        self.cols = np.zeros((n_c, 5), dtype=np.int64)          # bit rows use cols[:, 0] (A) and cols[:, 3] (B)
        self.coef_idx = rng.integers(0, 8, size=(n_c, 5)).astype(np.int64)                 # wide rows only
        base = ni + n_free
        self.layer_bounds = [n_c * l // self.N_LAYERS for l in range(self.N_LAYERS + 1)]
        bit_pool = [ni + np.nonzero(self.is_bit_var)[0]]
        wide_pool = [np.arange(1, ni), ni + np.nonzero(~self.is_bit_var)[0]]
        for l in range(self.N_LAYERS):
            lo, hi = self.layer_bounds[l], self.layer_bounds[l + 1]
            if hi == lo:
                continue
            pool = np.concatenate(bit_pool)
            rows = np.arange(lo, hi)
            wide = self.kind[rows] == self.KIND_WIDE
            n_avail = base + lo                                  # every variable created before this layer
            pick_bits = pool[rng.integers(0, len(pool), size=(hi - lo, 5))]
            pick_any = rng.integers(1, n_avail, size=(hi - lo, 5))
            wpool = np.concatenate(wide_pool)
            pick_wide = wpool[rng.integers(0, len(wpool), size=(hi - lo, 5))]
            lead = np.array([True, False, False, True, False])   # first A and first B operand of a wide row are wide
            self.cols[lo:hi] = np.where(wide[:, None], np.where(lead[None, :], pick_wide, pick_any), pick_bits)
            bit_pool.append(base + rows[~wide])
            wide_pool.append(base + rows[wide])
        self.seed = None
        self._z = None

    # ---- MultiStageConstraintSynthesizer ---------------------------------------------------------
    def total_num_stages(self):
        return 2

    def set_witness_seed(self, seed, instance=None):
        """Selects which subcircuit instance (which assignment) the next commit/prove is for.  `instance`: the job's
        public inputs (entry_chal, tr_chal, root - the same for every subcircuit of a job, which is what lets the
        aggregator combine the proofs, aggregation.rs:192-205); None = drawn from the seed."""
        self.seed = seed
        self.instance = None if instance is None else [int(v) % self.r for v in instance]
        self._z = None

    def generate_constraints(self, stage, cs):
        if self.seed is None:
            self.set_witness_seed(0)          # setup mode: any assignment of the right shape
        z = self.assignment_ints()
        ni = self.N_INST
        if stage == 0:
            cs.initialize_stage()
            cs.witness_assignment.extend(z[ni:ni + self.n0])
            cs.finalize_stage()
        else:
            cs.initialize_stage()
            cs.instance_assignment.extend(z[1:ni])
            cs.witness_assignment.extend(z[ni + self.n0:])
            cs._n_constraints += self.n_c
            cs.finalize_stage()

    # ---- static matrices ---------------------------------------------------------------------------
    def rows(self, i):
        """Row i as ark `ConstraintMatrices` rows: ([(coeff, col)] for A, B, C), coefficients as ints mod r."""
        r, k = self.r, int(self.kind[i])
        co = self.cols[i].tolist()
        p = self.N_INST + self.n_free + i
        if k == self.KIND_WIDE:
            wc, ci = self.wide_coeffs, self.coef_idx[i].tolist()
            return ([(wc[ci[0]], co[0]), (wc[ci[1]], co[1]), (wc[ci[2]], co[2])],
                    [(wc[ci[3]], co[3]), (wc[ci[4]], co[4])], [(1, p)])
        x0, x1 = co[0], co[3]
        if k == self.KIND_XOR:
            return [(2, x0)], [(1, x1)], [(1, x0), (1, x1), (r - 1, p)]
        a = [(1, 0), (r - 1, x0)] if self.neg[i, 0] else [(1, x0)]
        b = [(1, 0), (r - 1, x1)] if self.neg[i, 1] else [(1, x1)]
        return a, b, ([(1, 0), (r - 1, p)] if self.neg_out[i] else [(1, p)])

    def csr(self, fc):
        """The three matrices as (row_ptr u64, col u32, val Montgomery bytes) — built without a Python loop."""
        n_c, ni, r = self.n_c, self.N_INST, self.r
        kind, neg, cols = self.kind, self.neg, self.cols
        p = (ni + self.n_free + np.arange(n_c)).astype(np.int64)
        tab_vals = [1, 2, r - 1] + self.wide_coeffs               # value table: index 0: 1, 1: 2, 2: -1, 3..: wide
        tab = fc.enc(tab_vals).reshape(len(tab_vals), fc.nb)
        wide, xor, andk = kind == self.KIND_WIDE, kind == self.KIND_XOR, kind == self.KIND_AND

        def build(cnt, entries):
            """entries: list of (row mask, slot within the row, col array, value-table index array)"""
            row_ptr = np.zeros(n_c + 1, dtype=np.uint64)
            row_ptr[1:] = np.cumsum(cnt)
            nnz = int(row_ptr[-1])
            col = np.zeros(nnz, dtype=np.uint32)
            vidx = np.zeros(nnz, dtype=np.int64)
            start = row_ptr[:-1].astype(np.int64)
            for mask, slot, c, v in entries:
                pos = start[mask] + slot
                col[pos] = c[mask] if isinstance(c, np.ndarray) else c
                vidx[pos] = v[mask] if isinstance(v, np.ndarray) else v
            return row_ptr, col, np.ascontiguousarray(tab[vidx]).ravel()

        zero = np.zeros(n_c, dtype=np.int64)
        n0m, n1m = andk & neg[:, 0], andk & neg[:, 1]
        # A
        cntA = np.where(wide, 3, np.where(n0m, 2, 1))
        A = build(cntA, [(wide, 0, cols[:, 0], 3 + self.coef_idx[:, 0]), (wide, 1, cols[:, 1], 3 + self.coef_idx[:, 1]),
                         (wide, 2, cols[:, 2], 3 + self.coef_idx[:, 2]),
                         (xor, 0, cols[:, 0], 1), (andk & ~neg[:, 0], 0, cols[:, 0], 0),
                         (n0m, 0, zero, 0), (n0m, 1, cols[:, 0], 2)])
        cntB = np.where(wide, 2, np.where(n1m, 2, 1))
        B = build(cntB, [(wide, 0, cols[:, 3], 3 + self.coef_idx[:, 3]), (wide, 1, cols[:, 4], 3 + self.coef_idx[:, 4]),
                         (xor, 0, cols[:, 3], 0), (andk & ~neg[:, 1], 0, cols[:, 3], 0),
                         (n1m, 0, zero, 0), (n1m, 1, cols[:, 3], 2)])
        no = self.neg_out
        cntC = np.where(xor, 3, np.where(no, 2, 1))
        C = build(cntC, [(~xor & ~no, 0, p, 0), (no, 0, zero, 0), (no, 1, p, 2),
                         (xor, 0, cols[:, 0], 0), (xor, 1, cols[:, 3], 0), (xor, 2, p, 2)])
        return A, B, C

    def qap_evaluate(self, t):
        """instance_map_with_evaluation specialised to this circuit's static rows (generator.rs:75-76)."""
        p = CURVE_PARAMS[self.curve]
        r, ni, n_c = self.r, self.N_INST, self.n_c
        m, log_m = 1, 0
        while m < n_c + ni:
            m *= 2
            log_m += 1
        if log_m > p["two_adicity"]:
            raise SynthesisError("PolynomialDegreeTooLarge")
        w = pow(pow(p["gen"], (r - 1) >> p["two_adicity"], r), 1 << (p["two_adicity"] - log_m), r)
        zt = (pow(t, m, r) - 1) % r
        wi = [1] * m
        for i in range(1, m):
            wi[i] = wi[i - 1] * w % r
        den = _batch_inverse([m * (t - x) % r for x in wi], r)
        u = [zt * x % r * d % r for x, d in zip(wi, den)]
        a = [0] * self.n_v
        b = [0] * self.n_v
        c = [0] * self.n_v
        for j in range(ni):
            a[j] = u[n_c + j]
        cols = self.cols.tolist()
        cidx = self.coef_idx.tolist()
        kind = self.kind.tolist()
        neg = self.neg.tolist()
        neg_out = self.neg_out.tolist()
        wc = self.wide_coeffs
        base = ni + self.n_free
        for i in range(n_c):
            ui, k, co = u[i], kind[i], cols[i]
            if k == 1:                                   # XOR: (2 x0) * (x1) = x0 + x1 - p
                a[co[0]] += 2 * ui
                b[co[3]] += ui
                c[co[0]] += ui
                c[co[3]] += ui
                c[base + i] -= ui
            elif k == 2:                                 # AND with optional negations
                if neg[i][0]:
                    a[0] += ui
                    a[co[0]] -= ui
                else:
                    a[co[0]] += ui
                if neg[i][1]:
                    b[0] += ui
                    b[co[3]] -= ui
                else:
                    b[co[3]] += ui
                if neg_out[i]:
                    c[0] += ui
                    c[base + i] -= ui
                else:
                    c[base + i] += ui
            else:
                ci = cidx[i]
                a[co[0]] += ui * wc[ci[0]]
                a[co[1]] += ui * wc[ci[1]]
                a[co[2]] += ui * wc[ci[2]]
                b[co[3]] += ui * wc[ci[3]]
                b[co[4]] += ui * wc[ci[4]]
                c[base + i] += ui
        a = [x % r for x in a]
        b = [x % r for x in b]
        c = [x % r for x in c]
        return a, b, c, zt, m

    def query_density(self):
        """Fraction of variables with a non-zero column in A / in B (= non-infinity a_g / b_g, b_h bases, up to
        the measure-zero event that a column's QAP evaluation vanishes at the setup point)."""
        ina = np.zeros(self.n_v, dtype=bool)
        inb = np.zeros(self.n_v, dtype=bool)
        wide = self.kind == self.KIND_WIDE
        ina[self.cols[:, 0]] = True
        ina[self.cols[wide][:, 1:3].ravel()] = True
        inb[self.cols[:, 3]] = True
        inb[self.cols[wide][:, 4]] = True
        if self.neg[:, 0].any():
            ina[0] = True
        if self.neg[:, 1].any():
            inb[0] = True
        ina[:self.N_INST] = True                         # a[j] = u[n_c + j] for the instance variables
        return float(ina.mean()), float(inb.mean())

    # ---- per-subcircuit assignment ------------------------------------------------------------------
    def assignment_ints(self):
        """Full assignment instance || witness as ints (z[0] = 1), satisfying every row."""
        if self._z is not None:
            return self._z
        if self.seed is None:
            raise RuntimeError("set_witness_seed() first")
        import random
        r, ni, n_free, n_c = self.r, self.N_INST, self.n_free, self.n_c
        rng = np.random.default_rng(self.seed)
        prng = random.Random(int(self.seed))
        bits = np.zeros(self.n_v, dtype=np.int64)              # value of every boolean variable (others unused)
        free_bits = ni + np.nonzero(self.is_bit_var)[0]
        bits[free_bits] = rng.integers(0, 2, size=len(free_bits))
        z = [0] * self.n_v
        z[0] = 1
        for j in range(1, ni):
            z[j] = prng.randrange(r)
        if getattr(self, "instance", None) is not None:
            z[1:ni] = self.instance
        for j in np.nonzero(~self.is_bit_var)[0].tolist():
            z[ni + j] = prng.randrange(r)
        for j, v in zip(free_bits.tolist(), bits[free_bits].tolist()):
            z[j] = v
        base = ni + n_free
        wc = self.wide_coeffs
        cols, cidx, kind, neg = self.cols, self.coef_idx, self.kind, self.neg
        for l in range(self.N_LAYERS):
            lo, hi = self.layer_bounds[l], self.layer_bounds[l + 1]
            if hi == lo:
                continue
            k = kind[lo:hi]
            x0, x1 = bits[cols[lo:hi, 0]], bits[cols[lo:hi, 3]]
            av = np.where(neg[lo:hi, 0], 1 - x0, x0)
            bv = np.where(neg[lo:hi, 1], 1 - x1, x1)
            out = np.where(k == self.KIND_XOR, x0 ^ x1, np.where(self.neg_out[lo:hi], 1 - av * bv, av * bv))
            bit_rows = np.nonzero(k != self.KIND_WIDE)[0]
            bits[base + lo + bit_rows] = out[bit_rows]
            for i, v in zip((base + lo + bit_rows).tolist(), out[bit_rows].tolist()):
                z[i] = v
            for i in (lo + np.nonzero(k == self.KIND_WIDE)[0]).tolist():
                co, ci = cols[i].tolist(), cidx[i].tolist()
                az = z[co[0]] * wc[ci[0]] + z[co[1]] * wc[ci[1]] + z[co[2]] * wc[ci[2]]
                bz = z[co[3]] * wc[ci[3]] + z[co[4]] * wc[ci[4]]
                z[base + i] = az * bz % r
        self._z = z
        return self._z

    def full_assignment_bytes(self, cs=None):
        """Montgomery bytes of the full assignment (small values through a lookup table)."""
        z = self.assignment_ints()
        fc = FrCodec(self.curve)
        small_max = 64
        tab = fc.enc(list(range(small_max))).reshape(small_max, fc.nb)
        arr = np.zeros((len(z), fc.nb), dtype=np.uint8)
        small_idx, small_val, big_idx, big_val = [], [], [], []
        for i, x in enumerate(z):
            if x < small_max:
                small_idx.append(i); small_val.append(x)
            else:
                big_idx.append(i); big_val.append(x)
        arr[np.array(small_idx, dtype=np.int64)] = tab[np.array(small_val, dtype=np.int64)]
        if big_idx:
            arr[np.array(big_idx, dtype=np.int64)] = fc.enc(big_val).reshape(len(big_idx), fc.nb)
        return arr.ravel()

    def stage0_witness_bytes(self):
        z = self.assignment_ints()
        return FrCodec(self.curve).enc(z[self.N_INST:self.N_INST + self.n0])


# real SHA-256 subcircuits (sha_circuit.py): name -> (n_subcircuits, sha iterations, portals per subcircuit)
SHA_CONFIGS = {"big-merkle-sha-8x1": (8, 1, 4), "big-merkle-sha-64x32": (64, 32, 4)}
for _k, (_n, _ns, _np) in SHA_CONFIGS.items():
    FAMILIES[_k] = ("big-merkle", _n)


def make_sha_config(curve, name, class_rep, n_total=None):
    """The class of subcircuit `class_rep` of a real-SHA big-merkle job: first leaf, leaf, parent, root, padding
    (tree_hash_circuit.rs:192-216 order: leaves, parents level by level, root at n - 2, padding at n - 1)."""
    from .sha_circuit import ShaMerkleSubcircuit
    n, ns, n_portals = SHA_CONFIGS[name]
    n = n_total or n
    rep = 1 if class_rep is None else class_rep
    kind = "leaf" if rep < n // 2 else ("padding" if rep == n - 1 else ("root" if rep == n - 2 else "parent"))
    return ShaMerkleSubcircuit(curve, kind, ns, n_portals, first=(rep == 0), last=(rep == n - 1), depth=n.bit_length() - 1)


def make_config(curve, name, class_rep=None, n_total=None):
    """The synthetic subcircuit class of a BASELINE config.  `class_rep` (a representative subcircuit index from
    `unique_subcircuits`) selects which of the config's proving-key classes: same size, different matrices
    (class seed = base seed + representative index); None = the first class."""
    if name in SHA_CONFIGS:
        return make_sha_config(curve, name, class_rep, n_total)
    c = CONFIGS[name]
    seed = 0x48454B41544F4E31 + (0 if class_rep is None else int(class_rep))
    return SyntheticSubcircuit(curve, c["n_c"], c["n_free"], c["n0"], class_seed=seed)


def config_classes(name):
    """(family, n_subcircuits, [class representatives]) of a BASELINE config."""
    family, n = FAMILIES[name]
    return family, n, unique_subcircuits(family, n)


def prepare_class_host(job):
    """Worker-process half of a benchmark / test setup: builds one proving-key class's circuit, runs the host half of
    the trusted setup (`cp_groth16.setup_host`: toxic waste, QAP evaluation, scalar layout) and generates the
    assignments of `witness_seeds`.  Pure Python / numpy, no device: safe to run in a spawned process pool BEFORE the
    parent initialises HIP.  job = (curve, config name, class representative, setup seed bytes, [witness seeds]
    [, n_total [, job-wide public inputs]]);
    returns (class_rep, HostSetup, [(seed, full assignment bytes, stage-0 witness bytes)])."""
    from .cp_groth16 import SeededRng, setup_host
    curve, name, rep, seed, witness_seeds = job[:5]
    n_total = job[5] if len(job) > 5 else None
    instance = job[6] if len(job) > 6 else None
    circ = make_config(curve, name, rep, n_total)
    hs = setup_host(circ, curve, SeededRng(seed))
    assigns = []
    for ws in witness_seeds:
        if instance is not None and isinstance(circ, SyntheticSubcircuit):
            circ.set_witness_seed(ws, instance)
        else:
            circ.set_witness_seed(ws)
        assigns.append((ws, circ.full_assignment_bytes(), circ.stage0_witness_bytes()))
    return rep, hs, assigns


def prepare_classes_host(jobs, max_workers=None):
    """`prepare_class_host` over several classes in parallel (spawned workers; sequential when there is one)."""
    import os
    # under a profiler (rocprofv3 preloads its tool library, which initialises the GPU before Python starts) a worker
    # spawn would be an exec from a GPU-initialised process: stay in-process there
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or "HSA_TOOLS_LIB" in os.environ or \
        os.environ.get("HK_NO_SPAWN")
    if len(jobs) <= 1 or profiled:
        return [prepare_class_host(j) for j in jobs]
    import multiprocessing as mp
    n = min(len(jobs), max_workers or max(1, min(8, (os.cpu_count() or 2) // 2)))
    with mp.get_context("spawn").Pool(processes=n) as pool:
        return pool.map(prepare_class_host, jobs)
