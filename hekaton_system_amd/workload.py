"""Synthetic subcircuit workloads shaped like the reference's circuits (SURVEY.md §8 size table, §8d).

The reference synthesises its subcircuits with ark-r1cs-std gadgets (SHA-256, Poseidon; third-party,
absent) inside `SubcircuitWithPortalsProver` (distributed-prover/src/subcircuit_circuit.rs:134-277).
That synthesis is outside the hot path; what the hot path consumes is its OUTPUT: a two-stage R1CS
(stage 0 = the two portal subtraces, stage 1 = everything else), static per proving-key class, plus
one assignment per subcircuit.  `SyntheticSubcircuit` produces exactly that shape:

  * n_inst = 4 instance variables (1, entry_chal, tr_chal, root — subcircuit_circuit.rs:174-179)
  * stage 0: n0 full-width witnesses (ROM entry = 2 witnesses, rom_transcript.rs:286-306)
  * stage 1: free witnesses (85 % bits / 15 % full-width, SHA-like) and one product witness per
    constraint; rows have 3 / 2 / 1 non-zeros in A / B / C with small coefficients, and the
    assignment satisfies every row (so (ab - c)/Z is a polynomial and real proofs verify).
"""
import numpy as np

from .cp_groth16 import (CURVE_PARAMS, FrCodec, MultiStageConstraintSynthesizer, _batch_inverse,
                         SynthesisError)

# BASELINE.json configs -> (n_constraints, n_free witnesses incl. stage 0, n0)   [estimates, SURVEY §8]
CONFIGS = {
    "big-merkle-4x1": dict(n_c=61_000, n_free=4_000, n0=16),          # config 0: m = 2^16
    "big-merkle-64x32": dict(n_c=1_250_000, n_free=50_000, n0=16),    # config 1: m = 2^21, n_v ~ 1.3e6
    "big-merkle-512x64": dict(n_c=2_500_000, n_free=100_000, n0=16),  # config 2: m = 2^22, n_v ~ 2.6e6
    "vm-1024x1024": dict(n_c=1_000_000, n_free=700_000, n0=217_280),  # config 4: m = 2^20, n_v ~ 1.7e6
    "tiny": dict(n_c=1_000, n_free=100, n0=16),
}


class SyntheticSubcircuit(MultiStageConstraintSynthesizer):
    N_INST = 4
    BIT_COEFFS = [1, 1, 1, 2]

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
        bit_cols = ni + np.nonzero(self.is_bit_var)[0]
        if len(bit_cols) == 0:
            bit_cols = np.array([ni + n0], dtype=np.int64)
        self.row_is_bit = rng.random(n_c) < bit_fraction
        any_cols = rng.integers(0, ni + n_free, size=(n_c, 5))
        pick = bit_cols[rng.integers(0, len(bit_cols), size=(n_c, 5))]
        self.cols = np.where(self.row_is_bit[:, None], pick, any_cols).astype(np.int64)   # A: 0..2, B: 3..4
        self.coef_idx = np.where(self.row_is_bit[:, None], rng.integers(0, 4, size=(n_c, 5)),
                                 rng.integers(0, 8, size=(n_c, 5))).astype(np.int64)
        self.seed = None
        self._z = None

    # ---- MultiStageConstraintSynthesizer ---------------------------------------------------------
    def total_num_stages(self):
        return 2

    def set_witness_seed(self, seed):
        """Selects which subcircuit instance (which assignment) the next commit/prove is for."""
        self.seed = seed
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
    def _coef_int(self, i, k):
        return self.BIT_COEFFS[self.coef_idx[i, k]] if self.row_is_bit[i] else self.wide_coeffs[self.coef_idx[i, k]]

    def csr(self, fc):
        n_c, ni = self.n_c, self.N_INST
        bit_tab = fc.enc(self.BIT_COEFFS).reshape(4, fc.nb)
        wide_tab = fc.enc(self.wide_coeffs).reshape(8, fc.nb)
        vals = np.where(self.row_is_bit[:, None, None], bit_tab[self.coef_idx % 4], wide_tab[self.coef_idx])
        A = (np.arange(n_c + 1, dtype=np.uint64) * 3, self.cols[:, :3].astype(np.uint32).ravel(),
             np.ascontiguousarray(vals[:, :3]).ravel())
        B = (np.arange(n_c + 1, dtype=np.uint64) * 2, self.cols[:, 3:].astype(np.uint32).ravel(),
             np.ascontiguousarray(vals[:, 3:]).ravel())
        one = fc.enc1(1)
        C = (np.arange(n_c + 1, dtype=np.uint64), (ni + self.n_free + np.arange(n_c)).astype(np.uint32),
             np.tile(one, n_c))
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
        isbit = self.row_is_bit.tolist()
        bc, wc = self.BIT_COEFFS, self.wide_coeffs
        base = ni + self.n_free
        for i in range(n_c):
            ui = u[i]
            tab = bc if isbit[i] else wc
            ci, co = cidx[i], cols[i]
            a[co[0]] += ui * tab[ci[0]]
            a[co[1]] += ui * tab[ci[1]]
            a[co[2]] += ui * tab[ci[2]]
            b[co[3]] += ui * tab[ci[3]]
            b[co[4]] += ui * tab[ci[4]]
            c[base + i] = ui
        a = [x % r for x in a]
        b = [x % r for x in b]
        return a, b, c, zt, m

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
        base = np.zeros(ni + n_free, dtype=np.int64)
        base[ni:][self.is_bit_var] = rng.integers(0, 2, size=int(self.is_bit_var.sum()))
        z = base.tolist()
        z[0] = 1
        for j in range(1, ni):
            z[j] = prng.randrange(r)
        for j in np.nonzero(~self.is_bit_var)[0].tolist():
            z[ni + j] = prng.randrange(r)
        # bit rows, vectorised: small non-negative integers
        bt = np.array(self.BIT_COEFFS, dtype=np.int64)
        rows = np.nonzero(self.row_is_bit)[0]
        v = base[self.cols[rows]] * bt[self.coef_idx[rows]]
        prod = (v[:, 0] + v[:, 1] + v[:, 2]) * (v[:, 3] + v[:, 4])
        w = [0] * n_c
        for i, x in zip(rows.tolist(), prod.tolist()):
            w[i] = x
        wc = self.wide_coeffs
        cols, cidx = self.cols, self.coef_idx
        for i in np.nonzero(~self.row_is_bit)[0].tolist():
            co, ci = cols[i].tolist(), cidx[i].tolist()
            az = z[co[0]] * wc[ci[0]] + z[co[1]] * wc[ci[1]] + z[co[2]] * wc[ci[2]]
            bz = z[co[3]] * wc[ci[3]] + z[co[4]] * wc[ci[4]]
            w[i] = az * bz % r
        self._z = z + w
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


def make_config(curve, name):
    c = CONFIGS[name]
    return SyntheticSubcircuit(curve, c["n_c"], c["n_free"], c["n0"])
