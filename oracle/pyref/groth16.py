"""CP-Groth16 (commit-and-prove Groth16) — big-int restatement of the reference's
cp-groth16 crate plus the ark-groth16 LibsnarkReduction it calls.

TEST INFRASTRUCTURE (oracle) — see params.py header.  PARITY UNPINNED by
constants (the reference has no golden vectors for this path); pinned by the
Groth16 pairing equation (verifier.rs:23-43, see pairing_bn254.py) and by a
pairing-free trapdoor check on SRSs whose toxic waste we keep.

Follows, function by function:
  * R1CS.full_assignment / stage witness   cp-groth16/src/constraint_synthesizer.rs:96-106
  * witness_map                             ark-groth16 0.4 r1cs_to_qap.rs `witness_map_from_matrices`
                                            (SURVEY.md Appendix A.1), called at cp-groth16/src/prover.rs:123
  * instance_map_with_evaluation            ark-groth16 0.4 (called at cp-groth16/src/generator.rs:75-76)
  * generate_parameters                     cp-groth16/src/generator.rs:18-238
  * commit                                  cp-groth16/src/committer.rs:55-98
  * prove_last_stage / calculate_coeff      cp-groth16/src/prover.rs:53-171
  * prove (kappa correction)                cp-groth16/src/committer.rs:100-123
  * verify (pairing form)                   cp-groth16/src/verifier.rs:23-71
Randomness (r, s, kappa, toxic waste) is always injected by the caller: the
reference draws it from an RNG (prover.rs:28-29, committer.rs:85), we never do.
"""

from dataclasses import dataclass, field

from . import curve
from .poly import Domain


# --------------------------------------------------------------------------- R1CS

class R1CS:
    """Finalized multi-stage R1CS with its assignment.

    Column indices follow ark-relations `to_matrices`: instance variable i -> i,
    witness variable j -> num_instance + j.  z[0] is the constant 1.
    `stage_ranges[k] = (start, end)` are witness-index ranges
    (constraint_synthesizer.rs:21 `variable_range_for_stage`).
    """

    def __init__(self, r):
        self.r = r
        self.instance = [1]
        self.witness = []
        self.A, self.B, self.C = [], [], []
        self.stage_ranges = []

    # builder -----------------------------------------------------------------
    def begin_stage(self):          # constraint_synthesizer.rs:55-58 initialize_stage
        self.stage_ranges.append((len(self.witness), len(self.witness)))

    def end_stage(self):            # constraint_synthesizer.rs:62-66 finalize_stage
        s, _ = self.stage_ranges[-1]
        self.stage_ranges[-1] = (s, len(self.witness))

    def alloc_instance(self, v):
        self.instance.append(v % self.r)
        return ("i", len(self.instance) - 1)

    def alloc_witness(self, v):
        self.witness.append(v % self.r)
        return ("w", len(self.witness) - 1)

    def enforce(self, a, b, c):
        """a, b, c: lists of (coeff, var) with var = ("i", k) | ("w", k) | "one"."""
        self.A.append(list(a)); self.B.append(list(b)); self.C.append(list(c))

    # finalized views -----------------------------------------------------------
    @property
    def num_instance(self): return len(self.instance)
    @property
    def num_witness(self): return len(self.witness)
    @property
    def num_constraints(self): return len(self.A)

    def _col(self, var):
        if var == "one":
            return 0
        kind, k = var
        return k if kind == "i" else self.num_instance + k

    def matrices(self):
        """[(coeff, col)] rows, as ark `ConstraintMatrices`."""
        conv = lambda M: [[(c % self.r, self._col(v)) for c, v in row] for row in M]
        return conv(self.A), conv(self.B), conv(self.C)

    def full_assignment(self):      # constraint_synthesizer.rs:102-106
        return list(self.instance) + list(self.witness)

    def stage_witness(self, stage): # constraint_synthesizer.rs:96-99 (for the current = given stage)
        s, e = self.stage_ranges[stage]
        return self.witness[s:e]

    def is_satisfied(self):
        z = self.full_assignment()
        A, B, C = self.matrices()
        ev = lambda row: sum(c * z[j] for c, j in row) % self.r
        return all(ev(a) * ev(b) % self.r == ev(c) for a, b, c in zip(A, B, C))


def evaluate_constraint(row, z, r):
    return sum(c * z[j] for c, j in row) % r


def witness_map_from_matrices(cp, A, B, C, num_inputs, num_constraints, z):
    """LibsnarkReduction::witness_map_from_matrices — SURVEY.md Appendix A.1.
    Returns h of length m (the prover uses h[..m-1], prover.rs:128-129)."""
    r = cp.r
    dom = Domain(cp, num_constraints + num_inputs)
    m = dom.size
    a = [0] * m
    b = [0] * m
    for i in range(num_constraints):
        a[i] = evaluate_constraint(A[i], z, r)
        b[i] = evaluate_constraint(B[i], z, r)
    for j in range(num_inputs):
        a[num_constraints + j] = z[j]
    g = cp.fr_generator
    a = dom.coset_fft(dom.ifft(a), g)
    b = dom.coset_fft(dom.ifft(b), g)
    ab = [x * y % r for x, y in zip(a, b)]
    c = [0] * m
    for i in range(num_constraints):
        c[i] = evaluate_constraint(C[i], z, r)
    c = dom.coset_fft(dom.ifft(c), g)
    zinv = pow(dom.evaluate_vanishing_polynomial(g), -1, r)
    ab = [(x - y) * zinv % r for x, y in zip(ab, c)]
    return dom.coset_ifft(ab, g)


def instance_map_with_evaluation(cp, cs, t):
    """LibsnarkReduction::instance_map_with_evaluation (ark-groth16 0.4), as used
    by generator.rs:75-76.  Returns (a, b, c, zt, qap_num_variables, m_raw)."""
    r = cp.r
    A, B, C = cs.matrices()
    n_c, n_in = cs.num_constraints, cs.num_instance
    dom = Domain(cp, n_c + n_in)
    zt = dom.evaluate_vanishing_polynomial(t)
    u = dom.evaluate_all_lagrange_coefficients(t)
    qap_num_variables = (n_in - 1) + cs.num_witness
    a = [0] * (qap_num_variables + 1)
    b = [0] * (qap_num_variables + 1)
    c = [0] * (qap_num_variables + 1)
    for j in range(n_in):
        a[j] = u[n_c + j]
    for i in range(n_c):
        for coeff, idx in A[i]:
            a[idx] = (a[idx] + u[i] * coeff) % r
        for coeff, idx in B[i]:
            b[idx] = (b[idx] + u[i] * coeff) % r
        for coeff, idx in C[i]:
            c[idx] = (c[idx] + u[i] * coeff) % r
    return a, b, c, zt, qap_num_variables, dom.size


# --------------------------------------------------------------------------- keys

@dataclass
class VerifyingKey:             # data_structures.rs:33-46
    alpha_g: tuple
    beta_h: tuple
    gamma_h: tuple
    last_delta_h: tuple
    gamma_abc_g: list
    deltas_h: list


@dataclass
class CommitterKey:             # data_structures.rs:108-114
    last_delta_g: tuple
    deltas_abc_g: list


@dataclass
class ProvingKey:               # data_structures.rs:66-83
    vk: VerifyingKey
    beta_g: tuple
    a_g: list
    b_g: list
    b_h: list
    h_g: list
    ck: CommitterKey
    deltas_g: list

    def last_delta_g(self): return self.deltas_g[-1]        # data_structures.rs:91-93
    def last_delta_h(self): return self.vk.deltas_h[-1]     # data_structures.rs:95-97
    def last_ck(self): return self.ck.deltas_abc_g[-1]      # data_structures.rs:99-101


@dataclass
class Proof:                    # data_structures.rs:7-16
    a: tuple
    b: tuple
    c: tuple
    ds: list = field(default_factory=list)


@dataclass
class Trapdoor:
    """Toxic waste + generator logs kept by the TEST setup only."""
    alpha: int
    beta: int
    gamma: int
    deltas: list
    t: int
    g1: tuple
    g2: tuple
    a: list
    b: list
    c: list
    zt: int
    m: int


def generate_parameters(cp, cs, alpha, beta, gamma, deltas, t, g1_scalar=1, g2_scalar=1):
    """generator.rs:18-238 with every RNG draw replaced by an argument.
    `g1_scalar`/`g2_scalar` pick the (random in the reference, generator.rs:35-36)
    group generators as multiples of the standard ones."""
    r = cp.r
    G1, G2 = curve.G1(cp), curve.G2(cp)
    g = G1.mul(G1.gen, g1_scalar)
    h = G2.mul(G2.gen, g2_scalar)
    assert len(deltas) == len(cs.stage_ranges)
    n_in = cs.num_instance
    a, b, c, zt, qap_num_variables, m_raw = instance_map_with_evaluation(cp, cs, t)
    inv = lambda x: pow(x, -1, r)

    deltas_abc = []
    for delta, (s, e) in zip(deltas, cs.stage_ranges):          # generator.rs:93-106
        di = inv(delta)
        deltas_abc.append([(beta * a[i] + alpha * b[i] + c[i]) * di % r
                           for i in range(s + n_in, e + n_in)])
    gi = inv(gamma)
    gamma_abc = [(beta * a[i] + alpha * b[i] + c[i]) * gi % r for i in range(n_in)]   # :112-117
    last_delta_inv = inv(deltas[-1])
    hq = [zt * last_delta_inv % r * pow(t, i, r) % r for i in range(m_raw - 1)]        # :182

    mul1 = lambda k: G1.mul(g, k % r)
    mul2 = lambda k: G2.mul(h, k % r)
    vk = VerifyingKey(
        alpha_g=mul1(alpha), beta_h=mul2(beta), gamma_h=mul2(gamma),
        last_delta_h=mul2(deltas[-1]),
        gamma_abc_g=[mul1(k) for k in gamma_abc],
        deltas_h=[mul2(d) for d in deltas])
    deltas_g = [mul1(d) for d in deltas]
    pk = ProvingKey(
        vk=vk, beta_g=mul1(beta),
        a_g=[mul1(k) for k in a], b_g=[mul1(k) for k in b], b_h=[mul2(k) for k in b],
        h_g=[mul1(k) for k in hq],
        ck=CommitterKey(last_delta_g=deltas_g[-1],
                        deltas_abc_g=[[mul1(k) for k in v] for v in deltas_abc]),
        deltas_g=deltas_g)
    td = Trapdoor(alpha, beta, gamma, list(deltas), t, g, h, a, b, c, zt, m_raw)
    return pk, td


# --------------------------------------------------------------------------- prover

def commit(cp, cs, pk, stage, kappa):
    """CommitmentBuilder::commit — committer.rs:55-98 (randomness = kappa)."""
    G1 = curve.G1(cp)
    w = cs.stage_witness(stage)
    ck = pk.ck.deltas_abc_g[stage]
    assert len(w) == len(ck)                                   # committer.rs:83
    com = G1.add(G1.msm(ck, w), G1.mul(pk.ck.last_delta_g, kappa))
    return com


def calculate_coeff(G, initial, query, vk_param, assignment):
    """prover.rs:158-171."""
    acc = G.msm(query[1:], assignment)
    return G.sum([initial, query[0], acc, vk_param])


def prove_last_stage(cp, cs, pk, r_, s_):
    """CPGroth16::prove_last_stage — prover.rs:53-156.  Returns (A, B, C) affine."""
    G1, G2 = curve.G1(cp), curve.G2(cp)
    mod = cp.r
    z = cs.full_assignment()
    assignment = z[1:]                                         # prover.rs:78-82
    a_g = calculate_coeff(G1, G1.mul(pk.last_delta_g(), r_), pk.a_g, pk.vk.alpha_g, assignment)
    if r_ % mod == 0:                                          # prover.rs:92-93
        b_g = None
    else:
        b_g = calculate_coeff(G1, G1.mul(pk.last_delta_g(), s_), pk.b_g, pk.beta_g, assignment)
    b_h = calculate_coeff(G2, G2.mul(pk.last_delta_h(), s_), pk.b_h, pk.vk.beta_h, assignment)
    w_last = cs.stage_witness(len(cs.stage_ranges) - 1)
    ck_last = pk.last_ck()
    # E::G1::msm(..).unwrap_or(zero): length mismatch -> zero (prover.rs:116-117)
    l_aux = G1.msm(ck_last, w_last) if len(ck_last) == len(w_last) else None
    A, B, C = cs.matrices()
    h = witness_map_from_matrices(cp, A, B, C, cs.num_instance, cs.num_constraints, z)
    assert len(h) == len(pk.h_g) + 1                           # prover.rs:128
    h_acc = G1.msm(pk.h_g, h[:len(pk.h_g)])
    r_s_delta_g = G1.mul(pk.last_delta_g(), r_ * s_ % mod)
    c_g = G1.sum([G1.mul(a_g, s_), G1.mul(b_g, r_), G1.neg(r_s_delta_g), l_aux, h_acc])
    return a_g, b_h, c_g


def prove(cp, cs, pk, comms, comm_rands, r_, s_):
    """CommitmentBuilder::prove — committer.rs:100-123."""
    G1 = curve.G1(cp)
    a, b, c = prove_last_stage(cp, cs, pk, r_, s_)
    assert len(pk.deltas_g) == len(comm_rands) + 1             # committer.rs:112
    kappas_etas = G1.msm(pk.deltas_g, comm_rands)              # msm_unchecked: zip-truncates
    c = G1.sub(c, kappas_etas)
    return Proof(a=a, b=b, c=c, ds=list(comms))


# --------------------------------------------------------------------------- checks

def verify_proof_trapdoor(cp, cs, pk, td, proof, comm_rands, r_, s_):
    """Pairing-free exact check on an SRS whose toxic waste is known: recompute the
    discrete logs of A, B, C, D_i in Fr and compare group elements.  Also checks the
    Groth16 equation in the exponent (verifier.rs:23-43)."""
    mod = cp.r
    G1, G2 = curve.G1(cp), curve.G2(cp)
    z = cs.full_assignment()
    n_in = cs.num_instance
    inv = lambda x: pow(x, -1, mod)
    dl = td.deltas[-1]
    A, B, C = cs.matrices()
    h = witness_map_from_matrices(cp, A, B, C, n_in, cs.num_constraints, z)
    if h[-1] != 0:      # (ab - c)/Z is a polynomial of degree <= m-2 iff the R1CS is satisfied
        return False
    az = sum(zi * ai for zi, ai in zip(z, td.a)) % mod
    bz = sum(zi * bi for zi, bi in zip(z, td.b)) % mod
    log_a = (r_ * dl + az + td.alpha) % mod
    log_b = (s_ * dl + bz + td.beta) % mod
    abc = lambda i: (td.beta * td.a[i] + td.alpha * td.b[i] + td.c[i]) % mod
    s_last, e_last = cs.stage_ranges[-1]
    l_log = sum(z[n_in + j] * abc(n_in + j) for j in range(s_last, e_last)) % mod * inv(dl) % mod
    h_log = sum(h[i] * pow(td.t, i, mod) for i in range(td.m - 1)) % mod * td.zt % mod * inv(dl) % mod
    log_c = (s_ * log_a + r_ * log_b - r_ * s_ % mod * dl + l_log + h_log) % mod
    d_logs = []
    for k, kappa in enumerate(comm_rands):
        s0, e0 = cs.stage_ranges[k]
        d = sum(z[n_in + j] * abc(n_in + j) for j in range(s0, e0)) % mod * inv(td.deltas[k]) % mod
        d = (d + kappa * dl) % mod
        d_logs.append(d)
        log_c = (log_c - kappa * td.deltas[k]) % mod
    ok = proof.a == G1.mul(td.g1, log_a)
    ok &= proof.b == G2.mul(td.g2, log_b)
    ok &= proof.c == G1.mul(td.g1, log_c)
    ok &= len(proof.ds) == len(d_logs) and all(D == G1.mul(td.g1, d) for D, d in zip(proof.ds, d_logs))
    # Groth16 equation in the exponent
    ic = sum(z[i] * abc(i) for i in range(n_in)) % mod * inv(td.gamma) % mod
    lhs = log_a * log_b % mod
    rhs = (td.alpha * td.beta + ic * td.gamma + sum(d * dk for d, dk in zip(d_logs, td.deltas))
           + log_c * dl) % mod
    ok &= lhs == rhs
    return bool(ok)


def prepare_inputs(cp, vk, public_inputs):
    """verifier.rs:49-62."""
    G1 = curve.G1(cp)
    if len(public_inputs) + 1 != len(vk.gamma_abc_g):
        raise ValueError("MalformedVerifyingKey")
    acc = vk.gamma_abc_g[0]
    for x, base in zip(public_inputs, vk.gamma_abc_g[1:]):
        acc = G1.add(acc, G1.mul(base, x))
    return acc


# --------------------------------------------------------------------------- toy circuits

def poly_eval_circuit(cp, polynomial, point, two_stage=True):
    """The shape of the reference's own unit-test circuit `PolyEvalCircuit`
    (cp-groth16/src/lib.rs:30-100 two-stage, :182-245 single-stage): stage 0
    witnesses the coefficients of a monic polynomial and enforces the leading one
    is 1; the last stage inputs (point, evaluation) and proves the evaluation."""
    r = cp.r
    cs = R1CS(r)
    ev = sum(c * pow(point, i, r) for i, c in enumerate(polynomial)) % r
    cs.begin_stage()
    coeffs = [cs.alloc_witness(c) for c in polynomial]
    cs.enforce([(1, coeffs[-1])], [(1, "one")], [(1, "one")])
    if two_stage:
        cs.enforce([(1, coeffs[-1])], [(1, "one")], [(1, "one")])   # lib.rs:77-84 enforces twice
        cs.end_stage()
        cs.begin_stage()
    pt = cs.alloc_instance(point)
    evv = cs.alloc_instance(ev)
    # claimed_eval = sum coeff_i * point^i ; cur_pow chain
    cur_pow_val = 1
    cur_pow = "one"
    terms = []
    for i, cvar in enumerate(coeffs):
        prod = cs.alloc_witness(polynomial[i] * cur_pow_val % r)
        cs.enforce([(1, cvar)], [(1, cur_pow)], [(1, prod)])
        terms.append((1, prod))
        if i + 1 < len(coeffs):
            nxt_val = cur_pow_val * point % r
            nxt = cs.alloc_witness(nxt_val)
            cs.enforce([(1, cur_pow)], [(1, pt)], [(1, nxt)])
            cur_pow, cur_pow_val = nxt, nxt_val
    cs.enforce(terms, [(1, "one")], [(1, evv)])
    cs.end_stage()
    assert cs.is_satisfied()
    return cs, [point % r, ev]
