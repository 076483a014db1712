"""Host-side mirror of the reference's `cp-groth16` crate surface for the hot path, over the C ABI.

Same names, argument meaning and error behaviour as the reference (paths relative to its root):
    MultiStageConstraintSystem / MultiStageConstraintSynthesizer   cp-groth16/src/constraint_synthesizer.rs
    ProvingKey / VerifyingKey / CommitterKey / Proof               cp-groth16/src/data_structures.rs
    generate_parameters                                            cp-groth16/src/generator.rs:18-238
    CommitmentBuilder.{new,commit,prove}                           cp-groth16/src/committer.rs:38-123
    CPGroth16.prove_last_stage{,_with_zk,_without_zk}              cp-groth16/src/prover.rs:19-156

All group/field arithmetic of the hot path runs in libhekaton (HIP); this module only moves bytes and
does the scalar-field bookkeeping of the trusted setup (Lagrange coefficients, QAP evaluations) that
the reference's generator does on the host before its fixed-base MSMs.  Field elements are Python
ints (canonical) on this side and Montgomery little-endian bytes at the C ABI.  No oracle import.
"""
from dataclasses import dataclass, field
import hashlib

import numpy as np

from . import capi

# public curve parameters (same constants gen_params.py bakes into the kernels)
CURVE_PARAMS = {
    "bn254": dict(
        r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
        q=21888242871839275222246405745257275088696311157297823662689037894645226208583,
        fr_bytes=32, fq_bytes=32, two_adicity=28, gen=5,
        g1=(1, 2),
        g2=((10857046999023057135944570762232829481370756359578518086990519993285655852781,
             11559732032986387107991004021392285783925812861821192530917403151452391805634),
            (8495653923123431417604973247489272438418190587263600148770280649306958101930,
             4082367875863433681332203403145435568316851327593401208105741076214120093531))),
    "bls12_381": dict(
        r=52435875175126190479447740508185965837690552500527637822603658699938581184513,
        q=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
        fr_bytes=32, fq_bytes=48, two_adicity=32, gen=7,
        g1=(0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
            0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
        g2=((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
             0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
            (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
             0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))),
}


class SynthesisError(Exception):
    """ark_relations::r1cs::SynthesisError (PolynomialDegreeTooLarge, UnexpectedIdentity, ...)."""


# --------------------------------------------------------------------------------------- field codec
class FrCodec:
    """ints <-> Montgomery bytes of the scalar field, vectorised through Python's int.to_bytes."""

    def __init__(self, curve):
        p = CURVE_PARAMS[curve]
        self.r = p["r"]
        self.q = p["q"]
        self.nb = p["fr_bytes"]
        self.qb = p["fq_bytes"]
        self.R = 1 << (8 * self.nb)
        self.Rq = 1 << (8 * self.qb)
        self.Rinv = pow(self.R, -1, self.r)

    def enc(self, xs):
        r, R, nb = self.r, self.R, self.nb
        return np.frombuffer(b"".join((x % r * R % r).to_bytes(nb, "little") for x in xs), dtype=np.uint8).copy()

    def enc1(self, x):
        return self.enc([x])

    def enc_canon(self, xs):
        nb = self.nb
        return np.frombuffer(b"".join(x.to_bytes(nb, "little") for x in xs), dtype=np.uint8).copy()

    def dec(self, buf):
        b = bytes(buf)
        nb, r, Ri = self.nb, self.r, self.Rinv
        return [int.from_bytes(b[i:i + nb], "little") * Ri % r for i in range(0, len(b), nb)]

    def g1(self, P):
        return np.frombuffer(b"".join((c % self.q * self.Rq % self.q).to_bytes(self.qb, "little") for c in P),
                             dtype=np.uint8).copy()

    def g2(self, P):
        (x0, x1), (y0, y1) = P
        return self.g1((x0, x1, y0, y1))


# --------------------------------------------------------------------------------------- RNG
class SeededRng:
    """Deterministic test RNG for the CALLER-side draws (r, s, com_seed, setup toxic waste), where the reference
    takes any `RngCore` (prover.rs:28-29; worker.rs:91): SHA-256 in counter mode.  The one draw that is part of
    the protocol — kappa = Fr::rand(ChaCha12Rng::from_seed(com_seed)), worker.rs:129-137 — is reproduced bit for
    bit by `chacha.ChaCha12Rng`, which has this class's duck type and is what `worker.py` uses."""

    def __init__(self, seed: bytes):
        assert len(seed) == 32
        self.seed = bytes(seed)
        self.ctr = 0

    def bytes(self, n):
        out = b""
        while len(out) < n:
            out += hashlib.sha256(self.seed + self.ctr.to_bytes(8, "little")).digest()
            self.ctr += 1
        return out[:n]

    def fr(self, r):
        return int.from_bytes(self.bytes(48), "little") % r

    def gen_seed(self):
        return self.bytes(32)


# --------------------------------------------------------------------------------------- constraint system
class MultiStageConstraintSystem:
    """constraint_synthesizer.rs:14-117.  Holds instance/witness assignments, the witness range of
    every stage, and (in setup mode, or when asked) the constraint rows."""

    def __init__(self, r, construct_matrices=True):
        self.r = r
        self.instance_assignment = [1]
        self.witness_assignment = []
        self.variable_range_for_stage = []
        self.construct_matrices = construct_matrices
        self.A, self.B, self.C = [], [], []
        self._n_constraints = 0

    # variables are ("i", k) / ("w", k); "one" is instance 0
    def new_input_variable(self, v):
        self.instance_assignment.append(v % self.r)
        return ("i", len(self.instance_assignment) - 1)

    def new_witness_variable(self, v):
        self.witness_assignment.append(v % self.r)
        return ("w", len(self.witness_assignment) - 1)

    def enforce_constraint(self, a, b, c):
        self._n_constraints += 1
        if self.construct_matrices:
            self.A.append(a); self.B.append(b); self.C.append(c)

    def initialize_stage(self):                      # :55-58
        s = len(self.witness_assignment)
        self.variable_range_for_stage.append((s, s))

    def finalize_stage(self):                        # :62-66
        s, _ = self.variable_range_for_stage[-1]
        self.variable_range_for_stage[-1] = (s, len(self.witness_assignment))

    def synthesize_with(self, constraints):          # :69-77
        self.initialize_stage()
        constraints(self)
        self.finalize_stage()

    def num_instance_variables(self): return len(self.instance_assignment)
    def num_witness_variables(self): return len(self.witness_assignment)
    def num_constraints(self): return self._n_constraints

    def current_stage_witness_assignment(self):      # :96-99
        s, e = self.variable_range_for_stage[-1]
        return self.witness_assignment[s:e]

    def full_assignment(self):                       # :102-106
        return self.instance_assignment + self.witness_assignment

    def finalize(self):                              # LC inlining happens at synthesis time here
        pass

    def _col(self, v):
        if v == "one":
            return 0
        return v[1] if v[0] == "i" else len(self.instance_assignment) + v[1]

    def to_matrices(self):
        """ark `ConstraintMatrices` rows [(coeff, col)]; col: instance i -> i, witness j -> n_inst + j."""
        conv = lambda M: [[(c % self.r, self._col(v)) for c, v in row] for row in M]
        return conv(self.A), conv(self.B), conv(self.C)

    def is_satisfied(self):
        z = self.full_assignment()
        A, B, C = self.to_matrices()
        ev = lambda row: sum(c * z[j] for c, j in row) % self.r
        return all(ev(a) * ev(b) % self.r == ev(c) for a, b, c in zip(A, B, C))


class MultiStageConstraintSynthesizer:
    """constraint_synthesizer.rs:119-134."""

    def total_num_stages(self):
        raise NotImplementedError

    def last_stage(self):
        return self.total_num_stages() - 1

    def generate_constraints(self, stage, cs):
        raise NotImplementedError


def csr_from_rows(fc, rows):
    row_ptr = np.zeros(len(rows) + 1, dtype=np.uint64)
    cols, vals = [], []
    for i, row in enumerate(rows):
        for c, j in row:
            cols.append(j)
            vals.append(c)
        row_ptr[i + 1] = len(cols)
    return row_ptr, np.array(cols, dtype=np.uint32), fc.enc(vals)


# --------------------------------------------------------------------------------------- keys / proof
@dataclass
class VerifyingKey:            # data_structures.rs:33-46 (packed-affine bytes)
    alpha_g: np.ndarray
    beta_h: np.ndarray
    gamma_h: np.ndarray
    last_delta_h: np.ndarray
    gamma_abc_g: np.ndarray
    deltas_h: np.ndarray


@dataclass
class CommitterKey:            # data_structures.rs:108-114
    last_delta_g: np.ndarray
    deltas_abc_g: list


@dataclass
class ProvingKey:              # data_structures.rs:66-83 (+ the circuit class's matrices, see hk_pk_desc)
    vk: VerifyingKey
    beta_g: np.ndarray
    a_g: object
    b_g: object
    b_h: object
    h_g: object
    ck: CommitterKey
    deltas_g: np.ndarray
    matrices: tuple = None
    n_inst: int = 0
    n_constraints: int = 0
    device: object = None      # capi.DevicePk once uploaded

    def last_ck(self):
        return self.ck.deltas_abc_g[-1]

    def upload(self, ctx):
        """Makes the key resident on the device (hk_pk_upload); idempotent."""
        if self.device is None:
            self.device = ctx.pk_upload(
                a_g=self.a_g, b_g=self.b_g, b_h=self.b_h, h_g=self.h_g, ck_stages=self.ck.deltas_abc_g,
                deltas_g=self.deltas_g, last_delta_h=self.vk.last_delta_h, alpha_g=self.vk.alpha_g,
                beta_g=self.beta_g, beta_h=self.vk.beta_h, matrices=self.matrices, n_inst=self.n_inst,
                n_constraints=self.n_constraints)
        return self.device


@dataclass
class Proof:                   # data_structures.rs:7-16
    a: np.ndarray
    b: np.ndarray
    c: np.ndarray
    ds: list = field(default_factory=list)


@dataclass
class Trapdoor:                # toxic waste kept by TEST setups only (lets tests verify without pairings)
    alpha: int
    beta: int
    gamma: int
    deltas: list
    t: int
    g1_scalar: int
    g2_scalar: int
    a: list
    b: list
    c: list
    zt: int
    m: int
    stage_ranges: list = None          # witness range of every stage (variable_range_for_stage)
    n_inst: int = 0


# --------------------------------------------------------------------------------------- setup
def _batch_inverse(xs, r):
    pref = [1] * (len(xs) + 1)
    for i, x in enumerate(xs):
        pref[i + 1] = pref[i] * x % r
    inv = pow(pref[-1], -1, r)
    out = [0] * len(xs)
    for i in range(len(xs) - 1, -1, -1):
        out[i] = inv * pref[i] % r
        inv = inv * xs[i] % r
    return out


def qap_instance_map_with_evaluation(curve, A, B, C, n_inst, n_wit, n_c, t):
    """LibsnarkReduction::instance_map_with_evaluation (generator.rs:75-76): (a, b, c, zt, m)."""
    p = CURVE_PARAMS[curve]
    r = p["r"]
    m, log_m = 1, 0
    while m < n_c + n_inst:
        m *= 2
        log_m += 1
    if log_m > p["two_adicity"]:
        raise SynthesisError("PolynomialDegreeTooLarge")
    w = pow(pow(p["gen"], (r - 1) >> p["two_adicity"], r), 1 << (p["two_adicity"] - log_m), r)
    zt = (pow(t, m, r) - 1) % r
    # L_i(t) = zt * w^i / (m * (t - w^i))
    wi = [1] * m
    for i in range(1, m):
        wi[i] = wi[i - 1] * w % r
    den = _batch_inverse([m * (t - x) % r for x in wi], r)
    u = [zt * x % r * d % r for x, d in zip(wi, den)]
    n_v = n_inst + n_wit
    a = [0] * n_v
    b = [0] * n_v
    c = [0] * n_v
    for j in range(n_inst):
        a[j] = u[n_c + j]
    for i in range(n_c):
        ui = u[i]
        for coeff, idx in A[i]:
            a[idx] = (a[idx] + ui * coeff) % r
        for coeff, idx in B[i]:
            b[idx] = (b[idx] + ui * coeff) % r
        for coeff, idx in C[i]:
            c[idx] = (c[idx] + ui * coeff) % r
    return a, b, c, zt, m


@dataclass
class HostSetup:
    """Everything `generate_parameters` computes in the scalar field on the host (generator.rs:28-117,182), ready for
    the fixed-base MSMs: canonical little-endian scalar arrays, the class's CSR matrices and the toxic waste.  Pure
    Python / numpy - no device - so several classes can be prepared in parallel worker processes (bench.py)."""
    curve: str
    sc_small1: np.ndarray          # [alpha, beta] + deltas + gamma_abc, times the G1 generator scalar (Montgomery)
    sc_small2: np.ndarray          # [beta, gamma] + deltas, times the G2 generator scalar (Montgomery)
    sc_ck: list                    # per stage: deltas_abc * g1s (Montgomery)
    sc_a_g1: np.ndarray            # canonical
    sc_b_g1: np.ndarray
    sc_b_g2: np.ndarray
    sc_h_g1: np.ndarray
    matrices: tuple
    n_inst: int
    n_constraints: int
    n_stages: int
    td: Trapdoor


def setup_host(circuit, curve, rng):
    """Host half of generator.rs:18-238: synthesise every stage in setup mode, draw the toxic waste, evaluate the QAP
    at t, and lay out every scalar the fixed-base MSMs need."""
    p = CURVE_PARAMS[curve]
    r = p["r"]
    fc = FrCodec(curve)
    alpha, beta, gamma = rng.fr(r) or 1, rng.fr(r) or 1, rng.fr(r) or 1
    deltas = [rng.fr(r) or 1 for _ in range(circuit.total_num_stages())]
    g1s, g2s = rng.fr(r) or 1, rng.fr(r) or 1                      # random generators (generator.rs:35-36)
    fast = hasattr(circuit, "qap_evaluate")        # bulk circuits evaluate their own (static) matrices
    cs = MultiStageConstraintSystem(r, construct_matrices=not fast)
    for stage in range(circuit.total_num_stages()):
        circuit.generate_constraints(stage, cs)
    cs.finalize()
    n_inst, n_wit, n_c = cs.num_instance_variables(), cs.num_witness_variables(), cs.num_constraints()
    t = rng.fr(r)
    if fast:
        a, b, c, zt, m = circuit.qap_evaluate(t)
        matrices = circuit.csr(fc)
    else:
        A, B, C = cs.to_matrices()
        a, b, c, zt, m = qap_instance_map_with_evaluation(curve, A, B, C, n_inst, n_wit, n_c, t)
        matrices = (csr_from_rows(fc, A), csr_from_rows(fc, B), csr_from_rows(fc, C))
    inv = lambda x: pow(x, -1, r)
    deltas_abc = []
    for delta, (s, e) in zip(deltas, cs.variable_range_for_stage):          # generator.rs:93-106
        di = inv(delta)
        deltas_abc.append([(beta * a[i] + alpha * b[i] + c[i]) * di % r for i in range(s + n_inst, e + n_inst)])
    gi = inv(gamma)
    gamma_abc = [(beta * a[i] + alpha * b[i] + c[i]) * gi % r for i in range(n_inst)]   # :112-117
    ldi = inv(deltas[-1])
    hq = [0] * (m - 1)                                                        # :182 h_query_scalars
    cur = zt * ldi % r
    for i in range(m - 1):
        hq[i] = cur
        cur = cur * t % r
    td = Trapdoor(alpha, beta, gamma, deltas, t, g1s, g2s, a, b, c, zt, m,
                  stage_ranges=list(cs.variable_range_for_stage), n_inst=n_inst)
    canon = lambda xs, mult: fc.enc_canon([x * mult % r for x in xs])
    return HostSetup(
        curve=curve,
        sc_small1=fc.enc([x * g1s % r for x in [alpha, beta] + deltas + gamma_abc]),
        sc_small2=fc.enc([x * g2s % r for x in [beta, gamma] + deltas]),
        sc_ck=[fc.enc([x * g1s % r for x in v]) for v in deltas_abc],
        sc_a_g1=canon(a, g1s), sc_b_g1=canon(b, g1s), sc_b_g2=canon(b, g2s), sc_h_g1=canon(hq, g1s),
        matrices=matrices, n_inst=n_inst, n_constraints=n_c, n_stages=len(deltas), td=td)


def setup_device(hs, ctx, keep_on_device=False):
    """Device half of generator.rs:126-224: the fixed-base MSMs (hk_fixed_base_g1/g2) over a HostSetup.
    Every key element is (scalar * generator scalar) * G.  Returns (ProvingKey, Trapdoor)."""
    p = CURVE_PARAMS[hs.curve]
    fc = FrCodec(hs.curve)
    G1, G2 = fc.g1(p["g1"]), fc.g2(p["g2"])

    def fb(group, enc):
        n = len(enc) // fc.nb
        if keep_on_device:
            out = capi.DeviceBuffer(ctx, n * (ctx.g1_bytes if group == 1 else ctx.g2_bytes))
            ctx.fixed_base(group, G1 if group == 1 else G2, enc, n=n, montgomery=False, out=out)
            return out
        return ctx.fixed_base(group, G1 if group == 1 else G2, enc, montgomery=False)

    small1 = ctx.fixed_base(1, G1, hs.sc_small1)
    small2 = ctx.fixed_base(2, G2, hs.sc_small2)
    g1b, g2b = ctx.g1_bytes, ctx.g2_bytes
    k = hs.n_stages
    alpha_g, beta_g = small1[:g1b], small1[g1b:2 * g1b]
    deltas_g = small1[2 * g1b:(2 + k) * g1b]
    gamma_abc_g = small1[(2 + k) * g1b:]
    beta_h, gamma_h = small2[:g2b], small2[g2b:2 * g2b]
    deltas_h = small2[2 * g2b:]
    vk = VerifyingKey(alpha_g=alpha_g, beta_h=beta_h, gamma_h=gamma_h, last_delta_h=deltas_h[-g2b:],
                      gamma_abc_g=gamma_abc_g, deltas_h=deltas_h)
    ck = CommitterKey(last_delta_g=deltas_g[-g1b:],
                      deltas_abc_g=[np.asarray(ctx.fixed_base(1, G1, v)) if len(v) else np.zeros(0, np.uint8)
                                    for v in hs.sc_ck])
    pk = ProvingKey(vk=vk, beta_g=beta_g, a_g=fb(1, hs.sc_a_g1), b_g=fb(1, hs.sc_b_g1), b_h=fb(2, hs.sc_b_g2),
                    h_g=fb(1, hs.sc_h_g1), ck=ck, deltas_g=deltas_g,
                    matrices=hs.matrices, n_inst=hs.n_inst, n_constraints=hs.n_constraints)
    return pk, hs.td


def generate_parameters(circuit, curve, rng, ctx, keep_on_device=False):
    """generator.rs:18-238.  Synthesises every stage in setup mode, evaluates the QAP at a random
    point on the host (setup_host), then runs the fixed-base MSMs on the GPU (setup_device).
    Returns (ProvingKey, Trapdoor)."""
    return setup_device(setup_host(circuit, curve, rng), ctx, keep_on_device)


# --------------------------------------------------------------------------------------- prover
class CPGroth16:
    """prover.rs:15-171.  `cs`/`circuit` as in the reference; `pk` must be uploaded (ProvingKey.upload)."""

    @staticmethod
    def prove_last_stage_with_zk(cs, circuit, pk, rng, comm_rands=()):
        r_mod = cs.r
        r, s = rng.fr(r_mod), rng.fr(r_mod)                         # prover.rs:28-29
        return CPGroth16.prove_last_stage(cs, circuit, pk, r, s, comm_rands)

    @staticmethod
    def prove_last_stage_without_zk(cs, circuit, pk, comm_rands=()):
        return CPGroth16.prove_last_stage(cs, circuit, pk, 0, 0, comm_rands)

    @staticmethod
    def prove_last_stage(cs, circuit, pk, r, s, comm_rands=()):
        """Synthesises the last stage on the host (prover.rs:70-75), then one hk_prove call does
        prover.rs:78-155 and the kappa correction of committer.rs:112-114 on the GPU."""
        circuit.generate_constraints(circuit.last_stage(), cs)
        cs.finalize()
        fc = FrCodec(pk.device.ctx.curve)
        z = getattr(circuit, "full_assignment_bytes", None)
        z = z(cs) if z else fc.enc(cs.full_assignment())
        n_v = cs.num_instance_variables() + cs.num_witness_variables()
        a, b, c = pk.device.prove(z, fc.enc1(r), fc.enc1(s), fc.enc(list(comm_rands)), n_v=n_v)
        return a, b, c


class CommitmentBuilder:
    """committer.rs:17-123."""

    def __init__(self, circuit, pk, r_mod=None):
        if pk.device is None:
            raise RuntimeError("proving key is not resident: call pk.upload(ctx) first (no CPU path)")
        self.curve = pk.device.ctx.curve
        self.cs = MultiStageConstraintSystem(CURVE_PARAMS[self.curve]["r"], construct_matrices=False)
        self.circuit = circuit
        self.cur_stage = 0
        self.pk = pk

    @classmethod
    def new(cls, circuit, pk):
        return cls(circuit, pk)

    def commit(self, rng):
        """committer.rs:55-98: synthesise the current stage, com = msm(ck[stage], w) + kappa*delta_g.
        Returns (commitment bytes, randomness int); the randomness is the FIRST draw from `rng`."""
        self.circuit.generate_constraints(self.cur_stage, self.cs)
        w = self.cs.current_stage_witness_assignment()
        if self.cur_stage >= len(self.pk.ck.deltas_abc_g):
            raise IndexError("no more values left in committing key")       # committer.rs:81
        fc = FrCodec(self.curve)
        randomness = rng.fr(self.cs.r)                                         # committer.rs:85
        com = self.pk.device.commit(self.cur_stage, fc.enc(w), fc.enc1(randomness), n=len(w))
        self.cur_stage += 1
        return com, randomness

    def prove(self, comms, comm_rands, rng):
        """committer.rs:100-123."""
        if len(self.pk.deltas_g) // self.pk.device.ctx.g1_bytes != len(comm_rands) + 1:
            raise AssertionError("deltas_g.len() == comm_rands.len() + 1")      # committer.rs:112
        a, b, c = CPGroth16.prove_last_stage_with_zk(self.cs, self.circuit, self.pk, rng, comm_rands)
        return Proof(a=a, b=b, c=c, ds=list(comms))


# --------------------------------------------------------------------------------------- trapdoor verifier
def trapdoor_verify(ctx, curve, td, n_inst, stage_ranges, z, h, comms, kappas, r_, s_, proof):
    """Pairing-free acceptance test for proofs made under a TEST setup whose toxic waste `td` is known: recomputes
    log A, log B, log C and log D_i in Fr, rebuilds the four group elements with the GPU fixed-base path
    (hk_fixed_base_g1/g2 - a different kernel family from the prover's bucket MSMs) and compares bytes; then checks
    the verifier equation of cp-groth16/src/verifier.rs:23-43 in the exponent:
        log A * log B = alpha*beta + ic*gamma + sum_i D_i*delta_i + log C * delta_last.
    z: full assignment ints; h: quotient coefficients ints (m of them); stage_ranges: [(start, end)] witness ranges
    of every stage (MultiStageConstraintSystem.variable_range_for_stage); comms/kappas: one per earlier stage.
    Returns None, raises AssertionError naming the first mismatch."""
    p = CURVE_PARAMS[curve]
    mod = p["r"]
    fc = FrCodec(curve)
    inv = lambda x: pow(x, -1, mod)
    dl = td.deltas[-1]
    az = sum(x * y for x, y in zip(z, td.a)) % mod
    bz = sum(x * y for x, y in zip(z, td.b)) % mod
    abc = [(td.beta * a + td.alpha * b + c) % mod for a, b, c in zip(td.a, td.b, td.c)]
    log_a = (r_ * dl + az + td.alpha) % mod
    log_b = (s_ * dl + bz + td.beta) % mod
    ls, le = stage_ranges[-1]
    l_log = sum(z[i] * abc[i] for i in range(n_inst + ls, n_inst + le)) % mod * inv(dl) % mod
    hsum, tp = 0, 1
    for i in range(td.m - 1):
        hsum += h[i] * tp
        tp = tp * td.t % mod
    h_log = hsum % mod * td.zt % mod * inv(dl) % mod
    log_c = (s_ * log_a + r_ * log_b - r_ * s_ % mod * dl + l_log + h_log) % mod
    d_logs = []
    for k, kappa in enumerate(kappas):
        s0, e0 = stage_ranges[k]
        d = sum(z[i] * abc[i] for i in range(n_inst + s0, n_inst + e0)) % mod * inv(td.deltas[k]) % mod
        d_logs.append((d + kappa * dl) % mod)
        log_c = (log_c - kappa * td.deltas[k]) % mod
    G1, G2 = fc.g1(p["g1"]), fc.g2(p["g2"])
    g1b = ctx.g1_bytes
    want1 = np.asarray(ctx.fixed_base(1, G1, fc.enc([x * td.g1_scalar % mod for x in [log_a, log_c] + d_logs])))
    want2 = np.asarray(ctx.fixed_base(2, G2, fc.enc([log_b * td.g2_scalar % mod])))
    a, b, c = proof
    assert bytes(np.asarray(a, np.uint8)) == bytes(want1[:g1b]), "proof.a is not [log A] g"
    assert bytes(np.asarray(b, np.uint8)) == bytes(want2), "proof.b is not [log B] h"
    assert bytes(np.asarray(c, np.uint8)) == bytes(want1[g1b:2 * g1b]), "proof.c is not [log C] g"
    for k, com in enumerate(comms):
        assert bytes(np.asarray(com, np.uint8)) == bytes(want1[(2 + k) * g1b:(3 + k) * g1b]), "commitment %d" % k
    ic = sum(z[i] * abc[i] for i in range(n_inst)) % mod * inv(td.gamma) % mod
    lhs = log_a * log_b % mod
    rhs = (td.alpha * td.beta + ic * td.gamma + sum(d * dk for d, dk in zip(d_logs, td.deltas)) + log_c * dl) % mod
    assert lhs == rhs, "Groth16 verifier equation (verifier.rs:23-43) does not hold in the exponent"
