"""GPU: the host mirror of the reference's cp-groth16 / worker surface, used the way the reference's
own tests use theirs (cp-groth16/src/lib.rs:140-180: generate_parameters -> CommitmentBuilder::commit ->
prove -> verify_proof), with the SRS built on the GPU (hk_fixed_base_*) and the proof checked by the
oracle's pairing verifier."""
import numpy as np
import pytest

from hekaton_system_amd import capi
from hekaton_system_amd.cp_groth16 import (CURVE_PARAMS, CommitmentBuilder, FrCodec, MultiStageConstraintSynthesizer,
                                           SeededRng, generate_parameters)
from hekaton_system_amd.worker import (Stage0Request, Stage1Request, WorkerState, Stage1Response)
from hekaton_system_amd.workload import make_config
from oracle.pyref import groth16 as og, pairing_bn254
from oracle.pyref.codec import Codec
from oracle.pyref.params import BN254

pytestmark = pytest.mark.gpu


class PolyEvalCircuit(MultiStageConstraintSynthesizer):
    """cp-groth16/src/lib.rs:30-135 — stage 0 witnesses a monic polynomial, stage 1 proves an evaluation."""

    def __init__(self, polynomial, r):
        self.polynomial, self.r = polynomial, r
        self.point = None
        self.evaluation = None
        self.coeff_vars = None

    def add_point(self, point):
        self.point = point
        self.evaluation = sum(c * pow(point, i, self.r) for i, c in enumerate(self.polynomial)) % self.r

    def total_num_stages(self):
        return 2

    def generate_constraints(self, stage, cs):
        if stage == 0:
            def s0(cs):
                self.coeff_vars = [cs.new_witness_variable(c) for c in self.polynomial]
                for _ in range(2):                               # lib.rs:77-84 enforces it twice
                    cs.enforce_constraint([(1, self.coeff_vars[-1])], [(1, "one")], [(1, "one")])
            cs.synthesize_with(s0)
        else:
            def s1(cs):
                point = self.point if self.point is not None else 0
                ev = self.evaluation if self.evaluation is not None else 0
                pt = cs.new_input_variable(point)
                evv = cs.new_input_variable(ev)
                cur, cur_val, terms = "one", 1, []
                for i, cv in enumerate(self.coeff_vars):
                    prod = cs.new_witness_variable(self.polynomial[i] * cur_val)
                    cs.enforce_constraint([(1, cv)], [(1, cur)], [(1, prod)])
                    terms.append((1, prod))
                    if i + 1 < len(self.coeff_vars):
                        cur_val = cur_val * point % self.r
                        nxt = cs.new_witness_variable(cur_val)
                        cs.enforce_constraint([(1, cur)], [(1, pt)], [(1, nxt)])
                        cur = nxt
                cs.enforce_constraint(terms, [(1, "one")], [(1, evv)])
            cs.synthesize_with(s1)


def _oracle_vk(cd, pk, n_inst, n_stages):
    g1, g2 = cd.g1_bytes, cd.g2_bytes
    return og.VerifyingKey(
        alpha_g=cd.g1_from(pk.vk.alpha_g), beta_h=cd.g2_from(pk.vk.beta_h), gamma_h=cd.g2_from(pk.vk.gamma_h),
        last_delta_h=cd.g2_from(pk.vk.last_delta_h),
        gamma_abc_g=[cd.g1_from(pk.vk.gamma_abc_g[i * g1:(i + 1) * g1]) for i in range(n_inst)],
        deltas_h=[cd.g2_from(pk.vk.deltas_h[i * g2:(i + 1) * g2]) for i in range(n_stages)])


def test_poly_commit_like_reference(ctx_bn254):
    r = CURVE_PARAMS["bn254"]["r"]
    rng = SeededRng(b"\x01" * 32)
    polynomial = [rng.fr(r) for _ in range(10)] + [1]
    circuit = PolyEvalCircuit(polynomial, r)
    setup_circuit = PolyEvalCircuit(polynomial, r)
    setup_circuit.add_point(0)
    pk, _td = generate_parameters(setup_circuit, "bn254", rng, ctx_bn254)
    pk.upload(ctx_bn254)
    rng = SeededRng(b"\x02" * 32)
    cb = CommitmentBuilder.new(circuit, pk)
    comm, rand = cb.commit(rng)
    point = rng.fr(r)
    cb.circuit.add_point(point)
    inputs = [point, cb.circuit.evaluation]
    proof = cb.prove([comm], [rand], rng)
    cd = Codec(BN254)
    oproof = og.Proof(cd.g1_from(proof.a), cd.g2_from(proof.b), cd.g1_from(proof.c), [cd.g1_from(d) for d in proof.ds])
    assert pairing_bn254.verify_proof(_oracle_vk(cd, pk, 3, 2), oproof, inputs)        # lib.rs:179
    assert not pairing_bn254.verify_proof(_oracle_vk(cd, pk, 3, 2), oproof, [point, (inputs[1] + 1) % r])
    with pytest.raises(AssertionError):                                                # committer.rs:112
        CommitmentBuilder.new(circuit, pk).prove([], [], rng)
    pk.device.free()


def test_worker_two_round_flow(ctx_bn254):
    """all_in_one.rs:109-196 in miniature: stage 0 for every subcircuit, then stage 1, on synthetic
    big-merkle-shaped subcircuits; every proof must verify under the pairing equation."""
    r = CURVE_PARAMS["bn254"]["r"]
    circ_class = make_config("bn254", "tiny")
    pk, _td = generate_parameters(circ_class, "bn254", SeededRng(b"\x03" * 32), ctx_bn254)
    pk.upload(ctx_bn254)
    cd = Codec(BN254)
    vk = _oracle_vk(cd, pk, 4, 2)
    n_sub = 3
    states, circuits = [], []
    for i in range(n_sub):
        c = make_config("bn254", "tiny")
        c.set_witness_seed(100 + i)
        circuits.append(c)
        states.append(WorkerState(n_sub, lambda idx: pk, lambda idx, c=c: c, r))
    rng = SeededRng(b"\x04" * 32)
    resp0 = [st.stage_0(rng, Stage0Request(i)) for i, st in enumerate(states)]
    assert [x.subcircuit_idx for x in resp0] == list(range(n_sub))
    resp1 = [st.stage_1(rng, Stage1Request(i)) for i, st in enumerate(states)]
    for i, (r0, r1) in enumerate(zip(resp0, resp1)):
        rec = Stage1Response.from_record(r1.to_record(), cd.g1_bytes, cd.g2_bytes)      # wire round trip
        p = rec.proof
        oproof = og.Proof(cd.g1_from(p.a), cd.g2_from(p.b), cd.g1_from(p.c), [cd.g1_from(d) for d in p.ds])
        assert np.array_equal(p.ds[0], r0.com)
        assert pairing_bn254.verify_proof(vk, oproof, circuits[i].assignment_ints()[1:4])
    # the stage-0 pass over all three subcircuits with ONE hk_commit_batch call: the same draws, the same responses
    from hekaton_system_amd.worker import process_stage0_requests_batch
    fresh = []
    for i in range(n_sub):
        c = make_config("bn254", "tiny")
        c.set_witness_seed(100 + i)
        fresh.append(c)
    rng2 = SeededRng(b"\x04" * 32)
    batch = process_stage0_requests_batch([rng2] * n_sub, [pk] * n_sub, [Stage0Request(i) for i in range(n_sub)], fresh)
    for (b0, cb), r0 in zip(batch, resp0):
        assert b0.subcircuit_idx == r0.subcircuit_idx and b0.com_seed == r0.com_seed and np.array_equal(b0.com, r0.com)
        assert cb.cur_stage == 1
    pk.device.free()
