"""CPU: host-side logic of the mirror layer (no GPU): constraint-system bookkeeping, the trusted-setup
scalar computation against the oracle's restatement of ark-groth16, synthetic workloads, codecs, worker records."""
import random

import numpy as np
import pytest

from hekaton_system_amd.cp_groth16 import (CURVE_PARAMS, FrCodec, MultiStageConstraintSystem, SeededRng,
                                           qap_instance_map_with_evaluation, csr_from_rows)
from hekaton_system_amd.workload import SyntheticSubcircuit, make_config, CONFIGS
from hekaton_system_amd.worker import Stage0Response, Stage1Response, shard_range
from hekaton_system_amd.cp_groth16 import Proof
from oracle.pyref import groth16 as og
from oracle.pyref.codec import Codec
from oracle.pyref.params import CURVES


def test_multistage_constraint_system_bookkeeping():
    """constraint_synthesizer.rs:55-106: stage ranges, current-stage witnesses, full assignment order."""
    r = CURVE_PARAMS["bn254"]["r"]
    cs = MultiStageConstraintSystem(r)
    cs.synthesize_with(lambda c: [c.new_witness_variable(v) for v in (10, 11, 12)])
    assert cs.variable_range_for_stage == [(0, 3)]
    assert cs.current_stage_witness_assignment() == [10, 11, 12]

    def stage1(c):
        x = c.new_input_variable(5)
        w = c.new_witness_variable(25)
        c.enforce_constraint([(1, x)], [(1, x)], [(1, w)])
    cs.synthesize_with(stage1)
    assert cs.variable_range_for_stage == [(0, 3), (3, 4)]
    assert cs.current_stage_witness_assignment() == [25]
    assert cs.full_assignment() == [1, 5, 10, 11, 12, 25]            # instance || witness, z[0] = 1
    assert (cs.num_instance_variables(), cs.num_witness_variables(), cs.num_constraints()) == (2, 4, 1)
    A, B, C = cs.to_matrices()
    assert A == [[(1, 1)]] and C == [[(1, 2 + 3)]]                     # witness j -> column n_inst + j
    assert cs.is_satisfied()


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_qap_instance_map_matches_oracle(cname):
    """generator.rs:75-76 `instance_map_with_evaluation`: the host mirror's Lagrange/QAP evaluation against
    the oracle's independent restatement."""
    cp = CURVES[cname]
    rnd = random.Random(3)
    from tests.util import synthetic_r1cs
    cs = synthetic_r1cs(cp, rnd, n_inst=4, n_free=12, n_c=29, two_stage_split=5)
    A, B, C = cs.matrices()
    t = rnd.randrange(2, cp.r)
    a, b, c, zt, m = qap_instance_map_with_evaluation(cname, A, B, C, cs.num_instance, cs.num_witness,
                                                      cs.num_constraints, t)
    oa, ob, oc, ozt, _qnv, om = og.instance_map_with_evaluation(cp, cs, t)
    assert (a, b, c, zt, m) == (oa, ob, oc, ozt, om)


def test_synthetic_subcircuit_is_satisfiable_and_shaped():
    circ = make_config("bn254", "tiny")
    circ.set_witness_seed(7)
    z = circ.assignment_ints()
    assert z[0] == 1 and len(z) == circ.n_v == 4 + CONFIGS["tiny"]["n_free"] + CONFIGS["tiny"]["n_c"]
    fc = FrCodec("bn254")
    (rpa, ca, va), (rpb, cb, vb), (rpc, cc, vc) = circ.csr(fc)
    da, db, dc = fc.dec(va), fc.dec(vb), fc.dec(vc)
    r = circ.r
    dot = lambda rp, col, d, i: sum(d[k] * z[col[k]] for k in range(int(rp[i]), int(rp[i + 1]))) % r
    for i in range(circ.n_c):
        assert dot(rpa, ca, da, i) * dot(rpb, cb, db, i) % r == dot(rpc, cc, dc, i)
        ra, rb, rc = circ.rows(i)                                   # the per-row view agrees with the CSR
        assert [(da[k], int(ca[k])) for k in range(int(rpa[i]), int(rpa[i + 1]))] == ra
        assert [(db[k], int(cb[k])) for k in range(int(rpb[i]), int(rpb[i + 1]))] == rb
        assert [(dc[k], int(cc[k])) for k in range(int(rpc[i]), int(rpc[i + 1]))] == rc
    # SHA-like mixture and dense queries
    small = sum(1 for x in z if x < 2) / len(z)
    assert 0.75 < small < 0.95
    da_, db_ = circ.query_density()
    assert da_ > 0.3 and db_ > 0.3
    # stage split: 4 instance variables, n0 stage-0 witnesses first
    cs = MultiStageConstraintSystem(r, construct_matrices=False)
    circ.generate_constraints(0, cs)
    assert len(cs.current_stage_witness_assignment()) == circ.n0
    circ.generate_constraints(1, cs)
    assert cs.full_assignment() == z and cs.num_constraints() == circ.n_c
    # Montgomery bytes of the assignment round-trip
    assert fc.dec(circ.full_assignment_bytes()) == z
    # a different subcircuit of the same class: same matrices, different assignment
    circ.set_witness_seed(8)
    assert circ.assignment_ints() != z


def test_qap_evaluate_fast_path_matches_generic():
    circ = SyntheticSubcircuit("bn254", n_c=40, n_free=20, n0=6)
    fc = FrCodec("bn254")
    t = 123456789
    a, b, c, zt, m = circ.qap_evaluate(t)
    (rpa, ca, va), (rpb, cb, vb), (rpc, cc, vc) = circ.csr(fc)

    def rows(rp, col, val):
        d = fc.dec(val)
        return [[(d[k], int(col[k])) for k in range(int(rp[i]), int(rp[i + 1]))] for i in range(len(rp) - 1)]
    ga, gb, gc, gzt, gm = qap_instance_map_with_evaluation("bn254", rows(rpa, ca, va), rows(rpb, cb, vb),
                                                           rows(rpc, cc, vc), 4, circ.n_wit, circ.n_c, t)
    assert (a, b, c, zt, m) == (ga, gb, gc, gzt, gm)


def test_codecs_agree_with_oracle_codec():
    for cname in ("bn254", "bls12_381"):
        cp = CURVES[cname]
        fc, cd = FrCodec(cname), Codec(cp)
        xs = [0, 1, cp.r - 1, 0x1234567890ABCDEF]
        assert fc.enc(xs).tobytes() == cd.fr_vec_mont(xs).tobytes()
        assert fc.dec(fc.enc(xs)) == xs
        assert fc.g1(CURVE_PARAMS[cname]["g1"]).tobytes() == cd.g1(cp.g1_gen)
        assert fc.g2(CURVE_PARAMS[cname]["g2"]).tobytes() == cd.g2(cp.g2_gen)


def test_seeded_rng_first_draw_is_the_commitment_randomness():
    """mpi-snark/src/worker.rs:63-66: kappa is re-derived as the first draw of an RNG seeded with com_seed."""
    r = CURVE_PARAMS["bn254"]["r"]
    seed = bytes(range(32))
    assert SeededRng(seed).fr(r) == SeededRng(seed).fr(r)
    assert SeededRng(seed).fr(r) != SeededRng(bytes(32)).fr(r)


def test_response_records_round_trip():
    g1, g2 = 64, 128
    r0 = Stage0Response(17, np.arange(g1, dtype=np.uint8), bytes(range(32)))
    back = Stage0Response.from_record(r0.to_record(), g1)
    assert (back.subcircuit_idx, back.com.tobytes(), back.com_seed) == (17, r0.com.tobytes(), r0.com_seed)
    p = Proof(np.full(g1, 1, np.uint8), np.full(g2, 2, np.uint8), np.full(g1, 3, np.uint8), [np.full(g1, 4, np.uint8)])
    r1 = Stage1Response(9, p)
    rec = r1.to_record()
    assert len(rec) == 8 + g1 + g2 + g1 + g1
    b1 = Stage1Response.from_record(rec, g1, g2)
    assert b1.subcircuit_idx == 9 and b1.proof.ds[0].tobytes() == p.ds[0].tobytes()
    assert list(shard_range(512, 8, 7)) == list(range(448, 512))


def test_class_maps_follow_the_reference_layouts():
    """index -> proving-key class: tree_hash_circuit.rs:192-216, vkd_constraints.rs:199-214 (layout of
    vkd.rs:362-612), vm_constraints.rs:91-97."""
    from hekaton_system_amd.workload import (FAMILIES, config_classes, representative_subcircuit,
                                             unique_subcircuits)
    for name in FAMILIES:
        fam, n, reps = config_classes(name)
        for r in reps:
            assert representative_subcircuit(fam, n, r) == r        # a representative represents itself
        assert {representative_subcircuit(fam, n, i) for i in range(n)} == set(reps)
    assert unique_subcircuits("big-merkle", 64) == [0, 1, 63, 62, 61]
    bm = [representative_subcircuit("big-merkle", 64, i) for i in range(64)]
    assert bm[0] == 0 and bm[1:32] == [1] * 31 and bm[32:62] == [61] * 30 and bm[62:] == [62, 63]
    vk = [representative_subcircuit("vkd", 256, i) for i in range(256)]
    assert vk[:7] == [0] * 6 + [6] and vk[7:15] == [7, 8, 8, 10, 8, 8, 8, 8]
    assert vk[15:23] == [8, 8, 8, 8, 19, 8, 8, 8] and vk[255] == 255 and vk[19] == 19
    assert len(unique_subcircuits("vkd", 256)) == 7
    vm = [representative_subcircuit("vm", 1024, i) for i in range(1024)]
    assert vm[0] == 0 and set(vm[1:]) == {1}
    import pytest
    with pytest.raises(IndexError):
        representative_subcircuit("big-merkle", 64, 64)
    a = make_config("bn254", "tiny", 0)
    b = make_config("bn254", "tiny", 1)
    assert (a.cols != b.cols).any()                                 # classes differ in their matrices
