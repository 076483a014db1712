"""CPU: the big-int oracle pins itself — curve constants, pairing bilinearity, and every Groth16
fixture satisfies the reference's verifier equation (cp-groth16/src/verifier.rs:23-43), which is
the only acceptance criterion the reference's own tests have (lib.rs:179,312)."""
import random

import pytest

from oracle.pyref import curve, groth16, pairing_bn254 as pr, params
from oracle.pyref.codec import Codec
from oracle.pyref.params import BN254, CURVES
from tests import golden_util as gu


def test_params_self_check():
    assert params.self_check()


def test_pairing_bilinear_nondegenerate():
    G1, G2 = curve.G1(BN254), curve.G2(BN254)
    e1 = pr.pairing(G2.gen, G1.gen)
    assert e1 != pr.f12_one()
    assert pr.f12_pow(e1, BN254.r) == pr.f12_one()
    a, b = 0x1234567, 0x7654321
    assert pr.pairing(G2.mul(G2.gen, b), G1.mul(G1.gen, a)) == pr.f12_pow(e1, a * b)
    assert pr.pairing(G2.gen, None) == pr.f12_one()


def _proof_from_case(cd, case):
    pk = case["pk"]
    vk = groth16.VerifyingKey(
        alpha_g=cd.g1_from(gu.hb(pk["alpha_g"])), beta_h=cd.g2_from(gu.hb(pk["beta_h"])),
        gamma_h=cd.g2_from(gu.hb(pk["gamma_h"])), last_delta_h=cd.g2_from(gu.hb(pk["last_delta_h"])),
        gamma_abc_g=[cd.g1_from(gu.hb(pk["gamma_abc_g"])[i * cd.g1_bytes:(i + 1) * cd.g1_bytes])
                     for i in range(case["n_inst"])],
        deltas_h=[cd.g2_from(gu.hb(pk["deltas_h"])[i * cd.g2_bytes:(i + 1) * cd.g2_bytes])
                  for i in range(len(case["stage_ranges"]))])
    proof = groth16.Proof(cd.g1_from(gu.hb(case["proof"]["a"])), cd.g2_from(gu.hb(case["proof"]["b"])),
                          cd.g1_from(gu.hb(case["proof"]["c"])), [cd.g1_from(gu.hb(c)) for c in case["comms"]])
    return vk, proof


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_groth16_fixtures_satisfy_verifier_equation(idx):
    case = gu.load("groth16.json")["bn254"][idx]
    cd = Codec(BN254)
    vk, proof = _proof_from_case(cd, case)
    assert pr.verify_proof(vk, proof, case["public_inputs"])
    G1 = curve.G1(BN254)
    bad = groth16.Proof(proof.a, proof.b, G1.add(proof.c, G1.gen), proof.ds)
    assert not pr.verify_proof(vk, bad, case["public_inputs"])
    with pytest.raises(ValueError):                      # verifier.rs:53-55 MalformedVerifyingKey
        pr.verify_proof(vk, proof, case["public_inputs"][:-1])


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_trapdoor_verifier_rejects_tampering(cname):
    cp = CURVES[cname]
    rnd = random.Random(9)
    r = cp.r
    poly = [rnd.randrange(r) for _ in range(4)] + [1]
    cs, _ = groth16.poly_eval_circuit(cp, poly, rnd.randrange(r), two_stage=True)
    pk, td = groth16.generate_parameters(cp, cs, 3, 5, 7, [11, 13], 17, 2, 3)
    com = groth16.commit(cp, cs, pk, 0, 99)
    proof = groth16.prove(cp, cs, pk, [com], [99], 21, 22)
    assert groth16.verify_proof_trapdoor(cp, cs, pk, td, proof, [99], 21, 22)
    G1 = curve.G1(cp)
    bad = groth16.Proof(G1.add(proof.a, G1.gen), proof.b, proof.c, proof.ds)
    assert not groth16.verify_proof_trapdoor(cp, cs, pk, td, bad, [99], 21, 22)


def test_witness_map_quotient_is_polynomial():
    """h[m-1] == 0 iff the R1CS is satisfied (degree bound of (ab - c)/Z)."""
    cp = BN254
    rnd = random.Random(3)
    cs, _ = groth16.poly_eval_circuit(cp, [5, 7, 1], 9, two_stage=False)
    A, B, C = cs.matrices()
    z = cs.full_assignment()
    assert groth16.witness_map_from_matrices(cp, A, B, C, cs.num_instance, cs.num_constraints, z)[-1] == 0
    z[-1] = (z[-1] + 1) % cp.r
    assert groth16.witness_map_from_matrices(cp, A, B, C, cs.num_instance, cs.num_constraints, z)[-1] != 0
