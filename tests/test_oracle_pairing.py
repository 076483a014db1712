"""CPU: the tower pairing oracle (oracle/pyref/pairing.py, both curves) pins itself and the fixtures.

  * bilinear, non-degenerate, GT of order r; infinity members are skipped (ark multi_miller_loop);
  * ark's final-exponentiation addition chains reproduce their documented closed forms:
    value = f^((p^12-1)/r) raised to 2x(6x^2+3x+1) on BN254 (Fuentes-Castaneda) and to 3 on BLS12-381 (eprint 2020/875);
  * BN254: equal to the INDEPENDENT affine / flat-basis pairing of pairing_bn254.py raised to that multiple;
  * every Groth16 fixture of BOTH curves satisfies the reference's acceptance test, the verifier equation of
    cp-groth16/src/verifier.rs:23-43 (the BLS12-381 fixtures - the curve north_star names - had only the trapdoor
    check in round 1)."""
import pytest

from oracle.pyref import curve, groth16, pairing, pairing_bn254 as flat
from oracle.pyref.codec import Codec
from oracle.pyref.params import BN254, CURVES
from tests import golden_util as gu
from tests.test_oracle_py import _proof_from_case


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_bilinear_nondegenerate_and_closed_form(cname):
    cp = CURVES[cname]
    T = pairing.tower(cname)
    G1, G2 = curve.G1(cp), curve.G2(cp)
    P, Q = cp.g1_gen, cp.g2_gen
    e = T.pairing(P, Q)
    assert e != T.f12_one()
    assert T.f12_pow(e, cp.r) == T.f12_one()
    a, b = 0x1234567, 0x7654321
    assert T.pairing(G1.mul(P, a), G2.mul(Q, b)) == T.f12_pow(e, a * b % cp.r)
    assert T.multi_pairing([(G1.mul(P, a), Q), (P, G2.mul(Q, b)), (None, Q), (P, None)]) == T.f12_pow(e, a + b)
    assert T.multi_pairing([]) == T.f12_one()
    plain = T.f12_pow(T.multi_miller_loop([(P, Q)]), (cp.q ** 12 - 1) // cp.r)
    assert T.f12_pow(plain, T.hard_part_multiple % cp.r) == e
    assert T.f12_from_flat(T.f12_flat(e)) == e


def test_bn254_tower_equals_independent_flat_basis_pairing():
    T = pairing.tower("bn254")
    P, Q = BN254.g1_gen, BN254.g2_gen
    want = flat.f12_pow(flat.pairing(Q, P), T.hard_part_multiple % BN254.r)
    got = [0] * 12                        # tower -> Fq[w]/(w^12 - 18 w^6 + 82): (a0 + a1 u) v^j w^i, v = w^2, u = w^6 - 9
    e = T.pairing(P, Q)
    for i in range(2):
        for j in range(3):
            a0, a1 = e[i][j]
            k = i + 2 * j
            got[k] = (got[k] + a0 - 9 * a1) % T.p
            got[k + 6] = (got[k + 6] + a1) % T.p
    assert got == want


@pytest.mark.parametrize("cname,idx", [("bn254", 0), ("bn254", 2), ("bls12_381", 0), ("bls12_381", 1), ("bls12_381", 2)])
def test_fixtures_satisfy_the_verifier_equation_on_both_curves(cname, idx):
    cases = gu.load("groth16.json")[cname]
    if idx >= len(cases):
        pytest.skip("no such fixture")
    case = cases[idx]
    cp = CURVES[cname]
    cd = Codec(cp)
    vk, proof = _proof_from_case(cd, case)
    assert pairing.verify_proof(cname, vk, proof, case["public_inputs"])
    G1 = curve.G1(cp)
    bad = groth16.Proof(proof.a, proof.b, G1.add(proof.c, G1.gen), proof.ds)
    assert not pairing.verify_proof(cname, vk, bad, case["public_inputs"])
