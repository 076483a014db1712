"""GPU: TIPP prove / verify (hekaton_system_amd/tipa.py over hk_pairing_products / hk_points_lincomb / hk_scalar_pairing /
hk_msm_bases) closing the aggregation of REAL proofs: the instance `agg_front` produces for an 8-subcircuit big-merkle
job (5 key classes) is proved and accepted; every tampering is rejected; the folded commitment keys equal the
closed-form `ipa_polynomial` images of the SRS (checked against the oracle's group arithmetic)."""
import random

import numpy as np
import pytest

from hekaton_system_amd import aggregation as agg, tipa
from hekaton_system_amd.chacha import ChaCha12Rng
from hekaton_system_amd.cp_groth16 import FrCodec, Proof, SeededRng, generate_parameters
from hekaton_system_amd.workload import config_classes, make_config, representative_subcircuit
from oracle.pyref import curve
from oracle.pyref.codec import Codec
from oracle.pyref.params import CURVES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_tipp_closes_the_aggregation_of_real_proofs(cname, ctx_bn254, ctx_bls):
    ctx, cp = (ctx_bn254 if cname == "bn254" else ctx_bls), CURVES[cname]
    fc = FrCodec(cname)
    cd = Codec(cp)
    family, n, reps = config_classes("tiny")
    rnd = random.Random(77)
    keys = {}
    for rep in reps:
        circ = make_config(cname, "tiny", rep)
        pk, _td = generate_parameters(circ, cname, SeededRng(bytes([rep + 1]) * 32), ctx)
        keys[rep] = (circ, pk, pk.upload(ctx))
    proofs, coms, vks, pub = [], [], [], None
    for idx in range(n):
        circ, pk, dpk = keys[representative_subcircuit(family, n, idx)]
        circ.set_witness_seed(9)
        z = circ.assignment_ints()
        pub = pub or z[1:4]
        kappa = ChaCha12Rng(bytes([idx + 40]) * 32).fr(cp.r)
        com = dpk.commit(0, circ.stage0_witness_bytes(), fc.enc1(kappa))
        a, b, c = dpk.prove(circ.full_assignment_bytes(), fc.enc1(rnd.randrange(cp.r)), fc.enc1(rnd.randrange(cp.r)),
                            fc.enc([kappa]), n_v=circ.n_v)
        proofs.append(Proof(a, b, c, [com])); coms.append(com); vks.append(pk.vk)
    alpha, beta = rnd.randrange(2, cp.r), rnd.randrange(2, cp.r)
    srs = tipa.setup(ctx, cname, n, alpha, beta)
    apk = agg.AggProvingKey(ctx, cname, srs.ck, vks)
    super_com = apk.com.commit_only_left(srs.ck, np.concatenate(coms))
    twist, s, t = (rnd.randrange(2, cp.r) for _ in range(3))
    inst = apk.agg_front(super_com, proofs, pub, twist, s, t)
    T = tipa.Tipp(ctx, cname)
    proof = T.prove(srs, inst["left"], inst["right"], twist, inst["commitment"], inst["output"])
    vk = tipa.verifier_key(ctx, cname, srs)
    assert T.verify(vk, inst["commitment"], inst["output"], twist, proof)
    assert len(proof["rounds"]) == 3
    # rounds are taken two per pass through the pairing pipeline (tipa._round_pair: quarter-by-quarter inner products,
    # round k + 1's messages by bilinearity); round by round the prover sends the same proof, member for member
    import os
    os.environ["HK_TIPP_SINGLE_ROUNDS"] = "1"
    try:
        single = T.prove(srs, inst["left"], inst["right"], twist, inst["commitment"], inst["output"])
    finally:
        del os.environ["HK_TIPP_SINGLE_ROUNDS"]
    assert single["rounds"] == proof["rounds"]
    for key in ("final_a", "final_b", "final_v", "final_w", "open_v", "open_w"):
        assert np.array_equal(np.asarray(single[key]), np.asarray(proof[key])), key

    # ---- the folded keys are the closed-form images of the SRS: v' = f_v(alpha) h, w' = f_w(alpha) g
    F = T.F
    tr = tipa.Transcript(cp.r)
    tr.absorb(b"instance", F.encode(inst["commitment"].t), F.encode(inst["commitment"].u), F.encode(inst["output"]),
              twist.to_bytes(32, "little"), n.to_bytes(8, "little"))
    chal = []
    for rd in proof["rounds"]:
        tr.absorb(b"round", *(F.encode(rd[k]) for k in ("TL", "UL", "ZL", "TR", "UR", "ZR")))
        chal.append(tr.challenge(b"c"))
    ch_rev = chal[::-1]
    chi_rev = [pow(c, -1, cp.r) for c in ch_rev]
    G1, G2 = curve.G1(cp), curve.G2(cp)
    fv = tipa.ipa_polynomial_eval(chi_rev, 1, alpha, cp.r)
    fw = pow(alpha, n, cp.r) * tipa.ipa_polynomial_eval(ch_rev, pow(twist, -1, cp.r), alpha, cp.r) % cp.r
    assert cd.g2_from(bytes(proof["final_v"][0])) == G2.mul(cp.g2_gen, fv)
    assert cd.g1_from(bytes(proof["final_w"][0])) == G1.mul(cp.g1_gen, fw)
    coeffs = tipa.ipa_polynomial_coeffs(chi_rev, 1, cp.r)
    assert sum(c * pow(alpha, i, cp.r) for i, c in enumerate(coeffs)) % cp.r == fv

    # ---- soundness smoke tests: anything changed is rejected
    assert not T.verify(vk, inst["commitment"], inst["output"], (twist + 1) % cp.r, proof)
    assert not T.verify(vk, inst["commitment"], F.mul(inst["output"], inst["output"]), twist, proof)
    bad_com = agg.IppCom(F, F.mul(inst["commitment"].t, inst["commitment"].t), inst["commitment"].u)
    assert not T.verify(vk, bad_com, inst["output"], twist, proof)
    tampered = dict(proof)
    tampered["final_a"] = ctx.points_lincomb(1, [proof["final_a"]], fc.enc([2]), n=1)
    assert not T.verify(vk, inst["commitment"], inst["output"], twist, tampered)
    tampered = dict(proof)
    tampered["open_v"] = (proof["open_v"][1], proof["open_v"][0])
    assert not T.verify(vk, inst["commitment"], inst["output"], twist, tampered)
    tampered = dict(proof)
    tampered["rounds"] = [dict(rd) for rd in proof["rounds"]]
    tampered["rounds"][1]["ZL"] = F.mul(tampered["rounds"][1]["ZL"], tampered["rounds"][1]["ZR"])
    assert not T.verify(vk, inst["commitment"], inst["output"], twist, tampered)
    # a proof for a different witness (one aggregated proof's C swapped) does not verify against this instance
    wrong = T.prove(srs, inst["right"] if False else inst["left"], ctx.points_lincomb(2, [inst["right"]], fc.enc([3]), n=n), twist,
                    inst["commitment"], inst["output"])
    assert not T.verify(vk, inst["commitment"], inst["output"], twist, wrong)
    for _c, _pk, dpk in keys.values():
        dpk.free()
    for rb in srs.resident.values():
        rb.free()


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_agg_subcircuit_proofs_end_to_end_with_the_merlin_transcript(cname, ctx_bn254, ctx_bls):
    """aggregation.rs:138-345 whole, challenges from a merlin transcript with the label the reference's e2e test uses
    (coordinator.rs:411): prove, self-verify, and the verifier - re-deriving twist, s, t from the SAME public values -
    accepts; a transcript with another label derives other challenges and the instance no longer matches."""
    from hekaton_system_amd.merlin import Transcript as Merlin
    ctx, cp = (ctx_bn254 if cname == "bn254" else ctx_bls), CURVES[cname]
    fc = FrCodec(cname)
    family, n, reps = config_classes("tiny")
    rnd = random.Random(5)
    keys = {}
    for rep in reps:
        circ = make_config(cname, "tiny", rep)
        pk, _td = generate_parameters(circ, cname, SeededRng(bytes([rep + 9]) * 32), ctx)
        keys[rep] = (circ, pk, pk.upload(ctx))
    proofs, coms, vks, pub = [], [], [], None
    for idx in range(n):
        circ, pk, dpk = keys[representative_subcircuit(family, n, idx)]
        circ.set_witness_seed(3)
        z = circ.assignment_ints()
        pub = pub or z[1:4]
        kappa = ChaCha12Rng(bytes([idx + 90]) * 32).fr(cp.r)
        com = dpk.commit(0, circ.stage0_witness_bytes(), fc.enc1(kappa))
        a, b, c = dpk.prove(circ.full_assignment_bytes(), fc.enc1(rnd.randrange(cp.r)), fc.enc1(rnd.randrange(cp.r)),
                            fc.enc([kappa]), n_v=circ.n_v)
        proofs.append(Proof(a, b, c, [com])); coms.append(com); vks.append(pk.vk)
    srs = tipa.setup(ctx, cname, n, rnd.randrange(2, cp.r), rnd.randrange(2, cp.r))
    apk = agg.AggProvingKey(ctx, cname, srs.ck, vks)
    super_com = apk.com.commit_only_left(srs.ck, np.concatenate(coms))
    proof, inst = apk.agg_subcircuit_proofs(Merlin(b"test-e2e"), super_com, proofs, pub, srs)
    # determinism: the same transcript label gives the same challenges and the same instance
    inst2 = apk.agg_front(super_com, proofs, pub, pt=Merlin(b"test-e2e"))
    assert inst2["twist"] == inst["twist"] and inst2["output"] == inst["output"] and inst2["commitment"] == inst["commitment"]
    # and they are the values a reader of the transcript derives from the public messages alone
    pt = Merlin(b"test-e2e")
    pt.append_serializable(b"AB-commitment", inst["com_ab"].serialize_uncompressed())
    pt.append_serializable(b"C-commitment", inst["com_c"].serialize_uncompressed())
    pt.append_serializable(b"D-commitment", super_com.serialize_uncompressed())
    assert pt.challenge_scalar(b"r-random-fiatshamir", cp.r) == inst["twist"]
    T = tipa.Tipp(ctx, cname)
    vk = tipa.verifier_key(ctx, cname, srs)
    assert T.verify(vk, inst["commitment"], inst["output"], inst["twist"], proof)
    other = apk.agg_front(super_com, proofs, pub, pt=Merlin(b"another label"))
    assert other["twist"] != inst["twist"]
    assert not T.verify(vk, other["commitment"], other["output"], other["twist"], proof)
    for _c, _pk, dpk in keys.values():
        dpk.free()
    for rb in srs.resident.values():
        rb.free()
