"""GPU: the aggregation front half (hekaton_system_amd/aggregation.py over hk_pairing_products / hk_scalar_pairing /
hk_points_lincomb) on REAL proofs made by this prover for an 8-subcircuit big-merkle job with its 5 proving-key classes:

  * the reference's own consistency check holds - the pairing-product equation of the twisted proofs
    (distributed-prover/src/aggregation.rs:265-269), i.e. every proof satisfies the Groth16 verifier equation under ITS
    class's key, combined with random-looking twist powers;
  * IPP commitments (aggregation.rs:97-103,167-168; coordinator.rs:339) equal the tower oracle's pairings bit for bit and
    are homomorphic (commit(A) + commit(A') = commit(A + A'), commit(A) * k = commit(k A));
  * hk_points_lincomb equals the oracle's group arithmetic element by element;
  * the TIPA instance's output z_lr is the twisted inner product of the returned witness vectors."""
import random

import numpy as np
import pytest

from hekaton_system_amd import aggregation as agg
from hekaton_system_amd.chacha import ChaCha12Rng
from hekaton_system_amd.cp_groth16 import FrCodec, Proof, SeededRng, generate_parameters
from hekaton_system_amd.workload import config_classes, make_config, representative_subcircuit
from oracle.pyref import curve, pairing
from oracle.pyref.codec import Codec
from oracle.pyref.params import CURVES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_aggregation_front_half_on_real_proofs(cname, ctx_bn254, ctx_bls):
    ctx, cp = (ctx_bn254 if cname == "bn254" else ctx_bls), CURVES[cname]
    fc = FrCodec(cname)
    cd = Codec(cp)
    T = pairing.tower(cname)
    G1, G2 = curve.G1(cp), curve.G2(cp)
    family, n, reps = config_classes("tiny")              # big-merkle, 8 subcircuits, 5 classes
    keys = {}
    for rep in reps:
        circ = make_config(cname, "tiny", rep)
        pk, _td = generate_parameters(circ, cname, SeededRng(bytes([rep]) * 32), ctx)
        keys[rep] = (circ, pk, pk.upload(ctx))
    rnd = random.Random(2026)
    proofs, coms, vks = [], [], []
    pub = None
    for idx in range(n):
        circ, pk, dpk = keys[representative_subcircuit(family, n, idx)]
        circ.set_witness_seed(42)                          # same instance (entry_chal, tr_chal, root) for every subcircuit
        z = circ.assignment_ints()
        pub = pub or z[1:4]
        assert z[1:4] == pub
        kappa = ChaCha12Rng(bytes([idx]) * 32).fr(cp.r)
        com = dpk.commit(0, circ.stage0_witness_bytes(), fc.enc1(kappa))
        a, b, c = dpk.prove(circ.full_assignment_bytes(), fc.enc1(rnd.randrange(cp.r)), fc.enc1(rnd.randrange(cp.r)),
                            fc.enc([kappa]), n_v=circ.n_v)
        proofs.append(Proof(a, b, c, [com]))
        coms.append(com)
        vks.append(pk.vk)
    ck = agg.tipa_commitment_key(ctx, cname, n, a=rnd.randrange(1, cp.r), b=rnd.randrange(1, cp.r))
    apk = agg.AggProvingKey(ctx, cname, ck, vks)
    com_g = np.concatenate(coms)
    super_com = apk.com.commit_only_left(ck, com_g)                                  # coordinator.rs:339
    twist, s, t = (rnd.randrange(1, cp.r) for _ in range(3))
    res = apk.agg_front(super_com, proofs, pub, twist, s, t)                          # asserts aggregation.rs:265-269
    F = apk.F

    # IPP commitments against the oracle's pairings
    def g1s(buf): return [cd.g1_from(bytes(buf[i:i + cd.g1_bytes])) for i in range(0, len(buf), cd.g1_bytes)]
    def g2s(buf): return [cd.g2_from(bytes(buf[i:i + cd.g2_bytes])) for i in range(0, len(buf), cd.g2_bytes)]
    flat = lambda e: tuple(T.f12_flat(e))
    C_, V1, V2 = g1s(com_g), g2s(ck.v1), g2s(ck.v2)
    assert super_com.t == flat(T.multi_pairing(list(zip(C_, V1)))) and super_com.u == flat(T.multi_pairing(list(zip(C_, V2))))
    A_, B_ = g1s(np.concatenate([p.a for p in proofs])), g2s(np.concatenate([p.b for p in proofs]))
    W1 = g1s(ck.w1)
    assert res["com_ab"].t == flat(T.f12_mul(T.multi_pairing(list(zip(A_, V1))), T.multi_pairing(list(zip(W1, B_)))))
    assert res["com_ab"].ip == flat(T.multi_pairing(list(zip(A_, B_))))
    # homomorphism of the commitment
    A2 = ctx.fixed_base(1, fc.g1(cp.g1_gen), fc.enc([rnd.randrange(1, cp.r) for _ in range(n)]))
    a_bytes = np.concatenate([p.a for p in proofs])
    summed = ctx.points_lincomb(1, [a_bytes, A2], fc.enc([1, 1]), n=n)
    assert apk.com.commit_only_left(ck, a_bytes) + apk.com.commit_only_left(ck, A2) == apk.com.commit_only_left(ck, summed)
    k = rnd.randrange(1, cp.r)
    scaled = ctx.points_lincomb(1, [a_bytes], fc.enc([k]), n=n)
    assert apk.com.commit_only_left(ck, a_bytes) * k == apk.com.commit_only_left(ck, scaled)
    # hk_points_lincomb against the oracle's group law, element by element (prepared inputs, :192-203)
    S = [g1s(v) for v in apk.s]
    want = [G1.add(G1.add(S[0][i], G1.mul(S[1][i], pub[0])), G1.add(G1.mul(S[2][i], pub[1]), G1.mul(S[3][i], pub[2]))) for i in range(n)]
    assert g1s(res["prepared_input"]) == want
    # z_lr = prod e(left_i^(twist^i), right_i)
    L, R = g1s(res["left"]), g2s(res["right"])
    tw = [pow(twist, i, cp.r) for i in range(n)]
    assert res["output"] == flat(T.multi_pairing([(G1.mul(L[i], tw[i]), R[i]) for i in range(n)]))
    # a tampered proof must break the reference's equation
    bad = list(proofs)
    bad[3] = Proof(proofs[3].a, proofs[3].b, proofs[4].c, proofs[3].ds)
    with pytest.raises(AssertionError):
        apk.agg_front(super_com, bad, pub, twist, s, t)
    for _c, _pk, dpk in keys.values():
        dpk.free()


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_g2_fold_along_the_endomorphism_equals_the_plain_combination(cname, ctx_bn254, ctx_bls):
    """hk_points_fold_g2 (the challenge split into four ~64-bit parts along psi) gives, element by element and byte for
    byte, what hk_points_lincomb_g2 gives for lo + c * hi with the full 254 / 255-bit scalar - including infinity on either
    side, c = 0, 1, r - 1 and the eigenvalue itself."""
    from hekaton_system_amd.endo import psi4
    ctx = ctx_bn254 if cname == "bn254" else ctx_bls
    fc = FrCodec(cname)
    from hekaton_system_amd.cp_groth16 import CURVE_PARAMS
    p = CURVE_PARAMS[cname]
    rnd = random.Random(31)
    n = 37
    g2b = ctx.g2_bytes
    gen2 = fc.g2(p["g2"])
    lo = ctx.fixed_base(2, gen2, fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)])).copy()
    hi = ctx.fixed_base(2, gen2, fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)])).copy()
    lo[3 * g2b:4 * g2b] = 0                                   # infinity in lo
    hi[5 * g2b:6 * g2b] = 0                                   # infinity in hi
    lo[7 * g2b:8 * g2b] = 0
    hi[7 * g2b:8 * g2b] = 0                                   # both
    lam = psi4(cname).lam
    for c in [0, 1, 2, p["r"] - 1, lam, p["r"] - lam, (1 << 64) + 1] + [rnd.randrange(p["r"]) for _ in range(4)]:
        want = ctx.points_lincomb(2, [lo, hi], fc.enc([1, c]), n=n)
        got = ctx.points_fold_g2(lo, hi, c, n=n)
        assert np.array_equal(got, want), c


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_g1_fold_along_the_glv_endomorphism_equals_the_plain_combination(cname, ctx_bn254, ctx_bls):
    """hk_points_fold_g1 against hk_points_lincomb_g1 for lo + c * hi, byte for byte, infinity included."""
    from hekaton_system_amd.cp_groth16 import CURVE_PARAMS
    from hekaton_system_amd.endo import phi2
    ctx = ctx_bn254 if cname == "bn254" else ctx_bls
    fc = FrCodec(cname)
    p = CURVE_PARAMS[cname]
    rnd = random.Random(32)
    n = 41
    g1b = ctx.g1_bytes
    gen1 = fc.g1(p["g1"])
    lo = ctx.fixed_base(1, gen1, fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)])).copy()
    hi = ctx.fixed_base(1, gen1, fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)])).copy()
    lo[2 * g1b:3 * g1b] = 0
    hi[4 * g1b:5 * g1b] = 0
    lo[6 * g1b:7 * g1b] = 0
    hi[6 * g1b:7 * g1b] = 0
    lam = phi2(cname).lam
    for c in [0, 1, 2, p["r"] - 1, lam, p["r"] - lam, (1 << 127) + 5] + [rnd.randrange(p["r"]) for _ in range(4)]:
        want = ctx.points_lincomb(1, [lo, hi], fc.enc([1, c]), n=n)
        got = ctx.points_fold_g1(lo, hi, c, n=n)
        assert np.array_equal(got, want), c
