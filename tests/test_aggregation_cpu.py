"""CPU: the host arithmetic of the aggregation mirror - GT (gt.py) against the tower oracle, the polynomial helpers of
TIPP (tipa.py) against their definitions, the SHA-256 transcript's determinism."""
import random

import pytest

from hekaton_system_amd import tipa
from hekaton_system_amd.gt import GtField
from oracle.pyref import pairing
from oracle.pyref.params import CURVES


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_gt_arithmetic_matches_the_tower_oracle(cname):
    cp = CURVES[cname]
    T = pairing.tower(cname)
    F = GtField(cname)
    rnd = random.Random(8)
    a = tuple(rnd.randrange(cp.q) for _ in range(12))
    b = tuple(rnd.randrange(cp.q) for _ in range(12))
    A, B = T.f12_from_flat(list(a)), T.f12_from_flat(list(b))
    assert list(F.mul(a, b)) == T.f12_flat(T.f12_mul(A, B))
    e = rnd.randrange(1 << 200)
    assert list(F.pow(a, e)) == T.f12_flat(T.f12_pow(A, e))
    assert list(F.conj(a)) == T.f12_flat(T.f12_conj(A))
    assert F.decode(F.encode(a)) == a and F.mul(a, F.one) == a
    g = tuple(T.f12_flat(T.pairing(cp.g1_gen, cp.g2_gen)))           # in GT the conjugate is the inverse
    assert F.mul(g, F.conj(g)) == F.one


def test_ipa_polynomial_and_division():
    r = CURVES["bn254"].r
    rnd = random.Random(3)
    ch = [rnd.randrange(1, r) for _ in range(4)]
    shift = rnd.randrange(1, r)
    coeffs = tipa.ipa_polynomial_coeffs(ch, shift, r)
    assert len(coeffs) == 16
    for z in (0, 1, rnd.randrange(r)):
        direct = 1
        for k, c in enumerate(ch):
            direct = direct * (1 + c * pow(shift * z, 1 << k, r)) % r
        assert sum(c * pow(z, i, r) for i, c in enumerate(coeffs)) % r == direct == tipa.ipa_polynomial_eval(ch, shift, z, r)
    # (X - z) q(X) + f(z) = f(X)
    z = rnd.randrange(r)
    q = tipa._divide_by_linear(coeffs, z, r)
    assert len(q) == len(coeffs) and q[-1] == 0
    fz = sum(c * pow(z, i, r) for i, c in enumerate(coeffs)) % r
    x = rnd.randrange(r)
    fx = sum(c * pow(x, i, r) for i, c in enumerate(coeffs)) % r
    qx = sum(c * pow(x, i, r) for i, c in enumerate(q)) % r
    assert ((x - z) * qx + fz) % r == fx
    # the Montgomery-scaled form the prover uses: R f(X), whose quotient by (X - z) is R q(X) - no product per coefficient
    # when the quotient's Montgomery bytes are written out
    from hekaton_system_amd.cp_groth16 import FrCodec
    fc = FrCodec("bn254")
    scaled = tipa.ipa_polynomial_coeffs(ch, shift, r, fc.R)
    assert scaled == [c * fc.R % r for c in coeffs]
    assert bytes(fc.enc_canon(tipa._divide_by_linear(scaled, z, r))) == bytes(fc.enc(q))


def test_transcript_is_deterministic_and_order_sensitive():
    r = CURVES["bn254"].r
    t1, t2, t3 = tipa.Transcript(r), tipa.Transcript(r), tipa.Transcript(r)
    t1.absorb(b"a", b"xy", b"z"); t2.absorb(b"a", b"xy", b"z"); t3.absorb(b"a", b"x", b"yz2")
    c1, c2, c3 = t1.challenge(b"c"), t2.challenge(b"c"), t3.challenge(b"c")
    assert c1 == c2 != c3 and 0 < c1 < r
    assert t1.challenge(b"c") != c1


def test_rom_challenges_are_the_sha256_of_the_serialized_commitment():
    """rom_transcript.rs:42-75: SHA-256(context || uncompressed commitment), little-endian, reduced mod r."""
    import hashlib
    from hekaton_system_amd.aggregation import IppCom, rom_challenges
    from hekaton_system_amd.cp_groth16 import CURVE_PARAMS
    from hekaton_system_amd.gt import GtField
    F = GtField("bn254")
    r = CURVE_PARAMS["bn254"]["r"]
    t = tuple(range(1, 13))
    u = tuple(range(101, 113))
    com = IppCom(F, t, u)
    ser = com.serialize_uncompressed()
    assert len(ser) == 2 * 12 * 32 and ser[:32] == (1).to_bytes(32, "little") and ser[384:416] == (101).to_bytes(32, "little")
    e, tr = rom_challenges(com, r)
    assert e == int.from_bytes(hashlib.sha256(b"entry_chal" + ser).digest(), "little") % r
    assert tr == int.from_bytes(hashlib.sha256(b"tr_chal" + ser).digest(), "little") % r
    assert e != tr and rom_challenges(IppCom(F, u, t), r) != (e, tr)


def test_ippcom_lincomb_equals_the_operator_form():
    from hekaton_system_amd.aggregation import IppCom
    from hekaton_system_amd.gt import GtField
    import random
    F = GtField("bn254")
    rnd = random.Random(3)
    el = lambda: tuple(rnd.randrange(F.q) for _ in range(12))
    a, b, c = IppCom(F, el(), el(), el()), IppCom(F, el(), el()), IppCom(F, el(), el())
    k1, k2 = 0x1234567, 0x89abcdef0123
    want = a + b * k1 + c * k2
    got = IppCom.lincomb([(a, None), (b, k1), (c, k2)])
    assert got == want and got.t == want.t and got.u == want.u and got.ip == want.ip
