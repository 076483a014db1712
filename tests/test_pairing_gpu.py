"""GPU: hk_multi_pairing / hk_pairing_products against the tower oracle, bit for bit (GT is a unique Fq12 element:
ark's tower ordering c0..c11, Montgomery LE), at N in {0, 1, 2, 64} with infinity members on both curves, N = 1024 on
BN254 through bilinearity (all points are known multiples of the generators, so the product must be e(G,H)^(sum a_i b_i):
one oracle pairing + one oracle exponentiation instead of 1024 Miller loops), and the 4 x 4 cross-term batch of
aggregation.rs:255-263."""
import random

import numpy as np
import pytest

from oracle.pyref import curve, pairing
from oracle.pyref.params import CURVES
from tests.test_pairing_cpu import Enc

pytestmark = pytest.mark.gpu


def _ctx(cname, ctx_bn254, ctx_bls):
    return ctx_bn254 if cname == "bn254" else ctx_bls


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_multi_pairing_small_bit_exact(cname, ctx_bn254, ctx_bls):
    ctx = _ctx(cname, ctx_bn254, ctx_bls)
    cp = CURVES[cname]
    T = pairing.tower(cname)
    E = Enc(cp)
    G1, G2 = curve.G1(cp), curve.G2(cp)
    rnd = random.Random(77)
    assert ctx.gt_bytes == 12 * E.nb
    for n, with_inf in ((0, False), (1, False), (2, False), (64, True)):
        ps = [G1.mul(cp.g1_gen, rnd.randrange(1, cp.r)) for _ in range(n)]
        qs = [G2.mul(cp.g2_gen, rnd.randrange(1, cp.r)) for _ in range(n)]
        if with_inf:
            ps[3] = None
            qs[17] = None
            ps[40], qs[40] = None, None
        g1 = np.frombuffer(b"".join(E.g1(p) for p in ps), np.uint8) if n else np.zeros(0, np.uint8)
        g2 = np.frombuffer(b"".join(E.g2(q) for q in qs), np.uint8) if n else np.zeros(0, np.uint8)
        got = E.f12_dec(ctx.multi_pairing(g1, g2, n=n).tobytes())
        assert got == T.f12_flat(T.multi_pairing(list(zip(ps, qs)))), (cname, n)


def test_multi_pairing_1024_bilinearity_and_cross_terms(ctx_bn254):
    cname = "bn254"
    cp = CURVES[cname]
    T = pairing.tower(cname)
    E = Enc(cp)
    from hekaton_system_amd.cp_groth16 import FrCodec
    fc = FrCodec(cname)
    rnd = random.Random(4)
    n = 1024
    gen1 = np.frombuffer(E.g1(cp.g1_gen), np.uint8)
    gen2 = np.frombuffer(E.g2(cp.g2_gen), np.uint8)
    e_gen = T.pairing(cp.g1_gen, cp.g2_gen)

    def vec1():
        ks = [rnd.randrange(1, cp.r) for _ in range(n)]
        return ks, ctx_bn254.fixed_base(1, gen1, fc.enc(ks))

    def vec2():
        ks = [rnd.randrange(1, cp.r) for _ in range(n)]
        return ks, ctx_bn254.fixed_base(2, gen2, fc.enc(ks))

    a, A = vec1()
    b, B = vec2()
    got = E.f12_dec(ctx_bn254.multi_pairing(A, B).tobytes())
    want = T.f12_pow(e_gen, sum(x * y for x, y in zip(a, b)) % cp.r)
    assert got == T.f12_flat(want)
    # 4 x 4 cross terms in one batched call (aggregation.rs:255-263), every entry checked the same way
    lhs = [(a, A)] + [vec1() for _ in range(3)]
    rhs = [(b, B)] + [vec2() for _ in range(3)]
    out = ctx_bn254.pairing_products([v for _, v in lhs], [v for _, v in rhs])
    assert out.shape == (4, 4, ctx_bn254.gt_bytes)
    for i, (ka, _) in enumerate(lhs):
        for j, (kb, _) in enumerate(rhs):
            want = T.f12_pow(e_gen, sum(x * y for x, y in zip(ka, kb)) % cp.r)
            assert E.f12_dec(out[i, j].tobytes()) == T.f12_flat(want), (i, j)
    assert E.f12_dec(out[0, 0].tobytes()) == got
    # a LIST of pairs out of the 4 x 4 grid (hk_pairing_pairs: the ten cross terms of a GIPA round), repeats allowed
    pairs = [(0, 0), (3, 1), (1, 3), (2, 2), (3, 1), (0, 3)]
    sel = ctx_bn254.pairing_pairs([v for _, v in lhs], [v for _, v in rhs], pairs)
    assert sel.shape == (len(pairs), ctx_bn254.gt_bytes)
    for k, (i, j) in enumerate(pairs):
        assert np.array_equal(sel[k], out[i, j]), (k, i, j)


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_gt_pow_matches_the_oracle(cname, ctx_bn254, ctx_bls):
    """hk_gt_pow (one wavefront per element on the wave multiplier) against the tower oracle: the default form splits the
    exponent along the Frobenius (GT elements only: pi(z) = z^q, four parts <= 67 bits, one joint chain) - exponents 0, 1,
    r - 1, the eigenvalue and its negative, 2^64, random; hk_fq12_pow (the plain 254-step chain) also on a generic Fq12 base."""
    ctx = _ctx(cname, ctx_bn254, ctx_bls)
    cp = CURVES[cname]
    T = pairing.tower(cname)
    E = Enc(cp)
    from hekaton_system_amd.cp_groth16 import FrCodec
    from hekaton_system_amd.endo import psi4
    fc = FrCodec(cname)
    rnd = random.Random(21)
    g = T.pairing(cp.g1_gen, cp.g2_gen)
    h = T.f12_pow(g, 12345)
    lam = psi4(cname).lam
    exps = [0, 1, cp.r - 1, lam, cp.r - lam, lam * lam % cp.r, 1 << 64, (1 << 128) - 1] + [rnd.randrange(cp.r) for _ in range(6)]
    bases = [g if k % 2 == 0 else h for k in range(len(exps))]
    enc = lambda bs: np.frombuffer(b"".join(E.f12(T.f12_flat(b)) for b in bs), np.uint8)
    out = ctx.gt_pow(enc(bases), fc.enc(exps))
    for k, (b, e) in enumerate(zip(bases, exps)):
        assert E.f12_dec(out[k].tobytes()) == T.f12_flat(T.f12_pow(b, e)), k
    bases2 = [g, T.f12_from_flat([rnd.randrange(cp.q) for _ in range(12)]), h]
    exps2 = [cp.r - 1, rnd.randrange(cp.r), rnd.randrange(cp.r)]
    out2 = ctx.gt_pow(enc(bases2), fc.enc(exps2), in_gt=False)          # hk_fq12_pow
    for k, (b, e) in enumerate(zip(bases2, exps2)):
        assert E.f12_dec(out2[k].tobytes()) == T.f12_flat(T.f12_pow(b, e)), ("plain", k)
    assert np.array_equal(out2[0], out[2])
    # hk_gt_pow_prod: grouped multi-exponentiation (the fold check of the TIPA verifier) against the product of the single
    # powers, on both chains; group lengths 1 (= hk_gt_pow), 2, 7 and the whole vector
    def prod(xs):
        acc = xs[0]
        for x in xs[1:]:
            acc = T.f12_mul(acc, x)
        return acc
    single = [T.f12_pow(b, e) for b, e in zip(bases, exps)]
    for glen in (1, 2, 7, len(exps)):
        got = ctx.gt_pow_prod(enc(bases), fc.enc(exps), glen)
        assert got.shape == (len(exps) // glen, ctx.gt_bytes)
        for k in range(len(exps) // glen):
            assert E.f12_dec(got[k].tobytes()) == T.f12_flat(prod(single[k * glen:(k + 1) * glen])), (glen, k)
    got = ctx.gt_pow_prod(enc(bases2), fc.enc(exps2), 3, in_gt=False)
    assert E.f12_dec(got[0].tobytes()) == T.f12_flat(prod([T.f12_pow(b, e) for b, e in zip(bases2, exps2)]))
    with pytest.raises(Exception):
        ctx.gt_pow_prod(enc(bases), fc.enc(exps), 3)              # 14 elements are not a multiple of 3
