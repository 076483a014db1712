"""GPU regression for round 2's psi(0, 0) anomaly: infinity lanes - at every position that matters
inside and across wavefronts - go through psi (k_points_psi4), phi (k_points_phi2) and the negations both kernels apply,
on BLS12-381 and BN254, THROUGH THE LIBRARY (hk_points_fold_g2 / hk_points_fold_g1), and come back as (0, 0) while every
other lane equals the oracle's scalar multiple.  The first form of k_points_psi4 returned a launch-to-launch varying
non-zero y for the (0, 0) lane on BLS12-381: a hipcc miscompile (a deleted copy in the else arm of the inlined
`Fp::neg`), fenced at build time by tools/isa_lanecheck.py; this test is the run-time side of the fence."""
import random

import numpy as np
import pytest

from hekaton_system_amd.cp_groth16 import FrCodec
from hekaton_system_amd.endo import phi2, psi4
from oracle.pyref import curve
from oracle.pyref.codec import Codec
from oracle.pyref.params import CURVES

pytestmark = pytest.mark.gpu

INF_LANES = (0, 1, 31, 63, 64, 65, 69)          # first / last lane of a wavefront, the next wavefront, the tail


@pytest.mark.parametrize("cname", ["bls12_381", "bn254"])
def test_infinity_lanes_through_psi_and_its_negations(cname, ctx_bn254, ctx_bls):
    ctx = ctx_bn254 if cname == "bn254" else ctx_bls
    cp, fc, cd = CURVES[cname], FrCodec(cname), Codec(CURVES[cname])
    G2 = curve.G2(cp)
    rnd = random.Random(77)
    n = 70
    g2b = ctx.g2_bytes
    ks = [rnd.randrange(1, cp.r) for _ in range(n)]
    hi = ctx.fixed_base(2, cd.g2_vec([cp.g2_gen]), fc.enc(ks)).copy()
    for i in INF_LANES:
        hi[i * g2b:(i + 1) * g2b] = 0
    lo = np.zeros(n * g2b, dtype=np.uint8)                      # lo = O everywhere: the fold returns c * hi
    pts = [None if i in INF_LANES else G2.mul(cp.g2_gen, ks[i]) for i in range(n)]
    lam = psi4(cname).lam
    # c = lam^j exercises psi^j alone; r - lam^j its negation (neg_mask bit j); a random c all four images at once
    scalars = [1, lam, lam * lam % cp.r, pow(lam, 3, cp.r), cp.r - lam, cp.r - pow(lam, 3, cp.r), rnd.randrange(cp.r)]
    first = {}
    for rep in range(3):                                        # the wrong limb changed from launch to launch
        for c in scalars:
            got = ctx.points_fold_g2(lo, hi, c, n=n)
            for i in range(n):
                have = cd.g2_from(got[i * g2b:(i + 1) * g2b])
                if i in INF_LANES:
                    assert have is None, (cname, rep, c, i, "psi(O) must be O")
                elif rep == 0 and (i < 3 or i in (62, 66)):     # oracle multiples are slow: a few lanes around the O lanes
                    want = G2.mul(pts[i], c)
                    assert have == (want[0], want[1]), (cname, c, i)
            if rep:
                assert np.array_equal(got, first[c]), (cname, rep, c, "not deterministic")
            else:
                first[c] = got.copy()


@pytest.mark.parametrize("cname", ["bls12_381", "bn254"])
def test_infinity_lanes_through_phi_and_its_negations(cname, ctx_bn254, ctx_bls):
    ctx = ctx_bn254 if cname == "bn254" else ctx_bls
    cp, fc, cd = CURVES[cname], FrCodec(cname), Codec(CURVES[cname])
    G1 = curve.G1(cp)
    rnd = random.Random(78)
    n = 70
    g1b = ctx.g1_bytes
    ks = [rnd.randrange(1, cp.r) for _ in range(n)]
    hi = ctx.fixed_base(1, cd.g1_vec([cp.g1_gen]), fc.enc(ks)).copy()
    for i in INF_LANES:
        hi[i * g1b:(i + 1) * g1b] = 0
    lo = np.zeros(n * g1b, dtype=np.uint8)
    lam = phi2(cname).lam
    first = {}
    scalars = [1, lam, cp.r - 1, cp.r - lam, rnd.randrange(cp.r)]
    for rep in range(3):
        for c in scalars:
            got = ctx.points_fold_g1(lo, hi, c, n=n)
            for i in range(n):
                have = cd.g1_from(got[i * g1b:(i + 1) * g1b])
                if i in INF_LANES:
                    assert have is None, (cname, rep, c, i, "phi(O) must be O")
                elif rep == 0 and (i < 3 or i in (62, 66)):
                    want = G1.mul(G1.mul(cp.g1_gen, ks[i]), c)
                    assert have == (want[0], want[1]), (cname, c, i)
            if rep:
                assert np.array_equal(got, first[c]), (cname, rep, c, "not deterministic")
            else:
                first[c] = got.copy()


@pytest.mark.parametrize("cname", ["bls12_381", "bn254"])
@pytest.mark.parametrize("group", [1, 2])
def test_scalar_pairing_over_the_endomorphism_equals_the_oracle(cname, group, ctx_bn254, ctx_bls):
    """hk_scalar_pairing (k_scalar_mul_endo: the scalar split on the device along phi / psi) against the oracle's scalar
    multiplication, element by element: edge scalars (0, 1, r - 1, the eigenvalue, 2^128 - 1), infinity points, powers of
    one value as `structured_scalar_power` makes them (aggregation.rs:224), random scalars."""
    ctx = ctx_bn254 if cname == "bn254" else ctx_bls
    cp, fc, cd = CURVES[cname], FrCodec(cname), Codec(CURVES[cname])
    G = curve.G1(cp) if group == 1 else curve.G2(cp)
    gen = cp.g1_gen if group == 1 else cp.g2_gen
    pb = ctx.g1_bytes if group == 1 else ctx.g2_bytes
    dec = cd.g1_from if group == 1 else cd.g2_from
    rnd = random.Random(5 + group)
    lam = (phi2(cname) if group == 1 else psi4(cname)).lam
    tw = rnd.randrange(2, cp.r)
    scalars = [0, 1, cp.r - 1, lam, cp.r - lam, (1 << 128) - 1, 2] + [pow(tw, i, cp.r) for i in range(20)] + \
              [rnd.randrange(cp.r) for _ in range(43)]
    n = len(scalars)
    ks = [rnd.randrange(1, cp.r) for _ in range(n)]
    pts_b = ctx.fixed_base(group, (cd.g1_vec if group == 1 else cd.g2_vec)([gen]), fc.enc(ks)).copy()
    pts_b[9 * pb:10 * pb] = 0                                    # an infinity point with a non-trivial scalar
    got = ctx.scalar_pairing(group, pts_b, fc.enc(scalars), n=n)
    for i in range(n):
        have = dec(got[i * pb:(i + 1) * pb])
        if i == 9 or scalars[i] == 0:
            assert have is None, (cname, group, i)
            continue
        want = G.mul(G.mul(gen, ks[i]), scalars[i])
        assert have == (want[0], want[1]), (cname, group, i, scalars[i])


@pytest.mark.parametrize("cname", ["bls12_381", "bn254"])
@pytest.mark.parametrize("group", [1, 2])
def test_k_lanes_per_element_equals_one_lane_per_element(cname, group, ctx_bn254, ctx_bls, monkeypatch):
    """Short vectors run k_points_mul_split (K lanes per element, signed 4-bit windows, Jacobian chain), long ones the
    one-lane kernels (k_scalar_mul_endo / k_points_fold_endo): the same bytes from both, for `scalar_pairing` and for the
    fold lo + c * hi with a non-trivial lo, on a vector that is neither a multiple of the 64 / K elements of a workgroup
    nor free of infinity points; small magnitudes (leading zero digits of every lane) included."""
    ctx = ctx_bn254 if cname == "bn254" else ctx_bls
    cp, fc, cd = CURVES[cname], FrCodec(cname), Codec(CURVES[cname])
    gen = cp.g1_gen if group == 1 else cp.g2_gen
    pb = ctx.g1_bytes if group == 1 else ctx.g2_bytes
    rnd = random.Random(11 + group)
    n = 1501
    vec = (cd.g1_vec if group == 1 else cd.g2_vec)([gen])
    hi = ctx.fixed_base(group, vec, fc.enc([rnd.randrange(1, cp.r) for _ in range(n)])).copy()
    lo = ctx.fixed_base(group, vec, fc.enc([rnd.randrange(1, cp.r) for _ in range(n)])).copy()
    for i in (0, 5, 63, 64, 1500):
        hi[i * pb:(i + 1) * pb] = 0
    lo[5 * pb:6 * pb] = 0
    lo[7 * pb:8 * pb] = 0
    scalars = [rnd.randrange(cp.r) for _ in range(n)]
    scalars[:6] = [0, 1, cp.r - 1, 7, 8, 1 << 64]
    fold = ctx.points_fold_g1 if group == 1 else ctx.points_fold_g2
    coeffs = [rnd.randrange(cp.r), 1, 8, cp.r - 8, 0]
    split = [ctx.scalar_pairing(group, hi, fc.enc(scalars), n=n).copy()] + [fold(lo, hi, c, n=n).copy() for c in coeffs]
    monkeypatch.setenv("HK_ENDO_ONE_LANE", "1")
    one = [ctx.scalar_pairing(group, hi, fc.enc(scalars), n=n).copy()] + [fold(lo, hi, c, n=n).copy() for c in coeffs]
    for k, (a, b) in enumerate(zip(split, one)):
        assert np.array_equal(a, b), (cname, group, k)
    # lo + 0 * hi = lo, lo + 1 * hi where hi = O is lo
    assert np.array_equal(split[5], lo)
    assert np.array_equal(split[2][5 * pb:6 * pb], np.zeros(pb, np.uint8))


@pytest.mark.parametrize("cname", ["bls12_381", "bn254"])
@pytest.mark.parametrize("group", [1, 2])
def test_folds_that_share_a_scalar_in_one_call(cname, group, ctx_bn254, ctx_bls):
    """hk_points_fold_many_g1 / _g2 (three vector pairs, one scalar, one launch) = three hk_points_fold calls, into host
    arrays and into device buffers."""
    from hekaton_system_amd.capi import DeviceBuffer
    ctx = ctx_bn254 if cname == "bn254" else ctx_bls
    cp, fc, cd = CURVES[cname], FrCodec(cname), Codec(CURVES[cname])
    gen = cp.g1_gen if group == 1 else cp.g2_gen
    pb = ctx.g1_bytes if group == 1 else ctx.g2_bytes
    rnd = random.Random(21 + group)
    n = 37
    vec = (cd.g1_vec if group == 1 else cd.g2_vec)([gen])
    mk = lambda: ctx.fixed_base(group, vec, fc.enc([rnd.randrange(1, cp.r) for _ in range(n)])).copy()
    los, his = [mk() for _ in range(3)], [mk() for _ in range(3)]
    his[1][:pb] = 0
    los[2][pb:2 * pb] = 0
    c = rnd.randrange(cp.r)
    fold = ctx.points_fold_g1 if group == 1 else ctx.points_fold_g2
    want = [fold(lo, hi, c, n=n).copy() for lo, hi in zip(los, his)]
    outs = [np.zeros(n * pb, np.uint8) for _ in range(3)]
    ctx.points_fold_many(group, los, his, c, n, outs)
    for y in range(3):
        assert np.array_equal(outs[y], want[y]), (cname, group, y)
    dev = [DeviceBuffer(ctx, n * pb) for _ in range(3)]
    ctx.points_fold_many(group, los, his, c, n, dev)
    for y in range(3):
        assert np.array_equal(np.frombuffer(dev[y].to_host(), np.uint8), want[y]), (cname, group, y, "device out")
        dev[y].free()


def test_the_longest_vector_of_the_k_lane_form(ctx_bn254, monkeypatch):
    """n K = 65 536 lanes exactly (G2: n = 16 384; G1: n = 32 768) is the last size k_points_mul_split takes; one element
    more goes to the one-lane kernels.  Both sizes, both groups: the same bytes as the one-lane form, for the fold and for
    `scalar_pairing`."""
    ctx = ctx_bn254
    cp, fc, cd = CURVES["bn254"], FrCodec("bn254"), Codec(CURVES["bn254"])
    rnd = random.Random(41)
    # G2 has a third form below them: 16 n <= 65 536 lanes (n <= 4 096) runs with every Fq2 value on a quad of lanes
    for group, sizes in ((2, (4096, 4097, 16384, 16385)), (1, (32768, 32769))):
        gen = cp.g1_gen if group == 1 else cp.g2_gen
        vec = (cd.g1_vec if group == 1 else cd.g2_vec)([gen])
        fold = ctx.points_fold_g1 if group == 1 else ctx.points_fold_g2
        for n in sizes:
            hi = ctx.fixed_base(group, vec, fc.enc([rnd.randrange(1, cp.r) for _ in range(n)])).copy()
            lo = ctx.fixed_base(group, vec, fc.enc([rnd.randrange(1, cp.r) for _ in range(n)])).copy()
            scal = fc.enc([rnd.randrange(cp.r) for _ in range(n)])
            c = rnd.randrange(cp.r)
            monkeypatch.delenv("HK_ENDO_ONE_LANE", raising=False)
            a = (fold(lo, hi, c, n=n).copy(), ctx.scalar_pairing(group, hi, scal, n=n).copy())
            monkeypatch.setenv("HK_ENDO_ONE_LANE", "1")
            b = (fold(lo, hi, c, n=n).copy(), ctx.scalar_pairing(group, hi, scal, n=n).copy())
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (group, n)
