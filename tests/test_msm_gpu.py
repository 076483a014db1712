"""GPU parity: hk_msm_g1 / hk_msm_g2 (through the C ABI) vs the big-int oracle.

Mirrors the arkworks calls the reference makes: `G::Group::msm_bigint` (canonical scalars,
cp-groth16/src/prover.rs:167), `E::G1::msm` (Montgomery scalars, prover.rs:117,129;
committer.rs:89) and `msm_unchecked` (committer.rs:113).  Bit-exact after affine normalisation.
"""
import random

import numpy as np
import pytest

from hekaton_system_amd import capi
from oracle.pyref import curve
from oracle.pyref.codec import Codec
from oracle.pyref.params import CURVES
from tests.util import running_bases, mixed_scalars

pytestmark = pytest.mark.gpu


def _ctx(name, ctx_bn254, ctx_bls):
    return ctx_bn254 if name == "bn254" else ctx_bls


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 300])
@pytest.mark.parametrize("group", ["g1", "g2"])
def test_msm_small_vs_oracle(group, n, ctx_bn254):
    _msm_case("bn254", group, n, ctx_bn254, None)


@pytest.mark.parametrize("n", [33, 300])
@pytest.mark.parametrize("group", ["g1", "g2"])
def test_msm_small_by_the_bucket_method(group, n, ctx_bn254, monkeypatch):
    """Short one-off MSMs run as element-wise products + a sum (MsmRun::small_msm) by default; HK_MSM_NO_SMALL sends them
    through the Pippenger pass the long ones take - the same cases, the same oracle."""
    monkeypatch.setenv("HK_MSM_NO_SMALL", "1")
    _msm_case("bn254", group, n, ctx_bn254, None)


def _msm_case(cname, group, n, ctx_bn254, ctx_bls):
    cp = CURVES[cname]
    ctx = _ctx(cname, ctx_bn254, ctx_bls)
    cd = Codec(cp)
    G = curve.G1(cp) if group == "g1" else curve.G2(cp)
    rnd = random.Random(1000 * n + len(cname))
    bases = running_bases(G, n, s0=5)
    scalars = mixed_scalars(rnd, cp.r, n, dense=(n % 2 == 0))
    # edge cases the reference's domain has: zero / one / r-1 scalars, infinity and repeated bases
    if n >= 31:
        scalars[0], scalars[1], scalars[2] = 0, 1, cp.r - 1
        bases[3] = None
        bases[5] = bases[4]
        bases[7] = G.neg(bases[6]); scalars[7] = scalars[6]
    enc = cd.g1_vec if group == "g1" else cd.g2_vec
    dec = cd.g1_from if group == "g1" else cd.g2_from
    fn = ctx.msm_g1 if group == "g1" else ctx.msm_g2
    want = G.msm(bases, scalars)
    got_m = dec(fn(enc(bases), cd.fr_vec_mont(scalars), montgomery=True))
    got_c = dec(fn(enc(bases), cd.fr_vec_canon(scalars), montgomery=False))
    assert got_m == want
    assert got_c == want


def test_msm_length_semantics(ctx_bn254):
    """ark `msm` -> Err(min_len) on mismatch (HK_ERR_LEN); `msm_unchecked` zips to the shorter."""
    cp = CURVES["bn254"]
    cd = Codec(cp)
    G = curve.G1(cp)
    bases = running_bases(G, 4)
    scalars = [7, 8, 9]
    with pytest.raises(capi.HekatonError) as e:
        ctx_bn254.msm_g1(cd.g1_vec(bases), cd.fr_vec_mont(scalars), checked=True)
    assert e.value.status == capi.HK_ERR_LEN
    got = cd.g1_from(ctx_bn254.msm_g1(cd.g1_vec(bases), cd.fr_vec_mont(scalars), checked=False))
    assert got == G.msm(bases[:3], scalars)
    # committer.rs:110-113: deltas_g has one more element than comm_rands
    got = cd.g1_from(ctx_bn254.msm_g1(cd.g1_vec(bases[:2]), cd.fr_vec_mont([5]), checked=False))
    assert got == G.mul(bases[0], 5)
    # empty
    assert cd.g1_from(ctx_bn254.msm_g1(np.zeros(0, np.uint8), np.zeros(0, np.uint8))) is None


def test_msm_device_resident_inputs(ctx_bn254):
    cp = CURVES["bn254"]
    cd = Codec(cp)
    G = curve.G1(cp)
    rnd = random.Random(5)
    bases = running_bases(G, 64)
    scalars = [rnd.randrange(cp.r) for _ in range(64)]
    db = capi.DeviceBuffer.from_host(ctx_bn254, cd.g1_vec(bases))
    ds = capi.DeviceBuffer.from_host(ctx_bn254, cd.fr_vec_mont(scalars))
    got = cd.g1_from(ctx_bn254.msm_g1(db, ds, n_bases=64, n_scalars=64))
    assert got == G.msm(bases, scalars)
    db.free(); ds.free()


@pytest.mark.parametrize("n", [1, 33, 300])
@pytest.mark.parametrize("group", ["g1", "g2"])
def test_msm_small_vs_oracle_bls12_381(group, n, ctx_bls):
    _msm_case("bls12_381", group, n, None, ctx_bls)
