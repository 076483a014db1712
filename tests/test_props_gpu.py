"""GPU: size-independent properties at BASELINE sizes and schedule coverage of the NTT pass kernel.

  * every transform size 2^0 .. 2^15 and the BASELINE domain sizes 2^21 / 2^22 against the CPU oracle (each size
    picks a different pass schedule: one tile, 11 + k, 11 + 5 + 5, 11 + 6 + 5), forward / inverse, plain / coset;
  * linearity and round trips of the NTT at 2^21, the convolution theorem on a small size;
  * MSM linearity in the scalars at n = 2^20 (the metric's MSM size class), on both groups;
  * the B-query compaction (B1 / B2 over the non-infinity bases only) gives the bytes of the dense path.
"""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from hekaton_system_amd import capi
from hekaton_system_amd.cp_groth16 import CURVE_PARAMS, FrCodec
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rand_fr_bytes(seed, m, top_mask):
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
    raw[:, 31] &= top_mask                                  # < r (canonical Montgomery bytes of SOME element)
    return raw.ravel().copy()


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_ntt_every_small_size_all_four_modes(cname, ctx_bn254, ctx_bls):
    ctx = ctx_bn254 if cname == "bn254" else ctx_bls
    co = COracle(cname)
    for log_m in range(0, 16):
        x = _rand_fr_bytes(100 + log_m, 1 << log_m, 0x0f)
        for inverse in (False, True):
            for coset in (False, True):
                got = ctx.ntt(x.copy(), log_m, inverse=inverse, coset=coset)
                want = co.ntt(x.copy(), log_m, inverse=inverse, coset=coset)
                assert np.array_equal(got, want), (cname, log_m, inverse, coset)


@pytest.mark.parametrize("log_m", [21, 22])
def test_ntt_baseline_sizes_vs_oracle_and_round_trip(log_m, ctx_bn254):
    m = 1 << log_m
    x = _rand_fr_bytes(log_m, m, 0x0f)
    co = COracle("bn254")
    dev = capi.DeviceBuffer.from_host(ctx_bn254, x)
    ctx_bn254.ntt(dev, log_m, inverse=True, coset=False)
    assert np.array_equal(dev.to_host(), co.ntt(x.copy(), log_m, inverse=True, coset=False))
    ctx_bn254.ntt(dev, log_m, inverse=False, coset=False)
    assert np.array_equal(dev.to_host(), x)
    ctx_bn254.ntt(dev, log_m, inverse=False, coset=True)
    ctx_bn254.ntt(dev, log_m, inverse=True, coset=True)
    assert np.array_equal(dev.to_host(), x)
    dev.free()


def test_ntt_linearity_at_2_21_and_convolution_theorem(ctx_bn254):
    fc = FrCodec("bn254")
    r = fc.r
    log_m = 21
    m = 1 << log_m
    # linearity: NTT(x + c*e_k) = NTT(x) + c * (w^(k*i))_i ; checked through the inverse: NTT(x) + NTT(y) on a sparse y
    x = _rand_fr_bytes(7, m, 0x0f)
    k, c = 123457, 0x1234567
    y = np.zeros(m * 32, dtype=np.uint8)
    y[32 * k:32 * k + 32] = fc.enc1(c)
    xs = x.copy()
    xk = fc.dec(x[32 * k:32 * k + 32])[0]
    xs[32 * k:32 * k + 32] = fc.enc1((xk + c) % r)                      # x + y
    fx = ctx_bn254.ntt(x.copy(), log_m)
    fy = ctx_bn254.ntt(y.copy(), log_m)
    fs = ctx_bn254.ntt(xs, log_m)
    idx = [0, 1, 2, 77, m // 2, m - 1] + [random.Random(3).randrange(m) for _ in range(60)]
    for i in idx:
        a, b, s = (fc.dec(v[32 * i:32 * i + 32])[0] for v in (fx, fy, fs))
        assert (a + b) % r == s
    # convolution theorem on 2^10: iNTT(NTT(a) o NTT(b)) = cyclic convolution
    lg = 10
    n = 1 << lg
    rnd = random.Random(5)
    a = [rnd.randrange(r) for _ in range(n)]
    b = [0] * n
    for j in (0, 1, 5, 1000):
        b[j] = rnd.randrange(r)
    fa = fc.dec(ctx_bn254.ntt(fc.enc(a), lg))
    fb = fc.dec(ctx_bn254.ntt(fc.enc(b), lg))
    prod = fc.enc([u * v % r for u, v in zip(fa, fb)])
    conv = fc.dec(ctx_bn254.ntt(prod, lg, inverse=True))
    want = [0] * n
    for j in (0, 1, 5, 1000):
        for i in range(n):
            want[(i + j) % n] = (want[(i + j) % n] + a[i] * b[j]) % r
    assert conv == want


@pytest.mark.parametrize("group", [1, 2])
def test_msm_linearity_at_2_20(group, ctx_bn254):
    """msm(bases, s1) + msm(bases, s2) = msm(bases, s1 + s2) and the closed form (sum s_i k_i) * G, n = 2^20 (G1) /
    2^17 (G2); scalars in the SURVEY §8d mixture."""
    fc = FrCodec("bn254")
    p = CURVE_PARAMS["bn254"]
    r = p["r"]
    n = 1 << (20 if group == 1 else 17)
    rnd = random.Random(group)
    gen = fc.g1(p["g1"]) if group == 1 else fc.g2(p["g2"])
    ks = [rnd.randrange(1, r) for _ in range(n)]
    pb = ctx_bn254.g1_bytes if group == 1 else ctx_bn254.g2_bytes
    bases = capi.DeviceBuffer(ctx_bn254, n * pb)
    ctx_bn254.fixed_base(group, gen, fc.enc(ks), out=bases)
    s1 = [rnd.randrange(r) if rnd.random() < 0.15 else rnd.randrange(2) for _ in range(n)]
    s2 = [rnd.randrange(r) if rnd.random() < 0.15 else rnd.randrange(2) for _ in range(n)]
    msm = ctx_bn254.msm_g1 if group == 1 else ctx_bn254.msm_g2
    r1 = msm(bases, fc.enc(s1), n_bases=n)
    r2 = msm(bases, fc.enc(s2), n_bases=n)
    r12 = msm(bases, fc.enc([(a + b) % r for a, b in zip(s1, s2)]), n_bases=n)
    closed = lambda s: ctx_bn254.fixed_base(group, gen, fc.enc([sum(x * k for x, k in zip(s, ks)) % r]))
    assert np.array_equal(r1, closed(s1)) and np.array_equal(r2, closed(s2))
    assert np.array_equal(r12, closed([(a + b) % r for a, b in zip(s1, s2)]))
    # r1 + r2 == r12 through a 2-term MSM with unit scalars
    two = np.concatenate([r1, r2])
    assert np.array_equal(msm(two, fc.enc([1, 1])), r12)
    bases.free()


_CHILD = r"""
import json, sys
sys.path.insert(0, %(root)r)
import numpy as np
from hekaton_system_amd import capi
from hekaton_system_amd.cp_groth16 import FrCodec, SeededRng, generate_parameters
from hekaton_system_amd.workload import SyntheticSubcircuit
ctx = capi.Context("bn254", 0)
circ = SyntheticSubcircuit("bn254", n_c=20000, n_free=3000, n0=16)
pk, _ = generate_parameters(circ, "bn254", SeededRng(b"\x21" * 32), ctx)
dpk = pk.upload(ctx)
fc = FrCodec("bn254")
out = []
for seed in (1, 2):
    circ.set_witness_seed(seed)
    com = dpk.commit(0, circ.stage0_witness_bytes(), fc.enc([9]), n=circ.n0)
    a, b, c = dpk.prove(circ.full_assignment_bytes(), fc.enc1(5), fc.enc1(6), fc.enc([9]), n_v=circ.n_v)
    out.append([bytes(x).hex() for x in (com, a, b, c)])
print(json.dumps({"density": circ.query_density(), "proofs": out}))
"""


def test_b_query_compaction_gives_the_bytes_of_the_dense_path():
    """Same key, same assignments, once with the compacted B1 / B2 (default: B is ~50 %% populated here) and once
    with HK_B_COMPACT_BELOW=0 (all bases through the shared digit sort): identical commitments and proofs."""
    outs = []
    for thr in (None, "0"):
        env = dict(os.environ)
        if thr is not None:
            env["HK_B_COMPACT_BELOW"] = thr
        else:
            env.pop("HK_B_COMPACT_BELOW", None)
        res = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=env, capture_output=True, text=True,
                             timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(json.loads(res.stdout.strip().splitlines()[-1]))
    assert outs[0]["density"][1] < 0.75                     # the default run really took the compacted path
    assert outs[0]["proofs"] == outs[1]["proofs"]
