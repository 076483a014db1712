"""CPU: the C++ oracle (oracle/c/hk_oracle.cpp — large-n checker and CPU baseline) against the
golden fixtures produced by the big-int oracle."""
import numpy as np
import pytest

from oracle.c_oracle import COracle
from tests import golden_util as gu


@pytest.fixture(scope="module", params=["bn254", "bls12_381"])
def co(request):
    return COracle(request.param), request.param


def test_msm_golden(co):
    o, name = co
    for case in gu.load("msm.json")[name]:
        g = 1 if case["group"] == "g1" else 2
        for key, mont in (("scalars_mont", True), ("scalars_canon", False)):
            got = o.msm(g, gu.hb(case["bases"]), gu.hb(case[key]), montgomery=mont)
            assert got.tobytes().hex() == case["expect"], (name, case["group"], case["n"], key)


def test_msm_thread_count_independent(co):
    o, name = co
    case = [c for c in gu.load("msm.json")[name] if c["group"] == "g1" and c["n"] == 200][0]
    o.set_threads(1)
    a = o.msm(1, gu.hb(case["bases"]), gu.hb(case["scalars_mont"]))
    o.set_threads(3)
    b = o.msm(1, gu.hb(case["bases"]), gu.hb(case["scalars_mont"]))
    o.set_threads(0)
    assert a.tobytes() == b.tobytes() == bytes.fromhex(case["expect"])


def test_ntt_golden(co):
    o, name = co
    for case in gu.load("ntt.json")[name]:
        for key, inv, coset in (("fft", 0, 0), ("ifft", 1, 0), ("coset_fft", 0, 1), ("coset_ifft", 1, 1)):
            buf = gu.hb(case["input"])
            o.ntt(buf, case["log_m"], inverse=inv, coset=coset)
            assert buf.tobytes().hex() == case[key], (name, case["log_m"], key)


def test_ntt_degree_too_large(co):
    o, name = co
    with pytest.raises(ValueError):
        o.ntt(np.zeros(32, np.uint8), 33)


def test_groth16_golden(co):
    o, name = co
    for case in gu.load("groth16.json")[name]:
        A, B, C = gu.csr(case["A"]), gu.csr(case["B"]), gu.csr(case["C"])
        z = gu.hb(case["z_mont"])
        h, m = o.witness_map(A, B, C, case["n_inst"], case["n_constraints"], z)
        assert h.tobytes().hex() == case["h_mont"]
        pk = case["pk"]
        view = o.pk_view(a_g=gu.hb(pk["a_g"]), b_g=gu.hb(pk["b_g"]), b_h=gu.hb(pk["b_h"]), h_g=gu.hb(pk["h_g"]),
                         ck_stages=[gu.hb(c) for c in pk["ck"]], deltas_g=gu.hb(pk["deltas_g"]),
                         last_delta_h=gu.hb(pk["last_delta_h"]), alpha_g=gu.hb(pk["alpha_g"]),
                         beta_g=gu.hb(pk["beta_g"]), beta_h=gu.hb(pk["beta_h"]))
        kap = gu.hb(case["kappas_mont"])
        fr = o.fr_bytes
        for k, (s, e) in enumerate(case["stage_ranges"][:-1]):
            w = z[(case["n_inst"] + s) * fr:(case["n_inst"] + e) * fr]
            com = o.commit(view, k, w, kap[k * fr:(k + 1) * fr])
            assert com.tobytes().hex() == case["comms"][k]
        a, b, c = o.prove(view, A, B, C, case["n_inst"], case["n_constraints"], z, gu.hb(case["r_mont"]),
                          gu.hb(case["s_mont"]), kap)
        assert (a.tobytes().hex(), b.tobytes().hex(), c.tobytes().hex()) == \
               (case["proof"]["a"], case["proof"]["b"], case["proof"]["c"]), case["label"]


def test_pairing_matches_the_tower_oracle(co):
    """hko_multi_pairing (ark's chunked multi_miller_loop + final exponentiation, 64-bit limbs) against
    oracle/pyref/pairing.py on 0, 1, 5 (one chunk + one pair, with infinity members) and 9 pairs."""
    import random
    from oracle.pyref import curve, pairing
    from oracle.pyref.params import CURVES
    from tests.test_pairing_cpu import Enc
    o, name = co
    cp = CURVES[name]
    T = pairing.tower(name)
    E = Enc(cp)
    G1, G2 = curve.G1(cp), curve.G2(cp)
    rnd = random.Random(31)
    for n, with_inf in ((0, False), (1, False), (5, True), (9, False)):
        ps = [G1.mul(cp.g1_gen, rnd.randrange(1, cp.r)) for _ in range(n)]
        qs = [G2.mul(cp.g2_gen, rnd.randrange(1, cp.r)) for _ in range(n)]
        if with_inf:
            ps[0] = None
            qs[4] = None
        g1 = np.frombuffer(b"".join(E.g1(p) for p in ps) or bytes(2 * E.nb), np.uint8)
        g2 = np.frombuffer(b"".join(E.g2(q) for q in qs) or bytes(4 * E.nb), np.uint8)
        got = E.f12_dec(o.multi_pairing(g1, g2, n=n).tobytes())
        assert got == T.f12_flat(T.multi_pairing(list(zip(ps, qs)))), (name, n)
