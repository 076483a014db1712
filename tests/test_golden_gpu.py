"""GPU: the HIP path (through the C ABI) against the committed golden fixtures, both curves.
tests/golden/*.json were produced by the big-int oracle (tests/golden/gen_golden.py)."""
import numpy as np
import pytest

from tests import golden_util as gu

pytestmark = pytest.mark.gpu


def _ctx(name, ctx_bn254, ctx_bls):
    return ctx_bn254 if name == "bn254" else ctx_bls


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_msm_golden(cname, ctx_bn254, ctx_bls):
    ctx = _ctx(cname, ctx_bn254, ctx_bls)
    for case in gu.load("msm.json")[cname]:
        fn = ctx.msm_g1 if case["group"] == "g1" else ctx.msm_g2
        for key, mont in (("scalars_mont", True), ("scalars_canon", False)):
            got = fn(gu.hb(case["bases"]), gu.hb(case[key]), montgomery=mont)
            assert got.tobytes().hex() == case["expect"], (cname, case["group"], case["n"], key)


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_ntt_golden(cname, ctx_bn254, ctx_bls):
    ctx = _ctx(cname, ctx_bn254, ctx_bls)
    for case in gu.load("ntt.json")[cname]:
        for key, inv, coset in (("fft", 0, 0), ("ifft", 1, 0), ("coset_fft", 0, 1), ("coset_ifft", 1, 1)):
            buf = gu.hb(case["input"])
            ctx.ntt(buf, case["log_m"], inverse=inv, coset=coset)
            assert buf.tobytes().hex() == case[key], (cname, case["log_m"], key)


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_groth16_golden(cname, ctx_bn254, ctx_bls):
    """witness map, every stage commitment and the proof, byte for byte."""
    ctx = _ctx(cname, ctx_bn254, ctx_bls)
    fr = ctx.fr_bytes
    for case in gu.load("groth16.json")[cname]:
        A, B, C = gu.csr(case["A"]), gu.csr(case["B"]), gu.csr(case["C"])
        z = gu.hb(case["z_mont"])
        h, m = ctx.witness_map(A, B, C, case["n_inst"], case["n_constraints"], z)
        assert h.tobytes().hex() == case["h_mont"], (cname, case["label"])
        pk = case["pk"]
        dpk = ctx.pk_upload(a_g=gu.hb(pk["a_g"]), b_g=gu.hb(pk["b_g"]), b_h=gu.hb(pk["b_h"]), h_g=gu.hb(pk["h_g"]),
                            ck_stages=[gu.hb(c) for c in pk["ck"]], deltas_g=gu.hb(pk["deltas_g"]),
                            last_delta_h=gu.hb(pk["last_delta_h"]), alpha_g=gu.hb(pk["alpha_g"]),
                            beta_g=gu.hb(pk["beta_g"]), beta_h=gu.hb(pk["beta_h"]), matrices=(A, B, C),
                            n_inst=case["n_inst"], n_constraints=case["n_constraints"])
        kap = gu.hb(case["kappas_mont"])
        for k, (s, e) in enumerate(case["stage_ranges"][:-1]):
            w = z[(case["n_inst"] + s) * fr:(case["n_inst"] + e) * fr]
            com = dpk.commit(k, w, kap[k * fr:(k + 1) * fr], n=e - s)
            assert com.tobytes().hex() == case["comms"][k], (cname, case["label"], "commit")
        a, b, c = dpk.prove(z, gu.hb(case["r_mont"]), gu.hb(case["s_mont"]), kap)
        assert (a.tobytes().hex(), b.tobytes().hex(), c.tobytes().hex()) == \
               (case["proof"]["a"], case["proof"]["b"], case["proof"]["c"]), (cname, case["label"])
        dpk.free()
