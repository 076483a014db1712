"""CPU: the C-ABI library loads, exports every symbol include/hekaton.h declares, refuses to run
without a GPU (no CPU fallback), and its generated constants agree with the oracle's."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from hekaton_system_amd import capi
from oracle.pyref.params import CURVES
from oracle.pyref.codec import Codec
from tests import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    hdr = open(os.path.join(ROOT, "include", "hekaton.h")).read()
    declared = set(re.findall(r"\b(hk_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert b"gfx950" in lib.hk_version()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.HekatonError) as e:
        capi.Context("bn254", 0)
    assert e.value.status == capi.HK_ERR_DEVICE


def test_status_strings():
    assert capi.status_str(capi.HK_ERR_LEN) == "HK_ERR_LEN"
    assert capi.status_str(capi.HK_ERR_DOMAIN_TOO_LARGE) == "HK_ERR_DOMAIN_TOO_LARGE"


def test_generated_params_match_oracle():
    hdr = open(os.path.join(ROOT, "hekaton_system_amd", "csrc", "hk_params.h")).read()

    def limbs(name):
        m = re.search(r"#define %s\s+\{([^}]*)\}" % name, hdr)
        vals = [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]
        return sum(v << (32 * i) for i, v in enumerate(vals))

    for key, cname in (("BN254", "bn254"), ("BLS12_381", "bls12_381")):
        cp = CURVES[cname]
        assert limbs("HK_%s_FR_MOD" % key) == cp.r
        assert limbs("HK_%s_FQ_MOD" % key) == cp.q
        assert limbs("HK_%s_FR_ONE" % key) == cp.fr_R % cp.r
        assert limbs("HK_%s_FQ_ONE" % key) == cp.fq_R % cp.q
        assert limbs("HK_%s_FR_ROOT" % key) == cp.two_adic_root * cp.fr_R % cp.r
        assert limbs("HK_%s_FR_GEN" % key) == cp.fr_generator * cp.fr_R % cp.r


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    """The product's field.cuh / ec.cuh compiled for the host (g++) — same source the kernels run."""
    out = str(tmp_path_factory.mktemp("shim") / "field_shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out,
                           os.path.join(ROOT, "tests", "host_shim", "field_shim.cpp")])
    return ctypes.CDLL(out)


def test_product_field_and_curve_arithmetic_vs_golden(shim):
    kat = gu.load("field_curve_kat.json")
    ids = {"bn254": dict(fr=0, fq=1, fq2=4, g1=0, g2=1), "bls12_381": dict(fr=2, fq=3, fq2=5, g1=2, g2=3)}

    def fop(fid, op, a, b, n):
        out = ctypes.create_string_buffer(n)
        shim.shim_field_op(fid, op, bytes(a), bytes(b), out)
        return out.raw.hex()

    def gop(gid, op, a, b, n):
        out = ctypes.create_string_buffer(n)
        shim.shim_group_op(gid, op, bytes(a), bytes(b), out)
        return out.raw.hex()

    for cname, ent in kat.items():
        for key in ("fr", "fq"):
            for c in ent[key]:
                a, b = gu.hb(c["a"]), gu.hb(c["b"])
                n = len(a)
                assert fop(ids[cname][key], 0, a, b, n) == c["add"]
                assert fop(ids[cname][key], 1, a, b, n) == c["sub"]
                assert fop(ids[cname][key], 2, a, b, n) == c["mul"]
                if c["inv_a"]:
                    assert fop(ids[cname][key], 4, a, a, n) == c["inv_a"]
        for c in ent["fq2"]:
            a, b = gu.hb(c["a"]), gu.hb(c["b"])
            assert fop(ids[cname]["fq2"], 2, a, b, len(a)) == c["mul"]
            assert fop(ids[cname]["fq2"], 5, a, a, len(a)) == c["sqr_a"]
            assert fop(ids[cname]["fq2"], 4, a, a, len(a)) == c["inv_a"]
        for key in ("g1", "g2"):
            for c in ent[key]:
                p, q = gu.hb(c["p"]), gu.hb(c["q"])
                assert gop(ids[cname][key], 0, p, q, len(p)) == c["add"]       # mixed add
                assert gop(ids[cname][key], 1, p, q, len(p)) == c["add"]       # full XYZZ add
                assert gop(ids[cname][key], 0, p, p, len(p)) == c["dbl_p"]     # P + P corner


def test_binary_inversion_equals_fermat_and_python(shim):
    """csrc/ec.cuh `fp_inv` (binary extended Euclid on the Montgomery value's integer) against a^(p-2) from the same
    source and against Python's modular inverse, on edge values and random ones, all four prime fields; 0 -> 0."""
    import random
    from oracle.pyref.params import CURVES
    rnd = random.Random(99)
    for cname, (fr_id, fq_id) in (("bn254", (0, 1)), ("bls12_381", (2, 3))):
        cp = CURVES[cname]
        for fid, mod, R in ((fr_id, cp.r, cp.fr_R), (fq_id, cp.q, cp.fq_R)):
            nb = 32 if mod.bit_length() <= 256 else 48
            vals = [0, 1, 2, mod - 1, mod - 2, (mod + 1) // 2, 1 << 64, (1 << (mod.bit_length() - 1)) - 1] + \
                   [rnd.randrange(1, mod) for _ in range(200)]
            for x in vals:
                a = (x * R % mod).to_bytes(nb, "little")
                out, ref = ctypes.create_string_buffer(nb), ctypes.create_string_buffer(nb)
                shim.shim_field_op(fid, 4, a, a, out)
                shim.shim_field_op(fid, 6, a, a, ref)
                got = int.from_bytes(out.raw, "little") % mod
                assert got == int.from_bytes(ref.raw, "little") % mod, (cname, fid, x)
                want = pow(x, -1, mod) * R % mod if x else 0
                assert got == want, (cname, fid, x)


def test_msm_window_count_is_the_smallest_that_holds_every_scalar(tmp_path_factory):
    """csrc/msm.cuh `msm_num_windows` (host code, compiled here with g++): with the signed-digit recoding
    s' = s + sum_{w<W} 2^(cw+c-1), W windows are enough iff (r - 1) + that constant < 2^(cW).  The plan must pick the
    SMALLEST such W that is >= the always-sufficient ceil((bits+2)/c) - 1, for every window size either curve can get:
    16 windows at c = 16 on both (BLS12-381's 17th digit would always be zero), and never a W that lets s' overflow."""
    import ctypes
    import subprocess
    from oracle.pyref.params import CURVES
    out = str(tmp_path_factory.mktemp("shim") / "field_shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out,
                           os.path.join(ROOT, "tests", "host_shim", "field_shim.cpp")])
    lib = ctypes.CDLL(out)
    buf = (ctypes.c_uint * 16)()
    for cid, name, bits in ((0, "bn254", 254), (1, "bls12_381", 255)):
        r = CURVES[name].r
        assert r.bit_length() == bits
        for c in range(3, 17):
            lib.shim_msm_plan(cid, 1 << 20, c, 1, 1 << 18, buf)
            W = buf[0]
            kconst = sum(buf[6 + i] << (32 * i) for i in range(10))
            assert kconst == sum(1 << (c * w + c - 1) for w in range(W))
            fits = lambda w: (r - 1) + sum(1 << (c * k + c - 1) for k in range(w)) < (1 << (c * w))
            assert fits(W), (name, c, W)
            upper = (bits + 2 + c - 1) // c
            assert W in (upper, upper - 1)
            if W == upper:
                assert not fits(upper - 1), (name, c, "one window fewer would have been enough")
            # every digit of the extreme scalars is in range and the top window never carries out
            for s in (0, 1, r - 1, r - 2, (1 << (bits - 1)) - 1):
                sp = s + kconst
                assert sp >> (c * W) == 0
                digs = [((sp >> (c * w)) & ((1 << c) - 1)) - (1 << (c - 1)) for w in range(W)]
                assert sum(d << (c * w) for w, d in enumerate(digs)) == s
        lib.shim_msm_plan(cid, 1 << 21, 16, 1, 1 << 18, buf)
        assert buf[0] == 16 and buf[1] == 16 and buf[2] == 1 << 15
