"""CPU: the PRODUCT's tower / pairing code (csrc/tower.cuh, csrc/pairing.cuh compiled for the host with g++ — the same
source the kernels run; the library itself never runs it on the CPU) against the tower oracle
(oracle/pyref/pairing.py): Fq12 ring operations, Frobenius maps, the final-exponentiation chain, and whole
multi-pairings with infinity members, on both curves."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle.pyref import curve, pairing
from oracle.pyref.params import CURVES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CID = {"bn254": 0, "bls12_381": 1}


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("shim") / "field_shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out,
                           os.path.join(ROOT, "tests", "host_shim", "field_shim.cpp")])
    lib = ctypes.CDLL(out)
    lib.shim_multi_pairing.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    return lib


class Enc:
    """ints <-> Montgomery little-endian bytes of Fq, and the packed forms the C ABI uses."""

    def __init__(self, cp):
        self.q = cp.q
        self.nb = 8 * cp.fq_limbs64
        self.R = 1 << (8 * self.nb)
        self.Ri = pow(self.R, -1, cp.q)

    def fq(self, x): return (x % self.q * self.R % self.q).to_bytes(self.nb, "little")
    def f12(self, flat): return b"".join(self.fq(x) for x in flat)
    def f12_dec(self, b): return [int.from_bytes(b[i:i + self.nb], "little") * self.Ri % self.q for i in range(0, len(b), self.nb)]
    def g1(self, P): return bytes(2 * self.nb) if P is None else self.fq(P[0]) + self.fq(P[1])
    def g2(self, Q): return bytes(4 * self.nb) if Q is None else b"".join(self.fq(c) for c in (Q[0][0], Q[0][1], Q[1][0], Q[1][1]))


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_fq12_ring_ops_and_final_exponentiation(cname, shim):
    cp = CURVES[cname]
    T = pairing.tower(cname)
    E = Enc(cp)
    rnd = random.Random(12)
    a = [rnd.randrange(cp.q) for _ in range(12)]
    b = [rnd.randrange(cp.q) for _ in range(12)]
    A, B = T.f12_from_flat(a), T.f12_from_flat(b)

    def op(code, x, y=None):
        out = ctypes.create_string_buffer(12 * E.nb)
        shim.shim_f12_op(CID[cname], code, E.f12(x), E.f12(y) if y is not None else None, out)
        return E.f12_dec(out.raw)

    assert op(0, a, b) == T.f12_flat(T.f12_mul(A, B))
    assert op(1, a) == T.f12_flat(T.f12_sqr(A))
    assert op(2, a) == T.f12_flat(T.f12_inv(A))
    assert op(3, a) == T.f12_flat(T.f12_conj(A))
    for k in (1, 2, 3):
        assert op(3 + k, a) == T.f12_flat(T.f12_frob(A, k))
    assert op(7, a) == T.f12_flat(T.f12_pow(A, T.x))
    assert op(8, a) == T.f12_flat(T.final_exponentiation(A))
    sparse = [1] + [0] * 11
    assert op(0, a, sparse) == a and op(2, sparse) == sparse


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_multi_pairing_matches_the_oracle(cname, shim):
    cp = CURVES[cname]
    T = pairing.tower(cname)
    E = Enc(cp)
    G1, G2 = curve.G1(cp), curve.G2(cp)
    rnd = random.Random(5)
    for n, with_inf in ((1, False), (2, False), (5, True), (0, False)):
        ps = [G1.mul(cp.g1_gen, rnd.randrange(1, cp.r)) for _ in range(n)]
        qs = [G2.mul(cp.g2_gen, rnd.randrange(1, cp.r)) for _ in range(n)]
        if with_inf:
            ps[1] = None
            qs[3] = None
        out = ctypes.create_string_buffer(12 * E.nb)
        shim.shim_multi_pairing(CID[cname], b"".join(E.g1(p) for p in ps) or b"\0", b"".join(E.g2(q) for q in qs) or b"\0",
                                n, out)
        assert E.f12_dec(out.raw) == T.f12_flat(T.multi_pairing(list(zip(ps, qs)))), (cname, n)
