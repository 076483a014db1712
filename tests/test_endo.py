"""The G2 endomorphism behind hk_points_fold_g2 (hekaton_system_amd/endo.py, csrc/pairing.cuh g2_psi), checked on the CPU
against the oracle's curve arithmetic: psi(Q) = (conj(x) * PSI_X, conj(y) * PSI_Y) has the eigenvalue q mod r on G2 (6 x^2
on BN254, x on BLS12-381) with exactly the constants the generator emits for each twist type, and every scalar splits into
four parts of at most 65 bits that recombine to it."""
import random
import re

import pytest

from hekaton_system_amd.endo import eigenvalue, psi4
from oracle.pyref import curve
from oracle.pyref.params import CURVES


def _f2m(a, b, q):
    return ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)


def _psi_constants(name):
    """(PSI_X, PSI_Y) as the product's generated header holds them (Montgomery limbs -> ints)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "hekaton_system_amd", "csrc", "hk_tower_params.h")).read()
    cp = CURVES[name]
    pre = "HK_%s_TW" % ("BN254" if name == "bn254" else "BLS12_381")
    out = []
    for key in ("PSI_X", "PSI_Y"):
        m = re.search(r"#define %s_%s \{ \{ ([^}]*) \}, \{ ([^}]*) \} \}" % (pre, key), txt)
        comps = []
        for g in m.groups():
            limbs = [int(x.strip().rstrip("u"), 16) for x in g.split(",")]
            v = sum(l << (32 * i) for i, l in enumerate(limbs))
            comps.append(v * pow(1 << (32 * len(limbs)), -1, cp.q) % cp.q)
        out.append(tuple(comps))
    return out


@pytest.mark.parametrize("name", ["bn254", "bls12_381"])
def test_psi_has_the_eigenvalue_q_mod_r_on_g2(name):
    cp = CURVES[name]
    G2 = curve.G2(cp)
    lam = eigenvalue(name)
    assert lam == cp.q % cp.r
    px, py = _psi_constants(name)
    rnd = random.Random(2)
    for _ in range(3):
        Q = G2.mul(cp.g2_gen, rnd.randrange(1, cp.r))
        conj = lambda a: (a[0], (-a[1]) % cp.q)
        img = (_f2m(conj(Q[0]), px, cp.q), _f2m(conj(Q[1]), py, cp.q))
        want = G2.mul(Q, lam)
        assert img == (want[0], want[1])


@pytest.mark.parametrize("name", ["bn254", "bls12_381"])
def test_every_scalar_splits_into_four_short_parts(name):
    P = psi4(name)
    rnd = random.Random(5)
    for c in [0, 1, P.r - 1, P.lam, P.r - P.lam, 1 << 200] + [rnd.randrange(P.r) for _ in range(500)]:
        k = P.decompose(c)
        assert sum(kj * pow(P.lam, j, P.r) for j, kj in enumerate(k)) % P.r == c % P.r
        assert max(abs(x) for x in k).bit_length() <= 65


@pytest.mark.parametrize("name", ["bn254", "bls12_381"])
def test_g1_endomorphism_eigenvalue_and_two_part_split(name):
    """phi(x, y) = (beta x, y) with the generator's BETA acts on G1 as multiplication by endo.Phi2.lam (oracle curve
    arithmetic), lam^2 + lam + 1 = 0 mod r, and every scalar splits into two parts of at most 129 bits."""
    import os
    from hekaton_system_amd.endo import phi2
    cp = CURVES[name]
    P = phi2(name)
    assert (P.lam * P.lam + P.lam + 1) % cp.r == 0 and pow(P.beta, 3, cp.q) == 1 and P.beta != 1
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "hekaton_system_amd", "csrc", "hk_tower_params.h")).read()
    pre = "HK_%s_TW" % ("BN254" if name == "bn254" else "BLS12_381")
    m = re.search(r"#define %s_BETA \{ ([^}]*) \}" % pre, txt)
    limbs = [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]
    beta_hdr = sum(l << (32 * i) for i, l in enumerate(limbs)) * pow(1 << (32 * len(limbs)), -1, cp.q) % cp.q
    assert beta_hdr == P.beta
    G1 = curve.G1(cp)
    rnd = random.Random(9)
    for _ in range(3):
        Q = G1.mul(cp.g1_gen, rnd.randrange(1, cp.r))
        want = G1.mul(Q, P.lam)
        assert (P.beta * Q[0] % cp.q, Q[1]) == (want[0], want[1])
    for c in [0, 1, cp.r - 1, P.lam, cp.r - P.lam] + [rnd.randrange(cp.r) for _ in range(500)]:
        k = P.decompose(c)
        assert (k[0] + k[1] * P.lam) % cp.r == c % cp.r and max(abs(x) for x in k).bit_length() <= 129


@pytest.mark.parametrize("name", ["bn254", "bls12_381"])
def test_device_side_split_recombines_and_is_short(name, tmp_path_factory):
    """csrc/endo.cuh `endo_decompose` (the split k_scalar_mul_endo runs per element, host-compiled here): for phi (2 parts)
    and psi (4 parts), sum_j (+-)|k_j| lambda^j = c mod r for edge and random scalars, and the parts fit the kernel's fixed
    chain lengths (131 / 68 doublings)."""
    import ctypes
    import os
    import subprocess
    import numpy as np
    from hekaton_system_amd.endo import phi2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path_factory.mktemp("shim") / "field_shim.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", out,
                           os.path.join(root, "tests", "host_shim", "field_shim.cpp")])
    lib = ctypes.CDLL(out)
    lib.shim_endo_decompose.restype = ctypes.c_uint
    cid = 0 if name == "bn254" else 1
    r = CURVES[name].r
    rnd = random.Random(12)
    for group, E, parts, bits in ((1, phi2(name), 2, 131), (2, psi4(name), 4, 68)):
        lam = E.lam
        for c in [0, 1, 2, r - 1, r - 2, lam, r - lam, (1 << 128) - 1, 1 << 253] + [rnd.randrange(r) for _ in range(400)]:
            cl = np.array([(c >> (32 * i)) & 0xffffffff for i in range(8)], np.uint32)
            mag = np.zeros(parts * 6, np.uint32)
            neg = lib.shim_endo_decompose(cid, group, cl.ctypes.data_as(ctypes.c_void_p), mag.ctypes.data_as(ctypes.c_void_p))
            ks = []
            for j in range(parts):
                m = sum(int(mag[6 * j + l]) << (32 * l) for l in range(6))
                assert m.bit_length() <= bits, (name, group, c, j, m.bit_length())
                ks.append(-m if (neg >> j) & 1 else m)
            assert sum(k * pow(lam, j, r) for j, k in enumerate(ks)) % r == c % r, (name, group, c)


@pytest.mark.parametrize("name", ["bn254", "bls12_381"])
def test_one_lane_of_the_split_kernel_multiplies_correctly(name, tmp_path_factory):
    """csrc/endo.cuh, host-compiled: the work of ONE lane of k_points_mul_split - the table 1P .. 8P, the magnitude biased by
    0x88..8, signed 4-bit digits, a Jacobian chain (jac_dbl_ni / jac_add_ni / jac_madd_ni) - equals the oracle's scalar
    multiple, in G1 (parts up to 131 bits) and G2 (68 bits), for the magnitudes at the digit boundaries (0, 7, 8, 9, 15, 16,
    0x88..8, all-ones) and random ones; infinity in, infinity out."""
    import ctypes
    import os
    import subprocess
    import numpy as np
    from oracle.pyref.codec import Codec
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path_factory.mktemp("shim") / "field_shim.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", out,
                           os.path.join(root, "tests", "host_shim", "field_shim.cpp")])
    lib = ctypes.CDLL(out)
    cp = CURVES[name]
    cd = Codec(cp)
    rnd = random.Random(31)
    for group, G, gen, enc, dec, bits in ((1, curve.G1(cp), cp.g1_gen, cd.g1_vec, cd.g1_from, 131),
                                          (2, curve.G2(cp), cp.g2_gen, cd.g2_vec, cd.g2_from, 68)):
        gid = (0 if name == "bn254" else 2) + (group - 1)
        P = G.mul(gen, rnd.randrange(1, cp.r))
        pb = bytes(enc([P]))
        mags = [0, 1, 2, 7, 8, 9, 15, 16, 17, 0x88, 0x8888888888888888, (1 << 64) - 1, (1 << bits) - 1, 1 << (bits - 1)] + \
               [rnd.randrange(1 << bits) for _ in range(12)]
        for m in mags:
            ml = np.array([(m >> (32 * i)) & 0xffffffff for i in range(6)], np.uint32)
            got = ctypes.create_string_buffer(len(pb))
            lib.shim_split_mul(gid, pb, ml.ctypes.data_as(ctypes.c_void_p), got)
            have = dec(np.frombuffer(got.raw, np.uint8))
            want = G.mul(P, m) if m % cp.r else None
            assert (have is None and want is None) or have == (want[0], want[1]), (name, group, hex(m))
        zero = ctypes.create_string_buffer(len(pb))
        lib.shim_split_mul(gid, bytes(len(pb)), np.array([5, 0, 0, 0, 0, 0], np.uint32).ctypes.data_as(ctypes.c_void_p), zero)
        assert zero.raw == bytes(len(pb))
