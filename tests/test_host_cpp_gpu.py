"""GPU: the C++ host mirror of the cp-groth16 surface (hekaton_system_amd/csrc/host/cp_groth16.hpp),
compiled with g++ against libhekaton.so, reproduces a golden commit+prove byte for byte; its wire codec
(host/ark_serialize.hpp: ark-serialize framing, ChaCha12Rng, Fr::rand) agrees with the Python mirror's bytes."""
import os
import subprocess

import numpy as np
import pytest

from tests import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror_commit_prove(tmp_path):
    case = gu.load("groth16.json")["bn254"][0]
    d = tmp_path / "case"
    d.mkdir()
    pk = case["pk"]
    for name in ("a_g", "b_g", "b_h", "h_g", "deltas_g", "last_delta_h", "alpha_g", "beta_g", "beta_h"):
        gu.hb(pk[name]).tofile(d / name)
    gu.hb(pk["ck"][0]).tofile(d / "ck0")
    gu.hb(pk["ck"][1]).tofile(d / "ck1")
    for m in "ABC":
        rp, col, val = gu.csr(case[m])
        rp.tofile(d / (m + "_row_ptr")); col.tofile(d / (m + "_col")); val.tofile(d / (m + "_val"))
    s0, e0 = case["stage_ranges"][0]
    np.array([case["n_inst"], case["n_constraints"], e0 - s0], dtype=np.uint64).tofile(d / "dims")
    gu.hb(case["z_mont"]).tofile(d / "z")
    gu.hb(case["kappas_mont"]).tofile(d / "kappa")
    gu.hb(case["r_mont"]).tofile(d / "r")
    gu.hb(case["s_mont"]).tofile(d / "s")
    gu.hb(case["comms"][0]).tofile(d / "expect_com")
    for k in "abc":
        gu.hb(case["proof"][k]).tofile(d / ("expect_" + k))
    # the wire bytes the Python mirror produces for the same records (big-int codec, no device)
    from hekaton_system_amd.ark_serialize import ArkCodec, commitment_randomness
    from hekaton_system_amd.cp_groth16 import Proof
    from hekaton_system_amd.worker import Stage0Response, Stage1Response
    cdx = ArkCodec("bn254")
    seed = bytes(range(100, 132))
    np.frombuffer(seed, dtype=np.uint8).tofile(d / "seed")
    com = gu.hb(case["comms"][0])
    proof = Proof(gu.hb(case["proof"]["a"]), gu.hb(case["proof"]["b"]), gu.hb(case["proof"]["c"]), [com])
    np.frombuffer(cdx.stage0_response_to_wire(Stage0Response(7, com, seed)), dtype=np.uint8).tofile(d / "expect_wire0")
    np.frombuffer(cdx.stage1_response_to_wire(Stage1Response(7, proof)), dtype=np.uint8).tofile(d / "expect_wire1")
    np.frombuffer(cdx.points_to_wire(1, proof.a, True), dtype=np.uint8).tofile(d / "expect_a_compressed")
    np.frombuffer(cdx.points_to_wire(2, proof.b, True), dtype=np.uint8).tofile(d / "expect_b_compressed")
    commitment_randomness("bn254", seed).tofile(d / "expect_kappa_from_seed")
    exe = str(tmp_path / "test_host_mirror")
    libdir = os.path.join(ROOT, "hekaton_system_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe,
                           os.path.join(ROOT, "tests", "host_cpp", "test_host_mirror.cpp"),
                           "-L" + libdir, "-lhekaton", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "HOST_MIRROR_OK" in out.stdout
