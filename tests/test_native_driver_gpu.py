"""GPU: the native job driver (apps/hk_all_in_one.cpp: C++ host mirror + threads over the C ABI, both curves, the shape of
mpi-snark/src/bin/all_in_one.rs:109-196) on an 8-subcircuit, 5-class job: every Stage0Response / Stage1Response file it
writes equals, byte for byte, the ark-serialize bytes of the Python worker path for the same subcircuit (same key, same
assignment, kappa from the same com_seed, same r and s) - two independent host stacks over one library."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_native_driver_writes_the_python_worker_bytes(cname, tmp_path, ctx_bn254, ctx_bls):
    from hekaton_system_amd.ark_serialize import ArkCodec
    from hekaton_system_amd.chacha import ChaCha12Rng
    from hekaton_system_amd.cp_groth16 import FrCodec, Proof
    from hekaton_system_amd.worker import Stage0Response, Stage1Response
    from tools.export_job import export
    exe = os.path.join(ROOT, "apps", "hk_all_in_one")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps")], stdout=subprocess.DEVNULL)
    job, out = str(tmp_path / "job"), str(tmp_path / "out")
    os.makedirs(out)
    n = 8
    ctx0 = ctx_bn254 if cname == "bn254" else ctx_bls
    reps, cls_of = export(job, "tiny", n, witnesses=2, ctx=ctx0, curve=cname)
    assert len(reps) == 5
    res = subprocess.run([exe, job, out, "--threads", "4", "--steps", "2", "--warmup", "1", "--curve", cname],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["subcircuits"] == n and line["classes"] == 5 and line["proofs_per_s"] > 0 and line["curve"] == cname
    sizes = {"bn254": (104, 336), "bls12_381": (136, 496)}[cname]
    assert (os.path.getsize(os.path.join(out, "stage0_resp_0.bin")), os.path.getsize(os.path.join(out, "stage1_resp_0.bin"))) == sizes
    # the same job through the Python host stack
    ctx, fc, codec = ctx0, FrCodec(cname), ArkCodec(cname)
    frb = ctx.fr_bytes
    rd = lambda *p: np.fromfile(os.path.join(job, *p), np.uint8)
    subs = np.fromfile(os.path.join(job, "subs"), np.uint64).reshape(n, 2)
    keys = {}
    for i in range(n):
        pos, k = int(subs[i, 0]), int(subs[i, 1])
        d = "class_%d" % pos
        if pos not in keys:
            dims = np.fromfile(os.path.join(job, d, "dims"), np.uint64)
            mats = tuple((np.fromfile(os.path.join(job, d, m + "_row_ptr"), np.uint64),
                          np.fromfile(os.path.join(job, d, m + "_col"), np.uint32), rd(d, m + "_val")) for m in "ABC")
            keys[pos] = (ctx.pk_upload(a_g=rd(d, "a_g"), b_g=rd(d, "b_g"), b_h=rd(d, "b_h"), h_g=rd(d, "h_g"),
                                       ck_stages=[rd(d, "ck0"), rd(d, "ck1")], deltas_g=rd(d, "deltas_g"),
                                       last_delta_h=rd(d, "last_delta_h"), alpha_g=rd(d, "alpha_g"), beta_g=rd(d, "beta_g"),
                                       beta_h=rd(d, "beta_h"), matrices=mats, n_inst=int(dims[0]), n_constraints=int(dims[1])),
                         int(dims[0]), int(dims[2]))
        dpk, n_inst, n0 = keys[pos]
        z = rd(d, "z_%d" % k)
        seed = hashlib.sha256(b"com_seed %d" % i).digest()
        kappa = fc.enc1(ChaCha12Rng(seed).fr(fc.r))
        r_, s_ = (fc.enc1(int.from_bytes(hashlib.sha256(t % i).digest(), "little") % fc.r) for t in (b"r %d", b"s %d"))
        com = dpk.commit(0, z[n_inst * frb:(n_inst + n0) * frb], kappa)
        a, b, c = dpk.prove(z, r_, s_, kappa)
        want0 = codec.stage0_response_to_wire(Stage0Response(i, com, seed))
        want1 = codec.stage1_response_to_wire(Stage1Response(i, Proof(a, b, c, [com])))
        assert open(os.path.join(out, "stage0_resp_%d.bin" % i), "rb").read() == bytes(want0), i
        assert open(os.path.join(out, "stage1_resp_%d.bin" % i), "rb").read() == bytes(want1), i
    for dpk, _a, _b in keys.values():
        dpk.free()
    # a job directory that lies about its size is refused, not indexed out of range
    np.array([n + 1, 5], np.uint64).tofile(os.path.join(job, "job"))
    bad = subprocess.run([exe, job, out, "--curve", cname], capture_output=True, text=True, timeout=300)
    assert bad.returncode == 4 and "do not match" in bad.stderr
