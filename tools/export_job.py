#!/usr/bin/env python3
"""Writes one proving job as the raw C-ABI arrays `apps/hk_all_in_one` reads (needs the GPU: the key points come from
hk_fixed_base).  Layout under <job_dir>:

    job                       u64[2]   n_subcircuits, n_classes
    subs                      u64[2 n] (class position, assignment index) per subcircuit
    com_seeds                 32 n     com_seed per subcircuit (worker.rs:129)
    rs                        64 n     the prover's blinders r, s (Montgomery Fr) per subcircuit
    class_<c>/                a_g b_g b_h h_g deltas_g last_delta_h alpha_g beta_g beta_h ck0 ck1   packed affine points
                              {A,B,C}_{row_ptr,col,val}                                              CSR matrices
                              dims = u64[4] n_inst, n_constraints, n0, n_assignments;  z_<k>          full assignments

usage: export_job.py <job_dir> [--config tiny] [--subcircuits 8] [--witnesses 2] [--single-class]"""
import argparse
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def export(job_dir, config="tiny", n=8, witnesses=2, single_class=False, ctx=None, curve="bn254"):
    from hekaton_system_amd import capi
    from hekaton_system_amd.cp_groth16 import FrCodec, setup_device
    from hekaton_system_amd.workload import config_classes, prepare_class_host, representative_subcircuit
    own = ctx is None
    ctx = ctx or capi.Context(curve, 0)
    fc = FrCodec(curve)
    family, _n, _reps = config_classes(config)
    cls_of = [1 if single_class else representative_subcircuit(family, n, i) for i in range(n)]
    reps = sorted(set(cls_of))
    os.makedirs(job_dir, exist_ok=True)
    np.array([n, len(reps)], np.uint64).tofile(os.path.join(job_dir, "job"))
    n_assign = {}
    for pos, rep in enumerate(reps):
        members = [i for i in range(n) if cls_of[i] == rep]
        k = min(witnesses, len(members))
        seed = hashlib.sha256(b"HEKATON1 class %d" % rep).digest()
        _rep, hs, assigns = prepare_class_host((curve, config, rep, seed, [1000 * rep + j + 1 for j in range(k)], n))
        pk, _td = setup_device(hs, ctx, keep_on_device=False)
        d = os.path.join(job_dir, "class_%d" % pos)
        os.makedirs(d, exist_ok=True)
        for name, arr in (("a_g", pk.a_g), ("b_g", pk.b_g), ("b_h", pk.b_h), ("h_g", pk.h_g), ("deltas_g", pk.deltas_g),
                          ("last_delta_h", pk.vk.last_delta_h), ("alpha_g", pk.vk.alpha_g), ("beta_g", pk.beta_g),
                          ("beta_h", pk.vk.beta_h), ("ck0", pk.ck.deltas_abc_g[0]), ("ck1", pk.ck.deltas_abc_g[1])):
            np.asarray(arr, np.uint8).tofile(os.path.join(d, name))
        for m, (rp, col, val) in zip("ABC", pk.matrices):
            np.asarray(rp, np.uint64).tofile(os.path.join(d, m + "_row_ptr"))
            np.asarray(col, np.uint32).tofile(os.path.join(d, m + "_col"))
            np.asarray(val, np.uint8).tofile(os.path.join(d, m + "_val"))
        n0 = len(pk.ck.deltas_abc_g[0]) // ctx.g1_bytes
        np.array([pk.n_inst, pk.n_constraints, n0, k], np.uint64).tofile(os.path.join(d, "dims"))
        for j, (_ws, zb, _w0) in enumerate(assigns):
            np.asarray(zb, np.uint8).tofile(os.path.join(d, "z_%d" % j))
        n_assign[rep] = k
    subs, seeds, rs = [], b"", b""
    count = {}
    for i in range(n):
        rep = cls_of[i]
        subs += [reps.index(rep), count.get(rep, 0) % n_assign[rep]]
        count[rep] = count.get(rep, 0) + 1
        seeds += hashlib.sha256(b"com_seed %d" % i).digest()
        for tag in (b"r %d", b"s %d"):
            rs += fc.enc1(int.from_bytes(hashlib.sha256(tag % i).digest(), "little") % fc.r).tobytes()
    np.array(subs, np.uint64).tofile(os.path.join(job_dir, "subs"))
    with open(os.path.join(job_dir, "com_seeds"), "wb") as f:
        f.write(seeds)
    with open(os.path.join(job_dir, "rs"), "wb") as f:
        f.write(rs)
    if own:
        ctx.close()
    return reps, cls_of


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("job_dir")
    ap.add_argument("--config", default="tiny")
    ap.add_argument("--subcircuits", type=int, default=8)
    ap.add_argument("--witnesses", type=int, default=2)
    ap.add_argument("--single-class", action="store_true")
    ap.add_argument("--curve", default="bn254")
    a = ap.parse_args()
    reps, _ = export(a.job_dir, a.config, a.subcircuits, a.witnesses, a.single_class, curve=a.curve)
    print("exported %d subcircuits, %d classes to %s" % (a.subcircuits, len(reps), a.job_dir))
