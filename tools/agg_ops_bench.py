#!/usr/bin/env python3
"""Latency of the aggregation-sized primitives: the KZG-opening MSMs of TIPA
(`G::Group::msm(&srs_powers, &coeffs)`, distributed-prover/src/kzg.rs:151-152 — N and 2N terms for N subcircuits)
through hk_msm_g1 / hk_msm_g2 over caller-supplied bases, and `scalar_pairing` (pairing_ops.rs:32-39) through
hk_scalar_pairing_g1 / g2; `resident` = the same MSM through hk_bases_upload / hk_msm_bases (bases uploaded once with
their shift tables, as a static SRS would be).  Inputs resident in HBM; the CPU column is the oracle's ark-style Pippenger on this
box's host threads (test infrastructure, timed here only as the baseline beside it)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from hekaton_system_amd import capi  # noqa: E402
from hekaton_system_amd.cp_groth16 import CURVE_PARAMS, FrCodec  # noqa: E402


def main():
    curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
    ctx = capi.Context(curve, 0)
    fc = FrCodec(curve)
    p = CURVE_PARAMS[curve]
    try:
        from oracle.c_oracle import COracle
        co = COracle(curve)
    except Exception:       # noqa: BLE001
        co = None
    rnd = random.Random(1)
    print("%-8s %8s %12s %14s %12s %14s" % ("group", "n", "msm ms", "resident ms", "cpu msm ms", "scalar_pair ms"))
    for group in (1, 2):
        gen = fc.g1(p["g1"]) if group == 1 else fc.g2(p["g2"])
        pb = ctx.g1_bytes if group == 1 else ctx.g2_bytes
        for n in (64, 256, 1024, 2048, 8192, 65536):
            ks = fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)])
            bases = capi.DeviceBuffer(ctx, n * pb)
            ctx.fixed_base(group, gen, ks, out=bases)
            scal_h = fc.enc([rnd.randrange(p["r"]) for _ in range(n)])
            scal = capi.DeviceBuffer.from_host(ctx, scal_h)
            msm = ctx.msm_g1 if group == 1 else ctx.msm_g2
            msm(bases, scal, n_bases=n, n_scalars=n)
            reps = 20 if n <= 8192 else 5
            t0 = time.time()
            for _ in range(reps):
                msm(bases, scal, n_bases=n, n_scalars=n)
            t_msm = (time.time() - t0) / reps
            rb = ctx.bases_upload(group, bases, n=n)
            rb.msm(scal, n_scalars=n)
            t0 = time.time()
            for _ in range(reps):
                rb.msm(scal, n_scalars=n)
            t_res = (time.time() - t0) / reps
            rb.free()
            ctx.scalar_pairing(group, bases, scal, n=n)
            t0 = time.time()
            for _ in range(reps):
                ctx.scalar_pairing(group, bases, scal, n=n)
            t_sp = (time.time() - t0) / reps
            t_cpu = float("nan")
            if co is not None and n <= 8192:
                bh = bases.to_host()
                t0 = time.time()
                co.msm(group, bh, scal_h)
                t_cpu = time.time() - t0
            print("%-8s %8d %12.3f %14.3f %12.3f %14.3f" % ("G%d" % group, n, t_msm * 1e3, t_res * 1e3, t_cpu * 1e3, t_sp * 1e3))
            bases.free()
            scal.free()


if __name__ == "__main__":
    main()
