#!/usr/bin/env python3
"""Whole-call rate of dense MSMs over resident bases (hk_msm_bases), G1 and G2, 2^17..2^21 uniform scalars:
separates the accumulate kernel's efficiency at full occupancy from the small-launch effects seen inside hk_prove."""
import sys, time, random
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hekaton_system_amd import capi
from hekaton_system_amd.cp_groth16 import CURVE_PARAMS, FrCodec
ctx = capi.Context("bn254", 0)
fc = FrCodec("bn254"); p = CURVE_PARAMS["bn254"]
rng = np.random.default_rng(1)
for group, logn in ((2, 20), (2, 17), (1, 21), (1, 18)):
    n = 1 << logn
    gen = fc.g1(p["g1"]) if group == 1 else fc.g2(p["g2"])
    pb = ctx.g1_bytes if group == 1 else ctx.g2_bytes
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); ks[:, 31] &= 0x0f
    bases = capi.DeviceBuffer(ctx, n * pb)
    ctx.fixed_base(group, gen, ks.ravel(), out=bases)
    rb = ctx.bases_upload(group, bases, n=n)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x0f
    scd = capi.DeviceBuffer.from_host(ctx, sc.ravel())
    rb.msm(scd, n_scalars=n)
    t0 = time.time()
    for _ in range(3): rb.msm(scd, n_scalars=n)
    dt = (time.time() - t0) / 3
    print("G%d n=2^%d: %.2f ms per MSM, %.2f G adds/s over the whole call" % (group, logn, dt * 1e3, n * 16 / dt / 1e9))
    rb.free(); bases.free(); scd.free()
