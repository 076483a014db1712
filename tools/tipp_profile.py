#!/usr/bin/env python3
"""cProfile of one TIPP prove (n = 1024, BN254) on the GPU: where the host time between the device calls goes."""
import cProfile
import os
import pstats
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hekaton_system_amd import capi, tipa  # noqa: E402
from hekaton_system_amd.cp_groth16 import CURVE_PARAMS, FrCodec  # noqa: E402

curve = "bn254"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = capi.Context(curve, 0)
fc = FrCodec(curve)
p = CURVE_PARAMS[curve]
rnd = random.Random(1)
srs = tipa.setup(ctx, curve, n, rnd.randrange(2, p["r"]), rnd.randrange(2, p["r"]))
A = ctx.fixed_base(1, fc.g1(p["g1"]), fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)]))
B = ctx.fixed_base(2, fc.g2(p["g2"]), fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)]))
T = tipa.Tipp(ctx, curve)
com = T.com.commit_with_ip(srs.ck, A, B)
twist = rnd.randrange(2, p["r"])
tw = [pow(twist, i, p["r"]) for i in range(n)]
z = T.F.decode(ctx.multi_pairing(ctx.scalar_pairing(1, A, fc.enc(tw), n=n), B, n=n))
T.prove(srs, A, B, twist, com, z)
t0 = time.time()
T.prove(srs, A, B, twist, com, z)
print("prove %.1f ms" % ((time.time() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
T.prove(srs, A, B, twist, com, z)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
