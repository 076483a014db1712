#!/usr/bin/env python3
"""One multi-pairing (n pairs) and one 4 x 4 batch, for `rocprofv3 --kernel-trace --stats -- python3 tools/pair_profile.py [curve] [n]`."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hekaton_system_amd import capi  # noqa: E402
from hekaton_system_amd.cp_groth16 import CURVE_PARAMS, FrCodec  # noqa: E402

curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = capi.Context(curve, 0)
fc = FrCodec(curve)
p = CURVE_PARAMS[curve]
rnd = random.Random(3)
v1 = [ctx.fixed_base(1, fc.g1(p["g1"]), fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)])) for _ in range(4)]
v2 = [ctx.fixed_base(2, fc.g2(p["g2"]), fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)])) for _ in range(4)]
for _ in range(3):
    ctx.multi_pairing(v1[0], v2[0], n=n)
for _ in range(3):
    ctx.pairing_products(v1, v2, n=n)
print("done")
