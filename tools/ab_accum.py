#!/usr/bin/env python3
"""A/B of one library build (HK_LIB) on the default workload: per-proof device timings of sequential hk_prove calls
(one lane), with the round-1 bench's small blinders and with full-width ones."""
import os
import sys
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hekaton_system_amd import capi  # noqa: E402
from hekaton_system_amd.cp_groth16 import FrCodec, SeededRng, generate_parameters  # noqa: E402
from hekaton_system_amd.workload import make_config  # noqa: E402

ctx = capi.Context("bn254", 0)
fc = FrCodec("bn254")
circ = make_config("bn254", "big-merkle-64x32")
pk, td = generate_parameters(circ, "bn254", SeededRng(b"HEKATON1" * 4), ctx, keep_on_device=True)
dpk = pk.upload(ctx)
circ.set_witness_seed(1)
z = capi.DeviceBuffer.from_host(ctx, circ.full_assignment_bytes())
ctx.set_profiling(True)
rnd = random.Random(1)
for label, r, s in (("small r,s", 0x1234567, 0x7654321), ("full r,s", rnd.randrange(fc.r), rnd.randrange(fc.r))):
    acc = []
    for k in range(6):
        dpk.prove(z, fc.enc1(r), fc.enc1(s), fc.enc([0x5555]), n_v=circ.n_v)
        t = ctx.last_timings()
        if k:
            acc.append(t)
    mean = lambda key: sum(t[key] for t in acc) / len(acc)
    print("%-28s %-10s total %.2f ms  accum_h %.3f  accum_all %.3f  witness_map %.2f  digits %.2f  finish %.2f" % (
        os.path.basename(capi.LIB_PATH), label, mean("total_ms"), mean("accum_h_ms"), mean("accum_kernel_ms"),
        mean("witness_map_ms"), mean("digits_ms"), mean("finish_ms")))
