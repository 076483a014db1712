#!/usr/bin/env python3
"""Kernel timeline of ONE TIPP prove (tipa.Tipp.prove), to see where a round's wall-clock goes.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 tools/tipp_timeline.py run bn254 64
    python3 tools/tipp_timeline.py show DIR/.../run_kernel_trace.csv

`run` proves twice with a 0.3 s pause in between; `show` takes the dispatches after the longest gap of the trace (the
second prove) and prints, per kernel, its start relative to the first one, its duration and its queue, then per round the
busy time of the critical queue."""
import csv
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(curve, n):
    from hekaton_system_amd import capi, tipa
    from hekaton_system_amd.cp_groth16 import CURVE_PARAMS, FrCodec
    ctx = capi.Context(curve, 0)
    fc, p = FrCodec(curve), CURVE_PARAMS[curve]
    rnd = random.Random(3)
    srs = tipa.setup(ctx, curve, n, rnd.randrange(2, p["r"]), rnd.randrange(2, p["r"]))
    A = ctx.fixed_base(1, fc.g1(p["g1"]), fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)]))
    B = ctx.fixed_base(2, fc.g2(p["g2"]), fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)]))
    T = tipa.Tipp(ctx, curve)
    com = T.com.commit_with_ip(srs.ck, A, B)
    twist = rnd.randrange(2, p["r"])
    tw = [pow(twist, i, p["r"]) for i in range(n)]
    z = T.F.decode(ctx.multi_pairing(ctx.scalar_pairing(1, A, fc.enc(tw), n=n), B, n=n))
    T.prove(srs, A, B, twist, com, z)
    time.sleep(0.3)
    t0 = time.time()
    T.prove(srs, A, B, twist, com, z)
    print("prove %.1f ms" % ((time.time() - t0) * 1e3))


def run_setup(curve, n):
    """The same for `tipa.setup` (TIPA::setup(N), inside the job's wall-clock in the reference): twice, the second one shown."""
    from hekaton_system_amd import capi, tipa
    from hekaton_system_amd.cp_groth16 import CURVE_PARAMS
    ctx = capi.Context(curve, 0)
    p = CURVE_PARAMS[curve]
    rnd = random.Random(3)
    for _ in range(2):
        srs = tipa.setup(ctx, curve, n, rnd.randrange(2, p["r"]), rnd.randrange(2, p["r"]))
        for rb in srs.resident.values():
            rb.free()
    time.sleep(0.3)
    t0 = time.time()
    tipa.setup(ctx, curve, n, rnd.randrange(2, p["r"]), rnd.randrange(2, p["r"]))
    print("setup %.1f ms" % ((time.time() - t0) * 1e3))


def show(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    gap_at, gap = 0, 0
    for k in range(1, len(rows)):
        g = int(rows[k]["Start_Timestamp"]) - int(rows[k - 1]["End_Timestamp"])
        if g > gap:
            gap_at, gap = k, g
    rows = rows[gap_at:]
    t0 = int(rows[0]["Start_Timestamp"])
    busy = 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].split("(")[0].replace("void hk::", "")[:70]
        print("%9.3f ms  +%8.3f ms  q%-3s grid %-8s %s" % (s / 1e6, (e - s) / 1e6, r.get("Queue_Id", "?"), r.get("Grid_Size", "?"), name))
        busy += e - s
    print("span %.3f ms, sum of kernel durations %.3f ms, %d dispatches" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e6, busy / 1e6, len(rows)))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "setup":
        run_setup(sys.argv[2], int(sys.argv[3]))
    else:
        show(sys.argv[2])
