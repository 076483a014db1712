#!/usr/bin/env python3
"""a pmc json from two rocprofv3 counter_collection.csv files (separate --pmc FETCH_SIZE and
--pmc WRITE_SIZE passes of `HK_SERIAL_STREAMS=1 bench.py --steps 1 --warmup 0 --subcircuits 2 --threads 1`).
usage: pmc_traffic.py <fetch.csv> <write.csv> <out.json>"""
import csv, json, sys


def per_dispatch(path, counter):
    d = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "k_msm_accum0<hk::Fp<" in r["Kernel_Name"]:
            d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return list(d.values())


f = per_dispatch(sys.argv[1], "FETCH_SIZE")
w = per_dispatch(sys.argv[2], "WRITE_SIZE")
out = {
    "kernel": "k_msm_accum0<Fp<Bn254FqP>> (BN254, big-merkle-64x32)",
    "FETCH_SIZE_KiB_avg": sum(f) / len(f), "WRITE_SIZE_KiB_avg": sum(w) / len(w),
    "FETCH_SIZE_KiB_H_launch": max(f), "WRITE_SIZE_KiB_H_launch": max(w), "launches": len(f),
    "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (HK_SERIAL_STREAMS=1 bench.py --steps 1 "
            "--warmup 0 --subcircuits 2 --threads 1); *_avg = mean over all launches of the kernel (5 per subcircuit), "
            "*_H_launch = the dense H-query launch. Random 64-B gathers: the gfx950 x2 FETCH_SIZE calibration for wide "
            "coalesced streams is not established for this pattern, raw values quoted.",
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
