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
            "*_H_launch = the dense H-query launch.  Calibration for THIS access pattern (MI355X_MICROARCH.md, HBM: only wide "
            "coalesced streams are known to read 1/2): the H launch REQUESTS (m-1) x 16 rows x 64 B + the 4-byte entries = "
            "2.28 GB and the counter reads 2.57 GB, i.e. 1.13 x the requested bytes - 64-byte row gathers are tallied at "
            "their size, so the raw values are quoted with no x2.",
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
