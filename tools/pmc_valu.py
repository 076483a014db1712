#!/usr/bin/env python3
"""Per-kernel VALU instruction budget of one proof from a rocprofv3 counter_collection.csv
(`--pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY` on
`HK_SERIAL_STREAMS=1 bench.py --single-class --steps 1 --warmup 0 --subcircuits N --threads 1 --no-verify`).
usage: pmc_valu.py <counter_collection.csv> <n_proofs>"""
import csv, re, sys
from collections import defaultdict

n = int(sys.argv[2])
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void hk::", ""))[:58]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
tot = sum(v["SQ_INSTS_VALU"] for v in acc.values())
print("%-60s %7s %14s %7s %9s %9s %9s" % ("kernel", "calls/p", "VALU inst/proof", "share", "active", "wait_inst", "wait_any"))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:24]:
    wc = v["SQ_WAVE_CYCLES"] or 1
    print("%-60s %7.1f %14.3e %6.1f%% %9.3f %9.3f %9.3f" % (k, len(calls[k]) / n, v["SQ_INSTS_VALU"] / n, 100 * v["SQ_INSTS_VALU"] / tot,
                                                      v["SQ_ACTIVE_INST_VALU"] / wc, v["SQ_WAIT_INST_ANY"] / wc, v["SQ_WAIT_ANY"] / wc))
print("total wave-level VALU instructions per proof: %.3e" % (tot / n))
