#!/usr/bin/env python3
"""Sums rocprofv3 counter_collection.csv per kernel: usage pmc_agg.py <counter_collection.csv>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
def short(n):
    m = re.match(r"void hk::(\w+)<(.*)", n)
    if not m: return n[:30]
    k, rest = m.group(1), m.group(2)
    return k + ('<G2>' if rest.startswith('hk::Fp2') else '')
names = set()
for r in rows:
    k = short(r['Kernel_Name']); agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id']); names.add(r['Counter_Name'])
names = sorted(names)
print("%-26s %4s " % ("kernel", "n") + " ".join("%22s" % n for n in names))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    print("%-26s %4d " % (k, len(disp[k])) + " ".join("%22.4g" % (v[n] / len(disp[k])) for n in names))
