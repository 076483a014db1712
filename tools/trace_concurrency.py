#!/usr/bin/env python3
"""Concurrency analysis of a rocprofv3 --kernel-trace CSV of bench.py: how much of the timed region has no
throughput-bound ("big") kernel in flight, how many run at once, and the summed duration per kernel per proof.
usage: trace_concurrency.py <kernel_trace.csv> [proofs_to_skip_as_warmup]"""
import collections
import csv
import re
import sys

BIG = {'k_msm_accum0', 'k_msm_accum0<G2>', 'k_ntt_pass4', 'k_msm_scatter', 'k_msm_hist', 'k_spmv', 'k_mul_pointwise'}


def short(n):
    m = re.match(r"void hk::(\w+)<(.*)", n)
    if not m:
        return n[:30]
    k, rest = m.group(1), m.group(2)
    if k.startswith('k_msm') and rest.startswith('hk::Fp2'):
        k += '<G2>'
    return k


def main(path, skip):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
    fin = [e for e in ev if e[2] == 'k_finish']
    t0, t1 = fin[skip - 1][1], fin[-1][1]
    n = len(fin) - skip
    W = t1 - t0
    print("window %.1f ms, %d proofs, %.2f ms/proof" % (W / 1e6, n, W / 1e6 / n))
    pts = []
    for s, e, k in ev:
        if e < t0 or s > t1:
            continue
        pts.append((max(s, t0), 1, k))
        pts.append((min(e, t1), -1, k))
    pts.sort()
    cur, last = collections.Counter(), t0
    tot, hist, combo = collections.Counter(), collections.Counter(), collections.Counter()
    for t, d, k in pts:
        dt = t - last
        if dt > 0:
            nb = sum(v for kk, v in cur.items() if kk in BIG)
            hist[min(nb, 6)] += dt
            tot['idle' if not sum(cur.values()) else ('only_small' if nb == 0 else 'big')] += dt
            combo[tuple(sorted((kk, v) for kk, v in cur.items() if kk in BIG and v > 0))] += dt
        cur[k] += d
        last = t
    print({k: round(v / W * 100, 1) for k, v in tot.items()})
    print("big kernels in flight, % of time:", {k: round(v / W * 100, 1) for k, v in sorted(hist.items())})
    for key, v in combo.most_common(10):
        print("  %5.1f%%  %s" % (v / W * 100, key))
    dur, cnt = collections.Counter(), collections.Counter()
    for s, e, k in ev:
        if s >= t0 and e <= t1:
            dur[k] += e - s
            cnt[k] += 1
    for k, v in dur.most_common(16):
        print("%-28s %6.2f ms/proof  n/proof=%.1f avg=%.1f us" % (k, v / 1e6 / n, cnt[k] / n, v / cnt[k] / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
