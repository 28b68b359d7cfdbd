#!/usr/bin/env python
"""Per-kernel durations and gaps of the steady-state launch chain from a rocprofv3 --kernel-trace CSV.
usage: python tools/chain_gaps.py <kernel_trace.csv> [first_kernel_name_substring]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "te_len"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# steady state: the last complete chains
chains = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][-200:]
dur = defaultdict(list); gap = defaultdict(list); tot = []
for a, b in chains:
    t0 = int(rows[a]["Start_Timestamp"])
    tot.append(int(rows[b]["Start_Timestamp"]) - t0)
    prev_end = None
    for k in range(a, b):
        r = rows[k]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0][:60]
        dur[(k - a, name)].append(e - s)
        if prev_end is not None:
            gap[(k - a, name)].append(s - prev_end)
        prev_end = max(prev_end or 0, e)
print("chains: %d, mean period %.1f us" % (len(chains), sum(tot) / len(tot) / 1e3))
sd = sg = 0
for key in sorted(dur):
    d = sum(dur[key]) / len(dur[key]) / 1e3
    g = sum(gap[key]) / len(gap[key]) / 1e3 if gap.get(key) else 0.0
    sd += d; sg += g
    print("%3d %-62s dur %7.1f us  gap before %6.1f us" % (key[0], key[1], d, g))
print("sum of durations %.1f us, sum of gaps %.1f us" % (sd, sg))
