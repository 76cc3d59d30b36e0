#!/usr/bin/env python3
"""Per-kernel time per step of two rocprofv3 --stats runs of bench.py (plain vs forced one-rank RCCL), largest differences first.
    python tools/dp_probe/diff_stats.py <plain dir> <forced dir> <steps>"""
import csv, glob, sys
def load(d, steps):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps) for r in csv.DictReader(open(f))}
a, b, steps = load(sys.argv[1], float(sys.argv[3])), load(sys.argv[2], float(sys.argv[3])), float(sys.argv[3])
rows = []
for k in set(a) | set(b):
    ta, ca = a.get(k, (0, 0)); tb, cb = b.get(k, (0, 0))
    rows.append((tb - ta, k[:80], ta, tb, ca, cb))
rows.sort(reverse=True)
print("sum plain %.1f ms/step, forced %.1f ms/step" % (sum(v[0] for v in a.values()), sum(v[0] for v in b.values())))
for r in rows[:14] + rows[-5:]:
    print("%+8.3f ms  %-80s plain %7.3f (%5.1f calls)  forced %7.3f (%5.1f)" % (r[0], r[1], r[2], r[4], r[3], r[5]))
