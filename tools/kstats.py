#!/usr/bin/env python3
"""Short per-kernel table of a rocprofv3 --stats CSV:  python tools/kstats.py t_kernel_stats.csv [min_us]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
for r in rows:
    n = r["Name"]
    m = re.search(r"(k_\w+(<[^>]*>)?|direct_copy\w*|CUDAFunctor\w*|rocclr\w*|normal_and\w*)", n)
    avg = float(r["AverageNs"]) / 1e3
    if avg >= lo:
        print("%-46s calls %5s  avg %8.1f us  total %8.2f ms" % ((m.group(1) if m else n[:46])[:46], r["Calls"], avg, float(r["TotalDurationNs"]) / 1e6))
