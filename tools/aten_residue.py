#!/usr/bin/env python3
"""Which ATen kernels are left in the training step, by launch size: reads a rocprofv3 --kernel-trace CSV of bench.py and
prints, per ATen kernel family, the launch count and time per step bucketed by grid size (elements = grid for the 4-wide
vectorised kernels x 4).   python tools/aten_residue.py <dir with *_kernel_trace.csv> <steps in the trace>"""
import csv, glob, sys, collections
d, steps = sys.argv[1], float(sys.argv[2])
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "at::" not in k and "at_cuda" not in k:
            continue
        name = "add" if "CUDAFunctor_add" in k else "fill" if "FillFunctor" in k else "mul" if "MulFunctor" in k else k[:60]
        key = (name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))
        agg[key][0] += 1
        agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = collections.defaultdict(float)
for (n, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot[n] += t / steps
    print("%-40s grid %10d  launches/step %6.1f  ms/step %7.3f" % (n, g, c / steps, t / steps))
print({k: round(v, 3) for k, v in tot.items()})
