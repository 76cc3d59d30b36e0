#!/usr/bin/env python3
"""Timeline summary of one training step from a rocprofv3 kernel trace (CSV):
how much of the step has a matrix-core convolution kernel running, how much only other kernels, how much nothing, and
which non-MFMA kernels are exposed (running while no MFMA kernel is).

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 3 --warmup 2 ...
    python tools/timeline.py DIR/t_kernel_trace.csv
"""
import collections
import csv
import re
import sys

MFMA = ("k_conv_mfma", "k_conv_wgrad", "k_conv_halo")


def short(n):
    m = re.search(r"(k_\w+|vqw_\w+)", n)
    return m.group(1) if m else n.split("(")[0][-50:]


def main(path):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    adam = [i for i, e in enumerate(ev) if "adam" in e[2].lower()]
    groups = []
    for i in adam:
        if groups and ev[i][0] - ev[groups[-1][-1]][1] < 5e6:
            groups[-1].append(i)
        else:
            groups.append([i])
    t0, t1 = ev[groups[-2][-1]][1], ev[groups[-1][-1]][1]      # last full step: optimiser end to optimiser end
    win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    ismfma = lambda n: any(k in n for k in MFMA)  # noqa: E731
    pts = []
    for idx, (s, e, n) in enumerate(win):
        pts.append((s, 1, idx))
        pts.append((e, -1, idx))
    pts.sort()
    cur, last = set(), t0
    tm = to = ti = 0
    exposed = collections.defaultdict(float)
    for t, d, idx in pts:
        dt = t - last
        if any(ismfma(win[i][2]) for i in cur):
            tm += dt
        elif cur:
            to += dt
            for i in cur:
                exposed[short(win[i][2])] += dt / len(cur)
        else:
            ti += dt
        last = t
        (cur.add if d > 0 else cur.discard)(idx)
    ti += t1 - last
    tot = t1 - t0
    print("step %.2f ms: MFMA conv kernel running %.1f%%, only other kernels %.1f%% (%.2f ms), idle %.1f%% (%.2f ms)"
          % (tot / 1e6, 100 * tm / tot, 100 * to / tot, to / 1e6, 100 * ti / tot, ti / 1e6))
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in win:
        agg[short(n)][0] += 1
        agg[short(n)][1] += e - s
    print("\nkernel                                              launches   sum ms   exposed ms")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%-50s %8d %8.2f %10.2f" % (k, c, t / 1e6, exposed.get(k, 0.0) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
