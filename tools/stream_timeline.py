#!/usr/bin/env python3
"""Per-stream view of the last full training step in a rocprofv3 kernel trace (CSV with Stream_Id):
5 ms bins with the share of time a matrix-core conv kernel runs, any kernel runs, and each stream is busy;
then the longest intervals in which no MFMA kernel runs, with what runs instead.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing
    python tools/stream_timeline.py DIR/t_kernel_trace.csv [bin_ms] [steps_back] [--json OUT.json]

--json: also write {step_ms, launches_per_step, mfma_active_ms, mfma_active_fraction, exposed_ms, csrc_digest} for bench.py's
`launches_per_step` / `mfma_active_fraction` fields (profiles/r04_stream_timeline.json).

steps_back: which step to look at, counted from the end (1 = last).  A default `bench.py` run ends with three serialised
steps (the roofline pass), so its last concurrent step is steps_back = 4.
"""
import collections
import csv
import re
import sys

MF = ("k_conv_mfma", "k_conv_wgrad", "k_conv_halo", "k_conv_wino", "k_conv_dil", "k_vq_mfma")


def ism(n):
    return any(k in n for k in MF)


def short(n):
    m = re.search(r"(k_\w+|vqw_\w+)", n)
    return m.group(1) if m else n.split("(")[0][-40:]


def cover(L, a, b):
    iv = sorted((max(e[0], a), min(e[1], b)) for e in L if e[1] > a and e[0] < b)
    tot, cur = 0, None
    for s, e in iv:
        if cur is None:
            cur = [s, e]
        elif s <= cur[1]:
            cur[1] = max(cur[1], e)
        else:
            tot += cur[1] - cur[0]
            cur = [s, e]
    if cur:
        tot += cur[1] - cur[0]
    return tot


def main(path, binms=5.0, back=1, json_out=None):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]) for r in rows)
    adam = [i for i, e in enumerate(ev) if "adam" in e[2].lower()]
    groups = []
    for i in adam:
        if groups and ev[i][0] - ev[groups[-1][-1]][1] < 5e6:
            groups[-1].append(i)
        else:
            groups.append([i])
    t0, t1 = ev[groups[-back - 1][-1]][1], ev[groups[-back][-1]][1]
    win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    print("step %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(win)))
    streams = collections.defaultdict(list)
    for e in win:
        streams[e[3]].append(e)
    for s, L in sorted(streams.items()):
        print("stream %s: %d kernels, busy %.1f ms (MFMA kernels %.1f), first at %.1f, last ends %.1f" % (
            s, len(L), sum(e[1] - e[0] for e in L) / 1e6, sum(e[1] - e[0] for e in L if ism(e[2])) / 1e6,
            (L[0][0] - t0) / 1e6, (L[-1][1] - t0) / 1e6))
    mf = [e for e in win if ism(e[2])]
    keys = sorted(streams)
    print(" ms  mfma%  any% | busy% per stream")
    b = binms * 1e6
    nb = int((t1 - t0) / b) + 1
    for i in range(nb):
        a, c = t0 + i * b, min(t0 + (i + 1) * b, t1)
        if c <= a:
            break
        print("%3d  %4.0f  %4.0f | " % (i * binms, 100 * cover(mf, a, c) / (c - a), 100 * cover(win, a, c) / (c - a)) +
              " ".join("%s:%3.0f" % (k, 100 * cover(streams[k], a, c) / (c - a)) for k in keys))
    print("MFMA-active %.1f ms of %.1f" % (cover(mf, t0, t1) / 1e6, (t1 - t0) / 1e6))
    # kernels exposed (running while no MFMA kernel is)
    pts = sorted([(e[0], 1) for e in mf] + [(e[1], -1) for e in mf])
    gaps, depth, last = [], 0, t0
    for t, d in pts:
        if depth == 0 and t > last:
            gaps.append((last, t))
        depth += d
        if depth == 0:
            last = t
    if last < t1:
        gaps.append((last, t1))
    exposed = collections.Counter()
    for a, c in gaps:
        for e in win:
            if not ism(e[2]) and e[1] > a and e[0] < c:
                exposed[short(e[2])] += min(e[1], c) - max(e[0], a)
    print("exposed non-MFMA kernel time (ms):", ", ".join("%s %.2f" % (k, v / 1e6) for k, v in exposed.most_common(14)))
    if json_out:
        import json
        import os
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        try:
            import bench
            digest = bench.csrc_digest()
        except Exception:
            digest = None
        act = cover(mf, t0, t1)
        json.dump(dict(step_ms=(t1 - t0) / 1e6, launches_per_step=len(win), mfma_active_ms=act / 1e6,
                       mfma_active_fraction=act / (t1 - t0), exposed_ms=(t1 - t0 - act) / 1e6,
                       exposed_by_kernel_ms={k: v / 1e6 for k, v in exposed.most_common(20)},
                       csrc_digest=digest, trace=os.path.basename(path), steps_back=back), open(json_out, "w"), indent=1)


if __name__ == "__main__":
    av = sys.argv[1:]
    jo = None
    if "--json" in av:
        i = av.index("--json")
        jo = av[i + 1]
        del av[i:i + 2]
    main(av[0], float(av[1]) if len(av) > 1 else 5.0, int(av[2]) if len(av) > 2 else 1, jo)
