#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into per-kernel HBM
bytes per launch.  Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KiB, and on gfx950
FETCH_SIZE counts exactly half of the bytes of wide coalesced reads, so reads = 2 * FETCH_SIZE * 1024.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_hbm_traffic.json
"""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FAMILIES = {"conv_mfma_fwd_dgrad": ("k_conv_mfma_fwd", "k_conv_halo", "k_conv_dilrow<"),
            "conv_mfma_wgrad": ("k_conv_wgrad9", "k_conv_wgrad_up", "k_conv_mfma_wgrad", "k_conv_wgrad_tile", "k_conv_dilrow_wgrad"),
            "conv_winograd": ("k_conv_wino",),
            "conv_generic_fwd": ("k_conv_direct_fwd", "k_stem_fwd", "k_head_fwd"),
            "conv_generic_wgrad": ("k_conv_direct_wgrad", "k_stem_wgrad", "k_head_wgrad")}


def load(d, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])      # also inside "(anonymous namespace)::"
                k = m.group(1) if m else r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
    return agg


STEPS = 3       # the PMC passes run `bench.py --steps 2 --warmup 1 --no-kernel-timing`: three training steps in all


def main(fetch_dir, write_dir, out):
    F, Wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in F:
        n = F[k][0]
        rd = 2.0 * F[k][1] * 1024 / n
        wr = Wr[k][1] * 1024 / max(Wr[k][0], 1) if k in Wr else 0.0
        kernels[k] = dict(launches=n, read_bytes_per_launch=rd, write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr)
    fam = {}
    for name, pats in FAMILIES.items():
        sel = [v for k, v in kernels.items() if any(k.startswith(p) for p in pats)]
        n = sum(v["launches"] for v in sel)
        if n:
            tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel)
            # per kernel launch, and per training step (an API call may be several kernel launches: the four parity
            # launches of a collapsed up-sampled forward; bench.py divides the per-step figure by ITS launches per step)
            fam[name] = dict(launches=n, hbm_bytes_per_launch=tot / n, hbm_bytes_per_step=tot / STEPS)
    import bench
    json.dump(dict(csrc_digest=bench.csrc_digest(), source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1`",
                   correction="reads = 2*FETCH_SIZE*1024 (gfx950 half-count), writes = WRITE_SIZE*1024",
                   families=fam, kernels=kernels), open(out, "w"), indent=1, sort_keys=True)
    for k, v in fam.items():
        print("%-24s %6d kernel launches  %8.1f MB / launch  %9.1f MB / step" % (k, v["launches"], v["hbm_bytes_per_launch"] / 1e6, v["hbm_bytes_per_step"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:4])
