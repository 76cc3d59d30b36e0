# Kernel trace of the default bench (concurrent schedule) + per-stream timeline of its last concurrent step.
#   gpurun -- 'bash tools/trace_step.sh TAG'   ->  gpurun_out/TAG/{timeline.txt, timeline.json, t_kernel_trace.csv, t_kernel_stats.csv, bench.json}
set -e
TAG=${1:-trace}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/bench_traced.json 2> $O/bench_traced.err
F=$(ls $O/tr/*/t_kernel_trace.csv 2>/dev/null | head -1); [ -z "$F" ] && F=$O/tr/t_kernel_trace.csv
python3 $R/tools/stream_timeline.py $F 5 1 --json $O/timeline.json > $O/timeline.txt
S=$(ls $O/tr/*/t_kernel_stats.csv 2>/dev/null | head -1); [ -z "$S" ] && S=$O/tr/t_kernel_stats.csv
cp $S $O/t_kernel_stats.csv
# keep the trace of the last step only (the whole trace is tens of MB)
python3 - "$F" "$O/last_step_trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(ev) if "adam" in r["Kernel_Name"].lower()]
groups = []
for i in adam:
    if groups and int(ev[i]["Start_Timestamp"]) - int(ev[groups[-1][-1]]["End_Timestamp"]) < 5e6:
        groups[-1].append(i)
    else:
        groups.append([i])
a, b = groups[-2][-1] + 1, groups[-1][-1] + 1
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["start_ns", "end_ns", "stream", "grid", "wg", "kernel"])
t0 = int(ev[a]["Start_Timestamp"])
for r in ev[a:b]:
    w.writerow([int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Stream_Id"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r["Kernel_Name"][:160]])
PY
rm -rf $O/tr
tail -5 $O/timeline.txt
