# Same-box A/B of environment switches: alternating bench runs, one line each.
#   gpurun -- 'bash tools/ab_env.sh TAG "" "VQW_X=0" "" "VQW_Y=0" ...'   (an empty string = the default tree)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 300 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/run$i.json 2> $O/run$i.err || { echo "run $i failed"; tail -3 $O/run$i.err; exit 1; }
  python3 - "$O/run$i.json" "$e" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-40s %8.2f ms/step  %7.1f images/s" % (sys.argv[2] or "(default)", d["ms_per_step"], d["value"]))
PY
done
