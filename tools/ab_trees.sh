# Same-box comparison of two trees (each with its own bench.py and library), alternating runs.
#   gpurun -- 'bash tools/ab_trees.sh TAG tools/ab_build/r03 . tools/ab_build/r03 . ...'   (paths relative to the repo root)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
i=0
for t in "$@"; do
  i=$((i+1))
  (cd $R/$t && timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/run$i.json 2> $O/run$i.err) || { echo "run $i failed"; tail -3 $O/run$i.err; exit 1; }
  python3 - "$O/run$i.json" "$t" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s %8.2f ms/step  %7.1f images/s" % (sys.argv[2], d["ms_per_step"], d["value"]))
PY
done
