# rocprofv3 kernel summary + HBM traffic (separate PMC passes) of the vector quantiser at the BASELINE shapes.
#   gpurun -- 'bash tools/vq_profile.sh r02'      -> gpurun_out/<tag>_vq/{stats,pmc_fetch,pmc_write}
set -e
TAG=${1:-r02}; CASES=${2:-"cfg4 cfg2"}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_vq; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in $CASES; do
  python3 $R/tools/vq_bench.py --case $c --iters 20 > $O/bench_$c.json 2> $O/bench_$c.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -o t -- python3 $R/tools/vq_bench.py --case $c --iters 10 > /dev/null 2> $O/stats_$c.err
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$c -o t -- python3 $R/tools/vq_bench.py --case $c --iters 3 > /dev/null 2> $O/pmc_fetch_$c.err
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$c -o t -- python3 $R/tools/vq_bench.py --case $c --iters 3 > /dev/null 2> $O/pmc_write_$c.err
  rm -f $O/pmc_fetch_$c/*kernel_trace.csv $O/pmc_write_$c/*kernel_trace.csv $O/stats_$c/*kernel_trace.csv
  echo "$c done"
done
cat $O/bench_*.json
