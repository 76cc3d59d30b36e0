# The timeline part of tools/profile_round.sh alone, followed by the three bench lines that quote it.
#   gpurun -- 'bash tools/timeline_round.sh TAG'
set -e
TAG=${1:-tl}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench_traced_for_timeline.json 2> $O/tl.err
F=$(ls $O/tl/*/t_kernel_trace.csv 2>/dev/null | head -1); [ -z "$F" ] && F=$O/tl/t_kernel_trace.csv
python3 $R/tools/stream_timeline.py $F 5 1 --json $O/stream_timeline.json > $O/stream_timeline.txt
cp $O/stream_timeline.json $R/profiles/r04_stream_timeline.json
rm -rf $O/tl
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_steps20.json 2> $O/b20.err
VQW_DP_FORCE=1 python3 $R/bench.py --no-cpu-baseline > $O/bench_rccl_world1.json 2> $O/rccl.err
head -8 $O/stream_timeline.txt
