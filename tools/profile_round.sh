# Round profile: bench (default + 20 steps), rocprofv3 kernel summaries of the default and the serialised schedule, the two
# PMC passes for HBM traffic; config 4 (training step + VQ alone) and config 5 (run_recon) with their kernel summaries.
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r04'   ->   gpurun_out/<tag>/...
set -e
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# HBM traffic first: the bench lines below read profiles/r04_hbm_traffic.json (tied to the sources by digest)
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "fetch done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_write.json 2> $O/pmc_write.err
echo "write done"
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $R/profiles/r04_hbm_traffic.json > $O/hbm_traffic.txt
cp $R/profiles/r04_hbm_traffic.json $O/hbm_traffic.json
echo "traffic done"
timeout -k 10 500 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench default done"
timeout -k 10 500 python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench_steps20.err
echo "bench20 done"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/conc -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/conc.err
echo "conc done"
# per-stream timeline of one CONCURRENT step: from a trace of its own WITHOUT the in-library event timing (two event records per conv
# launch on top of the tracer slow the host further, and the traced schedule drifts away from the untraced one: 102.6 ms / 0.78
# matrix-core-active with the timing on, 92.4 ms / 0.86 without, 90.6 ms untraced)
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench_traced_for_timeline.json 2> $O/tl.err
F=$(ls $O/tl/*/t_kernel_trace.csv 2>/dev/null | head -1); [ -z "$F" ] && F=$O/tl/t_kernel_trace.csv
python3 $R/tools/stream_timeline.py $F 5 1 --json $O/stream_timeline.json > $O/stream_timeline.txt || echo "timeline failed"
cp $O/stream_timeline.json $R/profiles/r04_stream_timeline.json      # the bench lines below (RCCL run) and a later bench.py read it
echo "timeline done"
export VQW_WGRAD_STREAM=0 VQW_CONCURRENT_VIEWS=0
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ser -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof_serialised.json 2> $O/ser.err
unset VQW_WGRAD_STREAM VQW_CONCURRENT_VIEWS
echo "ser done"
# config 4 and config 5
timeout -k 10 300 python3 $R/tools/config4_bench.py > $O/config4_bench.json 2> $O/config4_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4 -o t -- python3 $R/tools/config4_bench.py --steps 3 --warmup 1 > /dev/null 2> $O/cfg4.err
echo "cfg4 done"
timeout -k 10 300 python3 $R/tools/recon_bench.py > $O/config5_recon_bench.txt 2> $O/config5.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg5 -o t -- python3 $R/tools/recon_bench.py > /dev/null 2> $O/cfg5.err
echo "cfg5 done"
# per-shape table of the three passes, SQ counters of the Winograd kernels on the verdict's shape, the RCCL path on one GPU
timeout -k 10 300 python3 $R/tools/conv_bench.py > $O/conv_shapes.txt 2> /dev/null
echo "conv shapes done"
bash $R/tools/sq_counters.sh "128->128 k3 d1  @ 64" dgrad > $O/sq_dgrad.txt 2>&1
bash $R/tools/sq_counters.sh "128->128 k3 d1  @ 64" wgrad > $O/sq_wgrad.txt 2>&1
bash $R/tools/sq_counters.sh "256->128 k3 d1  @ 64 up" "" > $O/sq_up.txt 2>&1
echo "sq done"
# the UNTRACED schedule of the two views (HIP events around the phases of a step)
timeout -k 10 300 python3 $R/tools/debug/phase_events.py 6 2> /dev/null | tail -13 > $O/phase_events.txt
echo "phase events done"
cd /tmp
VQW_DP_FORCE=1 timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
echo "rccl done"
rm -f $O/*/*kernel_trace.csv $O/*/*agent_info.csv; rm -rf $R/gpurun_out/sq
ls -la $O | head -40
