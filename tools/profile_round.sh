set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r1h; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py --steps 20 --warmup 3 > $O/bench_steps20.json 2> $O/bench_steps20.err
echo "bench20 done"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/conc -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/conc.err
echo "conc done"
export VQW_WGRAD_STREAM=0 VQW_CONCURRENT_VIEWS=0
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ser -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof_serialised.json 2> $O/ser.err
unset VQW_WGRAD_STREAM VQW_CONCURRENT_VIEWS
echo "ser done"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "fetch done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_write.json 2> $O/pmc_write.err
echo "write done"
rm -f $O/pmc_fetch/*kernel_trace.csv $O/pmc_write/*kernel_trace.csv
ls -la $O $O/pmc_fetch | head -30
