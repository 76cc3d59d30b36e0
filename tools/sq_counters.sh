# SQ / LDS counters of the conv kernels on one layer shape, one rocprofv3 pass per counter group (kernel trace only).
#   gpurun -- 'bash tools/sq_counters.sh "128->128 k3 d1  @ 64" wgrad'   ->   gpurun_out/sq/<group>/..., summary on stdout
F=${1:-"128->128 k3 d1  @ 64"}; P=${2-dgrad}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o t -- python3 $R/tools/conv_bench.py --filter "$F" ${P:+--only $P} --iters 2 > /dev/null 2> $O/g$i.err || echo "group $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "conv" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.0f  (mean of %d dispatches)" % (c, sum(v) / len(v), len(v)))
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
        # MFMA busy cycles are summed over 1024 SIMDs, SQ_BUSY_CYCLES over 32 shader engines; 32 busy cycles per 16x16x4 fp32 MFMA
        line = "   derived: matrix-pipe busy %.3f of the kernel's duration" % ((m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (m["SQ_BUSY_CYCLES"] / 32))
        if "SQ_INSTS_VALU" in m:
            nm = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 32
            line += ", %.2f other VALU instructions per MFMA" % ((m["SQ_INSTS_VALU"] - nm) / nm)
        print(line)
PY
