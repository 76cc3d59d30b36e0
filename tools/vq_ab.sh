# Timing-only A/B builds of vq.hip (-DVQ_EXP=<bits>): results are wrong by construction, only the duration of k_vq_mfma counts.
#   here:      bash tools/vq_ab.sh build "1 2 3"
#   GPU box:   bash tools/vq_ab.sh run "1 2 3"
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/medical-image-editing_amd/csrc
if [ "$1" = build ]; then
  for v in $2; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DVQ_EXP=$v -c $C/vq.hip -o $C/build/vq_exp$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/medical-image-editing_amd/lib/libvqwnet_exp$v.so $(ls $C/build/*.o | grep -v "vq_exp\|/vq.o") $C/build/vq_exp$v.o
  done
else
  python3 $R/tools/vq_bench.py --case cfg4 --eval --iters 10
  for v in $2; do echo "VQ_EXP=$v"; VQW_LIB_PATH=$R/medical-image-editing_amd/lib/libvqwnet_exp$v.so python3 $R/tools/vq_bench.py --case cfg4 --eval --iters 10; done
fi
