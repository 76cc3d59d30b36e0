# Timing-only A/B builds of conv_wino.hip (results are wrong in the variants): which part of the Winograd kernel's time is
# the output store path.   bash tools/wino_ab.sh   (on the GPU box: gpurun -- 'bash tools/wino_ab.sh')
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; C=$R/medical-image-editing_amd/csrc; L=$R/medical-image-editing_amd/lib
for v in NO_STORE; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DWN_EXP_$v -c $C/conv_wino.hip -o /tmp/conv_wino_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libvqwnet_$v.so $(ls $C/build/*.o | grep -v conv_wino) /tmp/conv_wino_$v.o
done
for l in $L/libvqwnet_hip.so /tmp/libvqwnet_NO_STORE.so; do
  echo "== $l"
  VQW_LIB_PATH=$l python3 $R/tools/conv_bench.py --only dgrad --filter "k3 d1" 2>/dev/null | grep -E "32-> 32 k3 d1  @256|64-> 64 k3 d1  @128|256->512 k3 d1  @ 32|32-> 64 k3 d1  @256"
done
