# Timing-only A/B builds of conv_wino.hip (results are wrong in the variants): which part of the Winograd kernel's time goes
# where.  Build here (hipcc cross-compiles):  bash tools/wino_ab.sh build      On the GPU box:  bash tools/wino_ab.sh run
set -e
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; C=$R/medical-image-editing_amd/csrc; L=$R/medical-image-editing_amd/lib; O=$R/tools/ab_build
K=${K:-conv_wino64}; P=${P:-W6_EXP_}
VARIANTS=${VARIANTS:-"NO_BARRIER NO_XFORM NO_LOADS NO_HLOADS NO_ULOADS NO_BREAD NO_EPI NO_XFORM+NO_LOADS+NO_BREAD+NO_EPI+NO_BARRIER"}
if [ "$1" = build ]; then
  mkdir -p $O
  for v in $VARIANTS; do
    defs=$(echo $v | sed "s/+/ -D$P/g; s/^/-D$P/"); case $v in *=*) defs="-D$v";; esac
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize $defs -c $C/$K.hip -o $O/${K}_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libvqwnet_$v.so $(ls $C/build/*.o | grep -v /$K.o) $O/${K}_$v.o
    rm $O/${K}_$v.o
  done
  exit 0
fi
for l in $L/libvqwnet_hip.so $(for v in $VARIANTS; do echo $O/libvqwnet_$v.so; done); do
  echo "== $l"
  VQW_LIB_PATH=$l python3 $R/tools/conv_bench.py ${PASS:+--only $PASS} --filter "k3 d1" 2>/dev/null | grep -E "${SHAPES:- 64-> 64 k3 d1  @128|128->128 k3 d1  @ 64|256->512 k3 d1  @ 32|512->512 k3 d1  @ 16}"
done
