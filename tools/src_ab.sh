#!/bin/bash
# A/B of compile-time variants of ONE kernel source of the fp16 library on a stand-alone timing tool:
#   tools/src_ab.sh <source without .hip> "<python tool + args>" "<flags A>" "<flags B>" ...
# e.g. tools/src_ab.sh conv3x3_ws "tools/thin_bench.py 20" "" "-DTO_WPE=4"
set -e
SRC=$1; TOOL=$2; shift 2
cd $(dirname $0)/../diffusion-separation_amd/csrc
mkdir -p ../abl
EXTRA=""
[ "$SRC" = conv3x3_rw ] && EXTRA="-mllvm -pragma-unroll-threshold=1000000"
[ "$SRC" = sde ] && EXTRA="-fno-vectorize"
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DDS_HALF_F16 $v $EXTRA -c $SRC.hip -o /tmp/ab_$SRC.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_ab.so /tmp/ab_$SRC.o $(ls build_f16/*.o | grep -Ev "/($SRC\.o)\$")
  for rep in 1 2; do
    echo "== variant: ${v:-shipped} (run $rep)"
    (cd ../.. && DIFFSEP_LIB_F16=$PWD/diffusion-separation_amd/abl/lib_ab.so python $TOOL 2>&1 | grep -v amdgpu)
  done
done
