#!/bin/bash
# A/B of SOURCE variants of the register-weight kernel on the stand-alone shapes (fp16 build), same box, interleaved:
#   tools/rw_ab2.sh "<case filter>" "<src>|<flags>" "<src>|<flags>" ...      (src relative to the repo root; 2 rounds over all variants)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
cd $ROOT/diffusion-separation_amd/csrc
mkdir -p ../abl
FILTER=$1; shift
i=0
for v in "$@"; do
  src=${v%%|*}; flags=${v#*|}
  cp $ROOT/$src /tmp/rw_ab_src_$i.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DDS_HALF_F16 $flags -I$ROOT/diffusion-separation_amd/csrc -mllvm -pragma-unroll-threshold=1000000 -c /tmp/rw_ab_src_$i.hip -o /tmp/rw_ab_$i.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_rw_$i.so /tmp/rw_ab_$i.o $(ls build_f16/*.o | grep -Ev '/(conv3x3_rw\.o)$')
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for v in "$@"; do
    echo "== variant $i: $v (round $rep)"
    (cd $ROOT && RW_DT=f16 DIFFSEP_LIB_F16=$ROOT/diffusion-separation_amd/abl/lib_rw_$i.so python tools/rw_bench.py 20 "$FILTER" 2>&1 | grep -v amdgpu)
    i=$((i+1))
  done
done
