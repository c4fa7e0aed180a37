#!/bin/bash
# A/B of compile-time variants of the register-weight kernel on the stand-alone shapes (fp16 build):
#   tools/rw_ab.sh "<flags A>" "<flags B>" [case filter]        (RW_AB_MORE="<flags C>;<flags D>" adds variants)
set -e
cd $(dirname $0)/../diffusion-separation_amd/csrc
mkdir -p ../abl
IFS=';' read -ra MORE <<< "$RW_AB_MORE"
for v in "$1" "$2" "${MORE[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DDS_HALF_F16 $v -mllvm -pragma-unroll-threshold=1000000 -c conv3x3_rw.hip -o /tmp/rw_a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_rwa.so /tmp/rw_a.o $(ls build_f16/*.o | grep -Ev '/(conv3x3_rw\.o)$')
  for rep in 1 2; do
    echo "== variant: ${v:-shipped} (run $rep)"
    (cd ../.. && RW_DT=f16 DIFFSEP_LIB_F16=$PWD/diffusion-separation_amd/abl/lib_rwa.so python tools/rw_bench.py 10 "$3" 2>&1 | grep -v amdgpu)
  done
done
