#!/bin/bash
# A/B of compile-time variants of the register-weight kernel on the stand-alone shapes: tools/rw_ab.sh "<flags A>" "<flags B>" [case filter]
set -e
cd $(dirname $0)/../diffusion-separation_amd/csrc
mkdir -p ../abl
for v in "$1" "$2"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC $v -mllvm -pragma-unroll-threshold=1000000 -c conv3x3_rw.hip -o /tmp/rw_a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_rwa.so /tmp/rw_a.o $(ls build/*.o | grep -Ev '/(conv3x3_rw\.o)$')
  for rep in 1 2; do
    echo "== variant: ${v:-shipped} (run $rep)"
    (cd ../.. && DIFFSEP_LIB=$PWD/diffusion-separation_amd/abl/lib_rwa.so python tools/rw_bench.py 10 "$3" 2>&1 | grep -v amdgpu)
  done
done
