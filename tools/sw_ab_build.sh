#!/bin/bash
# Build SOURCE / FLAG variants of the streamed-weight kernel (fp16 build) HERE into diffusion-separation_amd/ab/ (travels with gpurun):
#   tools/sw_ab_build.sh "<flags>" "<flags>" ...   ->  ab/lib_sw_0.so, ab/lib_sw_1.so, ...   (run them with tools/sw_ab_run.sh)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
cd $ROOT/diffusion-separation_amd/csrc
mkdir -p ../ab
rm -f ../ab/lib_sw_*.so ../ab/variants.txt
i=0
for flags in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DDS_HALF_F16 $flags -mllvm -pragma-unroll-threshold=1000000 -c conv3x3_sw.hip -o /tmp/sw_ab_$i.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/lib_sw_$i.so /tmp/sw_ab_$i.o $(ls build_f16/*.o | grep -Ev '/(conv3x3_sw\.o)$') ) &
  echo "$i: $flags" >> ../ab/variants.txt
  i=$((i+1))
done
wait
cat ../ab/variants.txt
