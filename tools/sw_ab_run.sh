#!/bin/bash
# Run the variants built by tools/sw_ab_build.sh on one box, interleaved, two rounds: tools/sw_ab_run.sh "<case filter>" [timing]
ROOT=$(cd $(dirname $0)/.. && pwd)
cd $ROOT
n=$(ls diffusion-separation_amd/ab/lib_sw_*.so | wc -l)
for rep in 1 2; do
  for i in $(seq 0 $((n-1))); do
    echo "== variant $(grep "^$i:" diffusion-separation_amd/ab/variants.txt) (round $rep)"
    if [ "$2" = "timing" ]; then
      DIFFSEP_LIB_F16=$ROOT/diffusion-separation_amd/ab/lib_sw_$i.so python tools/sw_timing.py "$1" 2>&1 | grep -v amdgpu
    else
      DIFFSEP_LIB_F16=$ROOT/diffusion-separation_amd/ab/lib_sw_$i.so python tools/sw_bench.py 20 "$1" 2>&1 | grep -v amdgpu
    fi
  done
  [ "$2" = "timing" ] && break
done
