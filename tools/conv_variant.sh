#!/bin/bash
# Build variants of the generic conv kernel with extra -D flags and run the whole bench with each.
# usage: tools/conv_variant.sh "-DFLAG=1" "-DFLAG=2" ...
cd $(dirname $0)/..
C=diffusion-separation_amd/csrc
mkdir -p diffusion-separation_amd/abl
i=0
for f in "$@"; do
  i=$((i+1))
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $f -c $C/conv_mfma.hip -o /tmp/cv$i.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o diffusion-separation_amd/abl/lib_cv$i.so /tmp/cv$i.o $(ls $C/build/*.o | grep -Ev '/(conv_mfma\.o)$') ) &
done
wait
for rep in 1 2; do
  echo "== default"; timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-110
  i=0
  for f in "$@"; do
    i=$((i+1))
    echo "== variant $f"; DIFFSEP_LIB=$PWD/diffusion-separation_amd/abl/lib_cv$i.so timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-110
  done
done
