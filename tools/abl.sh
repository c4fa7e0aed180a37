#!/bin/bash
# Ablation variants of the conv kernel (profiling only): build one library per removed phase and time the
# dominant layer shapes with each.  Run on the GPU box: gpurun -- bash tools/abl.sh
cd $(dirname $0)/..
mkdir -p diffusion-separation_amd/abl
C=diffusion-separation_amd/csrc
for v in NOMFMA NOACT NOLOAD NOLDSW NOEPI; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DABL_$v -c $C/conv_mfma.hip -o /tmp/abl_$v.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DABL_$v -c $C/conv3x3_ws.hip -o /tmp/ablws_$v.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o diffusion-separation_amd/abl/lib_$v.so /tmp/abl_$v.o /tmp/ablws_$v.o $(ls $C/build/*.o | grep -Ev '/(conv_mfma\.o|conv3x3_ws\.o)$') ) &
done
wait
for v in "" NOMFMA NOACT NOLOAD NOLDSW NOEPI; do
  if [ -z "$v" ]; then unset DIFFSEP_LIB; else export DIFFSEP_LIB=$PWD/diffusion-separation_amd/abl/lib_$v.so; fi
  echo "== variant ${v:-BASE}"; timeout 60 python tools/bench_conv.py bf16 20 ${1:-0,1,2,5,14} 2>&1 | grep -E "^k"
done
