#!/bin/bash
# Ablations of the one-phase weight-stationary kernel that keep the data flow alive (no dead-code elimination):
# usage: tools/ws_ablate.sh   (BENCH_CONV_PLAIN=1 for the plain launch)
cd $(dirname $0)/..
C=diffusion-separation_amd/csrc
mkdir -p diffusion-separation_amd/abl
for v in NOLOAD NOACT NOLDSW NOMFMA NOEPI; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DABL2_$v -c $C/conv3x3_ws.hip -o /tmp/wsa_$v.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o diffusion-separation_amd/abl/lib_wsa_$v.so /tmp/wsa_$v.o $(ls $C/build/*.o | grep -Ev '/(conv3x3_ws\.o)$') ) &
done
wait
echo "== BASE"; timeout 60 python tools/bench_conv.py bf16 20 0 2>&1 | grep "^k"
for v in NOLOAD NOACT NOLDSW NOMFMA NOEPI; do
  echo "== $v"; DIFFSEP_LIB=$PWD/diffusion-separation_amd/abl/lib_wsa_$v.so timeout 60 python tools/bench_conv.py bf16 20 0 2>&1 | grep "^k"
done
