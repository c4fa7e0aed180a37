#!/bin/bash
# Build a variant of the weight-stationary conv kernel with extra -D flags and time it against the default.
# usage: tools/ws_variant.sh "-DFLAG1 -DFLAG2" [shape list]
cd $(dirname $0)/..
C=diffusion-separation_amd/csrc
mkdir -p diffusion-separation_amd/abl
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $1 -c $C/conv3x3_ws.hip -o /tmp/wsv.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o diffusion-separation_amd/abl/lib_wsv.so /tmp/wsv.o $(ls $C/build/*.o | grep -Ev '/(conv3x3_ws\.o)$')
echo "== default"; timeout 100 python tools/bench_conv.py bf16 20 ${2:-0,2} 2>&1 | grep "^k"
echo "== variant $1"; DIFFSEP_LIB=$PWD/diffusion-separation_amd/abl/lib_wsv.so timeout 100 python tools/bench_conv.py bf16 20 ${2:-0,2} 2>&1 | grep "^k"
