#!/bin/bash
# A/B of the GroupNorm + SiLU + FIR x2 resampling kernels (norm.hip) with compile-time variants.  Run via gpurun from the repo root.
#   VARIANTS="ship: rs8:-DFD_RS_MAX=8" bash tools/fir_ab.sh
set -e
cd ${GRAFT_REPO_ROOT:-.}/diffusion-separation_amd/csrc
mkdir -p ../abl
VARIANTS=${VARIANTS:-"ship: rs8:-DFD_RS_MAX=8 rs4:-DFD_RS_MAX=4 ship:"}
for V in $VARIANTS; do
  NAME=${V%%:*}; FL=$(echo ${V#*:} | tr ',' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DDS_HALF_F16 $FL -c norm.hip -o /tmp/nm_$NAME.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_nm_$NAME.so /tmp/nm_$NAME.o $(ls build_f16/*.o | grep -Ev '/(norm\.o)$')
done
cd ../..
for V in $VARIANTS; do
NAME=${V%%:*}
echo "== $NAME"
DIFFSEP_LIB_F16=$PWD/diffusion-separation_amd/abl/lib_nm_$NAME.so python tools/bench_resample.py 2>/dev/null
done
