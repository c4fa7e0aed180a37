#!/bin/bash
# Ablations of the register-weight conv kernel (results are wrong by construction: timing only).  Run via gpurun.
set -e
cd $(dirname $0)/../diffusion-separation_amd/csrc
mkdir -p ../abl
VARIANTS=("" "-DRW_ABL_NOEPI" "-DRW_ABL_NOSTAGE" "-DRW_ABL_NOEPI -DRW_ABL_NOSTAGE" "-DRW_ABL_NOSTORE" "-DRW_ABL_NOLOAD" "-DRW_ABL_NOSTATS")
for v in "${VARIANTS[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC $v -mllvm -pragma-unroll-threshold=1000000 -c conv3x3_rw.hip -o /tmp/rw_a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_rwa.so /tmp/rw_a.o $(ls build/*.o | grep -Ev '/(conv3x3_rw\.o)$')
  echo "== variant: ${v:-shipped}"
  (cd ../.. && DIFFSEP_LIB=$PWD/diffusion-separation_amd/abl/lib_rwa.so python tools/rw_bench.py 10 2>&1 | grep -v amdgpu | grep -v "res" | grep "64->64 conv0  \|64->64 plain\|cat(64,64)->64 conv0  ")
done
