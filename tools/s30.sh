#!/bin/bash
# Streamed-weight kernel: middle phases on all 8 rows per fragment (default) against half-phases everywhere (-DSW_NO_FULL)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sw_gpu.py -x -q -m gpu -k "not sws" 2>&1 | tail -3
for r in 1 2; do
  echo "== half-phases everywhere (round $r)"; DIFFSEP_LIB_F16=$PWD/diffusion-separation_amd/ab/lib_sw_nofull.so python tools/sw_bench.py 20 2>&1 | grep -v amdgpu | cut -c1-75
  echo "== middle phases on all rows (round $r)"; python tools/sw_bench.py 20 2>&1 | grep -v amdgpu | cut -c1-75
done > gpurun_out/sw_full_ab.txt 2>&1
cat gpurun_out/sw_full_ab.txt
