#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1500 bash tools/rw_ab2.sh "" "tools/ab_src/conv3x3_rw_r04.hip|" "diffusion-separation_amd/csrc/conv3x3_rw.hip|" "diffusion-separation_amd/csrc/conv3x3_rw.hip|-DRW_ACT_F32" > gpurun_out/rw_ab_s2.txt 2>&1
echo "ab rc=$?"
timeout 1200 python -m pytest tests/test_rw_gpu.py tests/test_engine_gpu.py -m gpu -x -q > gpurun_out/pytest_s2.txt 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_s2.txt
