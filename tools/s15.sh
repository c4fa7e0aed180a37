#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1200 bash tools/rw_ab2.sh "64->64" "diffusion-separation_amd/csrc/conv3x3_rw.hip|" "diffusion-separation_amd/csrc/conv3x3_rw.hip|-DRW_NO_PREACT" > gpurun_out/rw_ab_s15.txt 2>&1
grep -E "variant|res|conv0  " gpurun_out/rw_ab_s15.txt | cut -c1-120
timeout 1500 python -m pytest tests/test_rw_gpu.py tests/test_engine_gpu.py tests/test_f16_gpu.py -m gpu -x -q > gpurun_out/pytest_s15.txt 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/pytest_s15.txt
python tools/precision_probe.py 64 16 2>&1 | grep -v amdgpu
python tools/bench_brief.py "+res" "+skip64" 2>&1 | grep -v amdgpu | head -8
