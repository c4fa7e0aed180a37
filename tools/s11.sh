#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
echo "##### baseline"; python tools/bench_brief.py 2>&1 | grep -v amdgpu
echo "##### DIFFSEP_RW_HALF=1"; DIFFSEP_RW_HALF=1 python tools/bench_brief.py 2>&1 | grep -v amdgpu
echo "##### baseline again"; python tools/bench_brief.py 2>&1 | grep -v amdgpu
echo "##### gn8 fp32 (thin_out / ws1 activation as in round 4)"
bash tools/src_ab.sh conv3x3_ws "tools/bench_brief.py thin_out" "-DDS_GN8_F32" 2>&1 | grep -v amdgpu
} > gpurun_out/ab_s11.txt 2>&1
timeout 1200 bash tools/rw_ab2.sh "conv0" "diffusion-separation_amd/csrc/conv3x3_rw.hip|" "diffusion-separation_amd/csrc/conv3x3_rw.hip|-DRW_W_E=1 -DRW_W_N=3" "diffusion-separation_amd/csrc/conv3x3_rw.hip|-DRW_W_E=3 -DRW_W_N=5" "diffusion-separation_amd/csrc/conv3x3_rw.hip|-DRW_W_E=1 -DRW_W_N=1" > gpurun_out/rw_ab_s11.txt 2>&1
cat gpurun_out/ab_s11.txt
