#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
python tools/precision_probe.py 64 16 2>&1 | grep -v amdgpu
echo "##### baseline"; python tools/bench_brief.py "@128x128" 2>&1 | grep -v amdgpu | head -3
echo "##### DIFFSEP_RW_HALF=1"; DIFFSEP_RW_HALF=1 python tools/bench_brief.py "@128x128" 2>&1 | grep -v amdgpu | head -3
echo "##### DIFFSEP_RW_QUARTER=1"; DIFFSEP_RW_QUARTER=1 python tools/bench_brief.py "@128x128" 2>&1 | grep -v amdgpu | head -3
echo "##### baseline"; python tools/bench_brief.py "@128x128" 2>&1 | grep -v amdgpu | head -3
echo "##### DIFFSEP_RW_HALF=1"; DIFFSEP_RW_HALF=1 python tools/bench_brief.py "@128x128" 2>&1 | grep -v amdgpu | head -3
} > gpurun_out/ab_s12.txt 2>&1
cat gpurun_out/ab_s12.txt | cut -c1-250
