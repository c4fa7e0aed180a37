#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
echo "##### baseline"; python tools/bench_brief.py "@256x256" 2>&1 | grep -v amdgpu | head -2
echo "##### QUARTER"; DIFFSEP_RW_QUARTER=1 python tools/bench_brief.py "@256x256" 2>&1 | grep -v amdgpu | head -2
echo "##### QUARTER + BIG_HALF"; DIFFSEP_RW_QUARTER=1 DIFFSEP_RW_BIG_HALF=1 python tools/bench_brief.py "@256x256" 2>&1 | grep -v amdgpu | head -2
echo "##### QUARTER+HALF (eighth at 128^2)"; DIFFSEP_RW_QUARTER=1 DIFFSEP_RW_HALF=1 python tools/bench_brief.py "@256x256" 2>&1 | grep -v amdgpu | head -2
echo "##### baseline"; python tools/bench_brief.py "@256x256" 2>&1 | grep -v amdgpu | head -2
echo "##### QUARTER"; DIFFSEP_RW_QUARTER=1 python tools/bench_brief.py "@256x256" 2>&1 | grep -v amdgpu | head -2
} > gpurun_out/ab_s13.txt 2>&1
cat gpurun_out/ab_s13.txt | cut -c1-250
