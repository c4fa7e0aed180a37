#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
export RW_CASES="128,64,256,256,1;128,64,256,256,0;64,64,256,256,1;64,64,256,256,0;128,128,256,256,1"
for dbg in 0 1 2 3; do
  echo "##### DIFFSEP_RW_DBG=$dbg (bit 0: stores fall outside the tensor, bit 1: loads do)"
  DIFFSEP_RW_DBG=$dbg timeout 600 bash tools/rw_timing16.sh 2>&1 | grep -v amdgpu
done > gpurun_out/rw_timing_s3.txt 2>&1
echo "timing rc=$?"
for fl in "-DRW_ABL_NOEPI" "-DRW_ABL_NOSTAGE" "-DRW_ABL_NOSTAGE -DRW_ABL_NOEPI"; do
  echo "##### ablation $fl"
  RW_EXTRA="$fl" timeout 600 bash tools/rw_timing16.sh 2>&1 | grep -v amdgpu
done >> gpurun_out/rw_timing_s3.txt 2>&1
tail -30 gpurun_out/rw_timing_s3.txt
timeout 600 python -m pytest tests/test_f16_gpu.py -m gpu -x -q -k "fir_down or elementwise" > gpurun_out/pytest_s3.txt 2>&1; echo "pytest fir rc=$?"; tail -4 gpurun_out/pytest_s3.txt
RW_DT=f16 timeout 300 python tools/bench_resample.py > gpurun_out/resample_s3.txt 2>&1; cat gpurun_out/resample_s3.txt | grep -v amdgpu
