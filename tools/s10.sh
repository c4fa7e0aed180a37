#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
bash tools/stft_timing.sh 2>&1 | grep -v amdgpu
bash tools/s9.sh
