#!/bin/bash
# round 5, streamed-weight kernel: parity first, then the stand-alone A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sw_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/sw_test.txt
timeout 300 python tools/sw_bench.py 20 > gpurun_out/sw_bench.txt 2>&1
cat gpurun_out/sw_test.txt gpurun_out/sw_bench.txt
