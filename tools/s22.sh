#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sw_gpu.py -x -q -m gpu -k "sws or frag_index" 2>&1 | tail -15 > gpurun_out/sws_test.txt
timeout 300 python tools/sws_bench.py 10 > gpurun_out/sws_bench.txt 2>&1
cat gpurun_out/sws_test.txt gpurun_out/sws_bench.txt
