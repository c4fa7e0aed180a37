#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( bash tools/sw_ab_run.sh "256^2" timing; echo "##### stores outside"; DIFFSEP_SW_DBG=1 bash tools/sw_ab_run.sh "128->128 @256^2" timing; echo "##### loads outside"; DIFFSEP_SW_DBG=2 bash tools/sw_ab_run.sh "128->128 @256^2" timing ) > gpurun_out/sw_timing1.txt 2>&1
timeout 900 python -m pytest tests/test_sw_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/sw_test.txt
cat gpurun_out/sw_test.txt
