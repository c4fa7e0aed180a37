#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sw_gpu.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -12 > gpurun_out/sw_test3.txt
cat gpurun_out/sw_test3.txt
for round in 1 2; do
  for v in "DIFFSEP_NO_SW=1" "DIFFSEP_NO_SW_ROWS4=1" "DIFFSEP_X=0"; do
    echo "== round $round  $v"
    env $v timeout 600 python tools/bench_brief.py "@32x32" "@64x64" "192->64" 2>&1 | head -40
  done
done > gpurun_out/sw_engine_ab3.txt 2>&1
grep -E "^==|^value" gpurun_out/sw_engine_ab3.txt
