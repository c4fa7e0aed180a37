#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for round in 1 2 3; do
  for v in "DIFFSEP_NO_SW_QUARTER=1" "DIFFSEP_X=0"; do
    echo "== round $round  $v"
    env $v timeout 600 python tools/bench_brief.py "@32x32" 2>&1 | head -3
  done
done > gpurun_out/sw_quarter_ab.txt 2>&1
grep -E "^==|^value" gpurun_out/sw_quarter_ab.txt
