#!/bin/bash
# Streamed-weight kernel in the engine: same-box A/B of the throughput mode (DIFFSEP_NO_SW=1: the dispatch of before), then the GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for round in 1 2; do
  for v in 1 0; do
    echo "== round $round  NO_SW=$v"
    DIFFSEP_NO_SW=$v timeout 600 python tools/bench_brief.py "@64x64" 2>&1 | head -24
  done
done > gpurun_out/sw_engine_ab.txt 2>&1
grep -E "^==|^value" gpurun_out/sw_engine_ab.txt
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_s20.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_s20.txt
