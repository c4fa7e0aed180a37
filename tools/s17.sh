#!/bin/bash
# A/B of the two-launch register-weight route at the 64-row level in the throughput mode (DIFFSEP_NO_SPLIT64=1 = generic tile there).
cd /root/repo; mkdir -p gpurun_out
for round in 1 2 3; do
  for v in 0 1; do
    echo "== round $round  NO_SPLIT64=$v"
    DIFFSEP_NO_SPLIT64=$v timeout 600 python tools/bench_brief.py "@64x64" 2>&1 | head -12
  done
done > gpurun_out/split64_ab.txt 2>&1
grep -E "^==|^value" gpurun_out/split64_ab.txt
timeout 1200 python -m pytest tests/test_fullsize_gpu.py tests/test_engine_gpu.py tests/test_round5_gpu.py -m gpu -q -x > gpurun_out/pytest_s17.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_s17.txt
