#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python tools/precision_probe.py 64 16 2>&1 | grep -v amdgpu > gpurun_out/precision_s6.txt; python tools/precision_probe.py 128 4 2>&1 | grep -v amdgpu >> gpurun_out/precision_s6.txt; cat gpurun_out/precision_s6.txt
timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_s6.txt 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/pytest_s6.txt; grep -E "^\[|FAILED|Error" gpurun_out/pytest_s6.txt | head -60
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_s6.json 2> gpurun_out/bench_s6.err
echo "bench rc=$?"; cut -c1-200 gpurun_out/bench_s6.json
