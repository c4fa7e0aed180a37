#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_s4.json 2> gpurun_out/bench_s4.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/bench_s4.json
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_s4.txt 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_s4.txt
