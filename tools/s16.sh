#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_s16.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke_s16.txt
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/pytest_s16.txt 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/pytest_s16.txt
timeout 900 python bench.py > gpurun_out/bench_s16.json 2> gpurun_out/bench_s16.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/bench_s16.json
