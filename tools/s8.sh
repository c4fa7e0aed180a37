#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -s --deselect tests/test_round5_gpu.py::test_nf128_hybrid_default_full_length_parity_with_oracle > gpurun_out/pytest_s8.txt 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pytest_s8.txt | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/pytest_s8.txt | head -20
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_s8.json 2> gpurun_out/bench_s8.err
echo "bench rc=$?"; cut -c1-200 gpurun_out/bench_s8.json
