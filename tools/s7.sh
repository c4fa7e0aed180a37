#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_round5_gpu.py tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_f16_gpu.py -m gpu -q -s -x --deselect tests/test_round5_gpu.py::test_nf128_hybrid_default_full_length_parity_with_oracle > gpurun_out/pytest_s7.txt 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_s7.txt; grep -E "^\[f16 range|^\[f16 B=16|FAILED|^E  " gpurun_out/pytest_s7.txt | head -40
python tools/precision_probe.py 64 16 2>&1 | grep -v amdgpu
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_s7.json 2> gpurun_out/bench_s7.err
echo "bench rc=$?"; cut -c1-200 gpurun_out/bench_s7.json
