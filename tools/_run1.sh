cd /root/repo
python -m pytest tests/test_rw_gpu.py tests/test_f16_gpu.py -x -q 2>&1 | tail -4
python tools/rw_bench.py 20 2>&1 | grep -v amdgpu
python tools/rw_bench.py 20 2>&1 | grep -v amdgpu
