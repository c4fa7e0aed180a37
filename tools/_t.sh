cd /root/repo
python -m pytest tests/test_rw_gpu.py -x -q 2>&1 | tail -2
python tools/rw_bench.py 20 2>&1 | grep -v amdgpu
