cd /root/repo
python -m pytest tests/test_fullsize_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -150 > gpurun_out/fullsize_tests.log
tail -c 3000 gpurun_out/fullsize_tests.log
