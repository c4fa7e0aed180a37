cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "pyramid" 2>&1 | tail -2
python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "long_utterances" 2>&1 | grep -a "^\[nf64\|passed\|failed"
python tools/shape_table.py 64 f16 2>/dev/null | grep "thin_out\|total"
