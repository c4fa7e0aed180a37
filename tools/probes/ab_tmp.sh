python -m pytest tests/test_kernels_gpu.py -q -x -k "groupnorm or resample or upfirdn" 2>&1 | tail -2
python -m pytest tests/test_engine_gpu.py tests/test_round2_gpu.py -q -x 2>&1 | tail -3
for r in 1 2; do
python bench.py --no-extra-modes --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new', d['value'], d['one_batch_alone_ms'])"
done
