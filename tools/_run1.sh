cd /root/repo
python -m pytest tests/test_kernels_gpu.py tests/test_round2_gpu.py -x -q 2>&1 | tail -2
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extra-modes --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('deep   ', d['value'], d['one_batch_alone_ms'])"
DIFFSEP_NO_DEEP32=1 python bench.py --no-cpu-baseline --no-extra-modes --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('shallow', d['value'], d['one_batch_alone_ms'])"
done
python tools/shape_table.py 64 f16 2>/dev/null | grep "@32x32"
