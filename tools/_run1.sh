cd /root/repo
for i in 1 2; do
python bench.py --nf 128 --in-flight 2 --no-cpu-baseline --no-extra-modes --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('fused   ', d['value'], d['one_batch_alone_ms'])"
DIFFSEP_UNFUSE_SKIP256=1 python bench.py --nf 128 --in-flight 2 --no-cpu-baseline --no-extra-modes --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('unfused ', d['value'], d['one_batch_alone_ms'])"
done
DIFFSEP_UNFUSE_SKIP256=1 python -m pytest tests/test_round2_gpu.py tests/test_fullsize_gpu.py -x -q -k "nf128" 2>&1 | tail -2
