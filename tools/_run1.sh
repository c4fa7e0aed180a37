cd /root/repo
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extra-modes --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('folded ', d['value'], d['one_batch_alone_ms'])"
DIFFSEP_GN_ARRAYS=1 python bench.py --no-cpu-baseline --no-extra-modes --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('arrays ', d['value'], d['one_batch_alone_ms'])"
done
