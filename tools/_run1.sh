cd /root/repo
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extra-modes --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ws for res ', d['value'], d['one_batch_alone_ms'])"
DIFFSEP_RW_RES=1 python bench.py --no-cpu-baseline --no-extra-modes --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('rw for res ', d['value'], d['one_batch_alone_ms'])"
done
DIFFSEP_RW_RES=1 python tools/shape_table.py 64 f16 2>/dev/null | grep "+res @256\|+res @128"
