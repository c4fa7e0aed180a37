cd /root/repo
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extra-modes --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('default      ', d['value'], d['one_batch_alone_ms'])"
DIFFSEP_RW_SMALL=1 python bench.py --no-cpu-baseline --no-extra-modes --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('rw for small ', d['value'], d['one_batch_alone_ms'])"
done
python tools/shape_table.py 64 f16 2>/dev/null | grep "64->64.*@128x128\|128->128.*@32x32"
echo ---
DIFFSEP_RW_SMALL=1 python tools/shape_table.py 64 f16 2>/dev/null | grep "64->64.*@128x128\|128->128.*@32x32"
