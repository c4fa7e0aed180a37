cd /root/repo
python -m pytest tests/test_rw_gpu.py -x -q 2>&1 | tail -2
python tools/rw_bench.py 20 "cat(64,64)" 2>&1 | grep -v amdgpu
python tools/rw_bench.py 20 "cat(64,64)" 2>&1 | grep -v amdgpu
python bench.py --no-cpu-baseline --no-extra-modes > gpurun_out/b_nwl8.log 2>&1
python - <<'PY'
import json
for n in ("nwl8",):
    for line in open(f'gpurun_out/b_{n}.log'):
        if line.startswith('{'):
            d=json.loads(line); r=d['roofline']
            print(n, d['value'], d['one_batch_alone_ms'], r['kernel'], r['frac'], r['avg_launch_us'], [ (s['shape'],s['avg_us']) for s in r['per_shape'][:4]])
PY
