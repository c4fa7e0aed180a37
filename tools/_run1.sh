cd /root/repo
RW_DT=f16 python tools/rw_bench.py 20 conv0 2>&1 | grep -v amdgpu
python bench.py --no-cpu-baseline --no-extra-modes > gpurun_out/b_diet.log 2>&1
python bench.py --no-cpu-baseline --no-extra-modes --nf 128 --in-flight 2 > gpurun_out/b_diet_nf128.log 2>&1
python - <<'PY'
import json
for n in ("diet","diet_nf128"):
    for line in open(f'gpurun_out/b_{n}.log'):
        if line.startswith('{'):
            d=json.loads(line); r=d['roofline']
            print(n, d['value'], d['one_batch_alone_ms'], r['kernel'], r['frac'], r['avg_launch_us'])
PY
