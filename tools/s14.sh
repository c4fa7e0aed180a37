#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for k in 4 5 6 8 4; do
  echo "##### --in-flight $k"; python bench.py --no-extra-modes --no-cpu-baseline --no-roofline --in-flight $k --steps 12 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['one_batch_alone_ms'], r.get('throughput_mode'))"
done 2>&1 | cut -c1-300
