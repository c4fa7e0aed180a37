#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for round in 1 2; do
  echo "== round $round"; timeout 600 python bench.py --dtype split --no-extra-modes --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep "^{" | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('value', r['value'], 'ms_per_step', r['ms_per_step'], 'alone', r.get('one_batch_alone_ms'), 'in flight', r['config'].get('batches_in_flight'))"
done > gpurun_out/sws_engine_ab2.txt 2>&1
cat gpurun_out/sws_engine_ab2.txt
timeout 1500 python -m pytest tests/test_split_gpu.py -x -q -m gpu 2>&1 | tail -3
