#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sw_gpu.py tests/test_split_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/sws_test2.txt
cat gpurun_out/sws_test2.txt
for round in 1 2; do for v in 1 0; do
  echo "== round $round NO_SWS=$v"; DIFFSEP_NO_SWS=$v timeout 600 python bench.py --dtype split --no-extra-modes --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep "^{" | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('value', r['value'], 'ms_per_step', r['ms_per_step'], 'alone', r.get('one_batch_alone_ms'), 'in flight', r['config'].get('batches_in_flight'))"
done; done > gpurun_out/sws_engine_ab.txt 2>&1
cat gpurun_out/sws_engine_ab.txt
