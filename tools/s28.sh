#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sw_gpu.py -x -q -m gpu -k "not sws" 2>&1 | grep -v "^$" | tail -12 > gpurun_out/sw_test4.txt
cat gpurun_out/sw_test4.txt
for round in 1 2; do for v in 1 0; do
  echo "== round $round NO_SW=$v"; DIFFSEP_NO_SW=$v timeout 900 python bench.py --nf 128 --in-flight 2 --no-extra-modes --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep "^{" | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('value', r['value'], 'ms_per_step', r['ms_per_step'], 'alone', r.get('one_batch_alone_ms'), r['roofline']['kernel'], r['roofline']['frac'])"
done; done > gpurun_out/sw_nf128_ab.txt 2>&1
cat gpurun_out/sw_nf128_ab.txt
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3
