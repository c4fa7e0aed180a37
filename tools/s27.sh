#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_s27.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_s27.txt
for k in 3 5 6; do
  echo "== in flight $k"; timeout 600 python bench.py --no-extra-modes --no-cpu-baseline --no-roofline --in-flight $k 2>&1 | grep "^{" | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('value', r['value'], 'ms_per_step', r['ms_per_step'])"
done > gpurun_out/inflight_sweep.txt 2>&1
cat gpurun_out/inflight_sweep.txt
