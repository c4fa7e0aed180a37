#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_round5_gpu.py tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_f16_gpu.py -m gpu -q -x -k "stft or istft or concurrent or pre_process or score_forward or fir_down or resample" > gpurun_out/pytest_s9.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_s9.txt
timeout 900 python bench.py --no-cpu-baseline --no-extra-modes > gpurun_out/bench_s9.json 2> gpurun_out/bench_s9.err
echo "bench rc=$?"; python - <<'PY'
import json
r=json.load(open('gpurun_out/bench_s9.json'))
print(r['value'], r['ms_per_step'], r['one_batch_alone_ms'])
for h in r['roofline']['hbm_kernels'][:4]:
    print(h['kernel'][:40], h['avg_us'], h['frac_hbm'])
PY
