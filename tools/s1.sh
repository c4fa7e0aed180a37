#!/bin/bash
# round 5, GPU session 1: issue probe, A/B of the packed / staged activation, parity of the new kernel, bench line
cd /root/repo
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result tools/probes/mfma_valu_mix.hip -o /tmp/mfma_valu_mix && /tmp/mfma_valu_mix > gpurun_out/probe_mix.txt 2>&1
echo "probe rc=$?"
RW_AB_MORE="-DRW_ACT_F32;-DRW_NO_PIPE" timeout 1200 bash tools/rw_ab.sh "" "-DRW_ACT_F32 -DRW_NO_PIPE" "conv0" > gpurun_out/rw_ab_s1.txt 2>&1
echo "ab rc=$?"
timeout 1200 python -m pytest tests/test_rw_gpu.py tests/test_engine_gpu.py -m gpu -x -q > gpurun_out/pytest_s1.txt 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_s1.txt
timeout 900 python bench.py > gpurun_out/bench_s1.json 2> gpurun_out/bench_s1.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/bench_s1.json
