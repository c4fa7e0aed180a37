#!/bin/bash
# Everything under profiles/r03_* in one GPU call: PMC passes, per-shape tables, the full bench line, the rocprofv3 kernel
# trace of the bench command.  Run via gpurun from the repo root, then copy gpurun_out/{pmc_*,r03_*} into profiles/
# (tools/kernel_stats_md.py turns r03_kernel_stats.csv into the markdown table).
cd /root/repo
export COMMIT=720a6a8
timeout 900 bash tools/pmc_r03.sh gpurun_out/pmc_r03 f16 > gpurun_out/pmc_r03.log 2>&1
timeout 900 bash tools/pmc_traffic.sh gpurun_out/pmc_traffic f16 > gpurun_out/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic/pmc_conv3x3.json profiles/pmc_conv3x3.json
cp gpurun_out/pmc_r03/summary.json profiles/r03_pmc_mfma_util.json
python tools/shape_table.py 64 f16 > gpurun_out/r03_by_shape_nf64.md 2>/dev/null
python tools/shape_table.py 128 f16 > gpurun_out/r03_by_shape_nf128.md 2>/dev/null
python bench.py > gpurun_out/r03_bench_f16.json 2> gpurun_out/r03_bench_f16.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python bench.py --no-cpu-baseline --no-extra-modes --no-roofline > gpurun_out/r03_kt.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) gpurun_out/r03_kernel_stats.csv
tail -c 600 gpurun_out/r03_bench_f16.json
