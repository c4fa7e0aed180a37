#!/bin/bash
# Everything under profiles/r05_* in one GPU call: PMC passes, per-shape tables, the full bench line, the rocprofv3 kernel
# trace of the bench command, the ablation table.  Run via gpurun from the repo root (COMMIT=<hash> bash tools/collect_evidence_r05.sh),
# then copy gpurun_out/r05/* into profiles/ (tools/kernel_stats_md.py turns kernel_stats.csv into the markdown table).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout 900 bash tools/pmc_r05.sh $O/pmc_util f16 > $O/pmc_util.log 2>&1
timeout 900 bash tools/pmc_traffic.sh $O/pmc_traffic f16 > $O/pmc_traffic.log 2>&1
cp $O/pmc_traffic/pmc_conv3x3.json profiles/pmc_conv3x3.json
cp $O/pmc_util/summary.json profiles/r05_pmc_mfma_util.json
cp profiles/pmc_conv3x3.json $O/pmc_conv3x3.json; cp profiles/r05_pmc_mfma_util.json $O/r05_pmc_mfma_util.json
python tools/shape_table.py 64 f16 > $O/r05_by_shape_nf64.md 2>/dev/null
python tools/shape_table.py 128 f16 > $O/r05_by_shape_nf128.md 2>/dev/null
python tools/ablate_bench.py > $O/r05_ablation.txt 2>&1
python bench.py > $O/r05_bench_f16.json 2> $O/r05_bench_f16.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python bench.py --no-cpu-baseline --no-extra-modes --no-roofline > $O/r05_kt.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/r05_kernel_stats.csv
python tools/kernel_stats_md.py $O/r05_kernel_stats.csv "rocprofv3 --kernel-trace --stats summary of \`python bench.py --no-cpu-baseline --no-extra-modes --no-roofline\` (round 5, f16 build, commit $COMMIT; 4 batches in flight: 2 warm-up + 8 timed steps + the latency / bit-identity extras; the tracer serialises dispatches)" > $O/r05_bench_f16_kernel_stats.md
tail -c 1500 $O/r05_bench_f16.json
