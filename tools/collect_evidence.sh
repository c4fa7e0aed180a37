#!/bin/bash
# Everything under profiles/<TAG>_* in one GPU call: the PMC passes (MFMA utilisation / instruction mix, HBM traffic of the 3x3
# kernels), the per-shape tables, the ablation table, the full bench line, and the rocprofv3 --kernel-trace --stats summaries of the
# bench command (four batches in flight = the timed region; one batch at a time = what roofline.frac is measured on).
#   gpurun --timeout 3000 -- 'TAG=r06 COMMIT=<hash> bash tools/collect_evidence.sh'
# writes gpurun_out/$TAG/*; copy what is to be judged into profiles/ (README there names every file and its tool).
cd ${GRAFT_REPO_ROOT:-.}
TAG=${TAG:-r06}
COMMIT=${COMMIT:-$(git rev-parse --short HEAD 2>/dev/null)}
export COMMIT
DT=${DT:-f16}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 bash tools/pmc_util.sh $O/pmc_util $DT > $O/pmc_util.log 2>&1
timeout 900 bash tools/pmc_traffic.sh $O/pmc_traffic $DT > $O/pmc_traffic.log 2>&1
cp $O/pmc_traffic/pmc_conv3x3.json $O/pmc_conv3x3.json
cp $O/pmc_util/summary.json $O/${TAG}_pmc_mfma_util.json
python tools/shape_table.py 64 $DT > $O/${TAG}_by_shape_nf64.md 2>/dev/null
python tools/shape_table.py 128 $DT > $O/${TAG}_by_shape_nf128.md 2>/dev/null
python tools/ablate_bench.py > $O/${TAG}_ablation.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
for MODE in inflight alone; do
  EXTRA=""; SUF=""
  if [ $MODE = alone ]; then EXTRA="--in-flight 1"; SUF="_alone"; fi
  rm -rf /tmp/kt
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python bench.py --dtype $DT --no-cpu-baseline --no-extra-modes --no-roofline $EXTRA > $O/${TAG}_kt$SUF.log 2>&1
  cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats$SUF.csv
  python tools/kernel_stats_md.py $O/${TAG}_kernel_stats$SUF.csv "rocprofv3 --kernel-trace --stats summary of \`python bench.py --dtype $DT --no-cpu-baseline --no-extra-modes --no-roofline $EXTRA\` ($TAG, commit $COMMIT; $MODE: 2 warm-up + 8 timed steps + the latency / bit-identity extras; the tracer serialises dispatches)" > $O/${TAG}_bench_${DT}${SUF}_kernel_stats.md
done
# (the bench line last: its roofline.frac_rocprof reads the summaries just written when they have been copied to profiles/)
cp $O/${TAG}_bench_${DT}_kernel_stats.md $O/${TAG}_bench_${DT}_alone_kernel_stats.md $O/${TAG}_pmc_mfma_util.json profiles/ 2>/dev/null
cp $O/pmc_conv3x3.json profiles/pmc_conv3x3.json 2>/dev/null
python bench.py --dtype $DT > $O/${TAG}_bench_$DT.json 2> $O/${TAG}_bench_$DT.err
tail -c 1500 $O/${TAG}_bench_$DT.json
