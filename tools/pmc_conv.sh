#!/bin/bash
# PMC passes over the conv micro-benchmark (one counter group per run, kernel-trace only).
# usage: tools/pmc_conv.sh <dtype> <shape-index-list> <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DT=${1:-bf16}; SEL=${2:-0}; OUT=${3:-gpurun_out/pmc}
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o pmc --output-format csv -- python tools/bench_conv.py $DT 5 $SEL > $OUT/p$i.log 2>&1
done
ls -R $OUT | head -30
